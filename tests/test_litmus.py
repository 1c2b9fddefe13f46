"""The three hand-off forms of the cooperative encoders as a litmus (scripts/microbench/litmus.hip): tag-in-granule (split encoder, shared
rows), data + drained stores + arrival counter (gangs), give-up by compare-and-swap -- seconds of it in the suite, each next to a negative
control (round 5's review, Next #3).  The long run is profiles/r06r/litmus.log."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "scripts", "microbench", "litmus")


@pytest.mark.gpu
def test_cross_xcd_hand_offs_hold_and_their_controls_fail():
    assert os.path.exists(BIN), "scripts/microbench/litmus is built by __graft_entry__.build() (make -C scripts/microbench litmus)"
    r = subprocess.run([BIN, "--seconds", "7"], capture_output=True, text=True, timeout=120)
    out = r.stdout
    assert r.returncode == 0, out + r.stderr
    rows = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in re.finditer(r"^(\d\w?) .*?\s(\d+) [a-z].*?,\s+(\d+) errors", out, re.M)}
    assert set(rows) == {"1", "1c", "2", "2m", "2c", "3", "3c"}, out
    for form in ("1", "2", "2m", "3"):
        assert rows[form][0] > 1000 and rows[form][1] == 0, (form, out)
    for control in ("1c", "3c"):                     # (2c: reported, see the note the program prints)
        assert rows[control][1] > 0, (control, out)
    assert "RAN INTO THE DEADLINE" not in out
