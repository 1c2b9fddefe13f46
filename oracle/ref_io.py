"""Loads the REAL reference `rec.io` (ArithmeticCoder, write/read_compressed_code) for fixture generation and
cross-checks.  TEST INFRASTRUCTURE ONLY, and only usable in the build container: it needs /root/reference (python
sources rec/io/utils.py, rec/io/data_structures.py) plus oracle/_ref/entropy_coding*.so built by oracle/build_ref.sh.
Nothing here may be imported by -m gpu tests, smoke() or bench.py.
"""
import os
import sys
import types

REF = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REF, "rec", "io")) and any(
        f.startswith("entropy_coding") and f.endswith(".so") for f in os.listdir(os.path.join(_HERE, "_ref"))
        ) if os.path.isdir(os.path.join(_HERE, "_ref")) else False


def load():
    """Returns the reference's rec.io.utils module (with .ArithmeticCoder bound from the compiled extension)."""
    if "rec.io.utils" in sys.modules:
        return sys.modules["rec.io.utils"]
    if not available():
        raise RuntimeError("reference rec.io not available (run oracle/build_ref.sh in the build container)")
    # synthetic packages: rec.io resolves python files in the reference tree and the extension in oracle/_ref; the
    # reference's own rec/io/__init__.py (`from .utils import *`) is bypassed, rec/__init__.py is empty anyway
    rec = types.ModuleType("rec"); rec.__path__ = [os.path.join(REF, "rec")]
    io = types.ModuleType("rec.io"); io.__path__ = [os.path.join(REF, "rec", "io"), os.path.join(_HERE, "_ref")]
    sys.modules.setdefault("rec", rec)
    sys.modules["rec.io"] = io
    import importlib
    return importlib.import_module("rec.io.utils")
