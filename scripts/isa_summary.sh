#!/bin/bash
# Per-kernel register / scratch / LDS / instruction-mix summary of the HIP sources, from hipcc's own assembly output.
# Usage: scripts/isa_summary.sh [out.txt]   (no GPU needed; cross-compiles for gfx950)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/dev/stdout}
TMP=$(mktemp -d)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -I$ROOT/include -I$ROOT/relative-entropy-coding_amd/csrc --cuda-device-only -S"
{
echo "# ISA summary of the gfx950 kernels (hipcc $(hipcc --version | grep -m1 -o 'HIP version: [0-9.]*'), flags of csrc/Makefile)"
echo "# kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | static LDS B | scratch ops | ds_read | v_pk_fma | v_sqrt | s_barrier | MFMA | scoring loops (innermost loops holding >= 40 v_pk_fma_f32: lines / ds_read_b32 / scratch ops)"
echo "# (round 5, last column: the loop of 80 ds_read_b32 and 0 scratch operations is the software-pipelined steady state that every full step of a 20-beam build"
echo "#  runs -- 20 beams x 4 dim slots per sample; the loops listed before it are the beam-by-beam fall-back of partly filled stripes and the first step's"
echo "#  wide path: the headline build's 588 B/lane of scratch are outside the steady state)"
for f in irec_team irec_team_margin irec_ten irec_lone irec_kernels irec_decode; do
  extra=""; [ $f = irec_team -o $f = irec_team_margin -o $f = irec_ten -o $f = irec_kernels ] && extra="-mllvm -sink-insts-to-avoid-spills=true"
  hipcc $FLAGS $extra "$ROOT/relative-entropy-coding_amd/csrc/$f.hip" -o "$TMP/$f.s" 2>/dev/null
  python3 - "$TMP/$f.s" <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
# kernel bodies: from "name:" to the matching ".end_amdhsa_kernel"-less "s_endpgm ... .Lfunc_end"
for m in re.finditer(r'^(_Z\w+):\s*; @\1\n(.*?)^\.Lfunc_end\d+:', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    tail = txt[m.end():m.end() + 6000]
    def field(k):
        mm = re.search(r'; %s: (\d+)' % k, tail)
        return mm.group(1) if mm else '?'
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r'\(irec::\w+\)$', '', dem).replace('void irec::', '')
    cnt = lambda pat: len(re.findall(pat, body))
    # innermost loops (label .. backward branch to it) that hold the scoring arithmetic
    bl = body.split('\n')
    labels = {mm.group(1): i for i, l in enumerate(bl) for mm in [re.match(r'^(\.LBB\d+_\d+):', l)] if mm}
    loops = []
    for i, l in enumerate(bl):
        mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i))
    def c2(a, b, pat): return sum(1 for l in bl[a:b] if re.search(pat, l))
    hot = [(a, b) for a, b in loops if c2(a, b, 'v_pk_fma_f32') >= 40]
    inner = sorted({(a, b) for a, b in hot if not any((a2, b2) != (a, b) and a <= a2 and b2 <= b for a2, b2 in hot)})
    # (several back edges of one loop body: keep the widest range per header)
    byhead = {}
    for a, b in inner: byhead[a] = max(byhead.get(a, b), b)
    seen, keep = [], []
    for a, b in sorted(byhead.items()):
        if not any(a2 <= a and b <= b2 + 40 for a2, b2 in keep): keep.append((a, b))
    loops_s = ", ".join(f"{b - a}L/{c2(a, b, 'ds_read_b32')}r/{c2(a, b, 'scratch_')}s" for a, b in keep) or "-"
    print(f"{dem} | {field('NumVgprs')} | {field('NumAgprs')} | {field('TotalNumSgprs')} | {field('ScratchSize')} | {field('LDSByteSize')} | "
          f"{cnt(r'scratch_(load|store)')} | {cnt(r'ds_read_b32')} | {cnt(r'v_pk_fma_f32')} | {cnt(r'v_sqrt_f32')} | {cnt(r's_barrier')} | {cnt(r'v_mfma')} | {loops_s}")
PY
done
} > "$OUT"
rm -rf "$TMP"
