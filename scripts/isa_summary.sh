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
echo "# kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | static LDS B | scratch ops | ds_read | v_pk_fma | v_sqrt | s_barrier | MFMA"
for f in irec_team irec_lone irec_kernels irec_decode; do
  extra=""; [ $f = irec_team -o $f = irec_kernels ] && extra="-mllvm -sink-insts-to-avoid-spills=true"
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
    print(f"{dem} | {field('NumVgprs')} | {field('NumAgprs')} | {field('TotalNumSgprs')} | {field('ScratchSize')} | {field('LDSByteSize')} | "
          f"{cnt(r'scratch_(load|store)')} | {cnt(r'ds_read_b32')} | {cnt(r'v_pk_fma_f32')} | {cnt(r'v_sqrt_f32')} | {cnt(r's_barrier')} | {cnt(r'v_mfma')}")
PY
done
} > "$OUT"
rm -rf "$TMP"
