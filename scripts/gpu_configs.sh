#!/bin/bash
# Timing of the other BASELINE.json configurations' coder settings on RVAE-shaped latents (diagnostic).
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== cfg2 B=20 O=3 e=1.2"; LATENTS=1024 python scripts/run_variant.py | tail -1
echo "== cfg4 B=10 O=3 e=1.0 (lossy settings)"; BEAMS=10 OMEGA=3.0 EPS1=1.0 LATENTS=1024 python scripts/run_variant.py | tail -1
echo "== cfg5 B=30 O=5 e=1.0 (stress S=148)"; BEAMS=30 OMEGA=5.0 EPS1=1.0 LATENTS=256 python scripts/run_variant.py | tail -1
echo "== cfg5b B=30 O=5 e=1.2 (S=403)"; BEAMS=30 OMEGA=5.0 EPS1=1.2 LATENTS=64 python scripts/run_variant.py | tail -1
echo "== B=20 O=5 e=1.0 (S=148: one striped team, sample passes)"; BEAMS=20 OMEGA=5.0 EPS1=1.0 LATENTS=256 python scripts/run_variant.py | tail -1
echo "== B=20 O=5 e=1.0 one-table (fused Philox)"; IREC_VARIANT=fused BEAMS=20 OMEGA=5.0 EPS1=1.0 LATENTS=256 python scripts/run_variant.py | tail -1
echo "== cfg5 fused"; IREC_VARIANT=fused BEAMS=30 OMEGA=5.0 EPS1=1.0 LATENTS=256 python scripts/run_variant.py | tail -1
} 2>&1 | tee gpurun_out/configs.log
