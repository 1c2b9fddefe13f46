set -u
export TMPDIR=/tmp
mkdir -p gpurun_out
C=$PWD/relative-entropy-coding_amd/csrc
{
echo "== stamps (default = 3 teams)"; IREC_LIB_PATH=$C/variants/stamps.so IREC_STAMPS=1 LATENTS=2048 REPS=2 timeout 300 python scripts/run_variant.py 2>&1 | tail -13
for v in SCORING SELECT UPDATE; do echo "== without $v"; IREC_LIB_PATH=$C/variants/ablate_$v.so LATENTS=2048 REPS=3 timeout 300 python scripts/run_variant.py 2>&1 | tail -1; done
echo "== full"; LATENTS=2048 REPS=3 timeout 300 python scripts/run_variant.py 2>&1 | tail -1
} 2>&1 | tee gpurun_out/stamps_r02b.log
