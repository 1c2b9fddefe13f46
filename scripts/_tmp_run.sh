set -u
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | grep -v amdgpu.ids | tail -3
for L in 38 76 2048; do echo "== LATENTS=$L"; LATENTS=$L REPS=5 python scripts/run_variant.py 2>&1 | tail -2; done
echo "== config3 with 38 images (one GPU's share of 300 over 8)"; timeout 600 python scripts/config3_harness.py --images 38 --no-graph 2>&1 | grep -v amdgpu.ids | tail -1
