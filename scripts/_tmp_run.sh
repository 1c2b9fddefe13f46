set -u
export TMPDIR=/tmp
C=$PWD/relative-entropy-coding_amd/csrc
for lib in libirec_hip.so variants/exp_A.so variants/exp_B.so; do
echo "== $lib B=20"; IREC_LIB_PATH=$C/$lib LATENTS=2048 REPS=5 timeout 300 python scripts/run_variant.py 2>&1 | tail -2
echo "== $lib B=10"; IREC_LIB_PATH=$C/$lib BEAMS=10 OMEGA=3.0 EPS1=1.0 LATENTS=2048 REPS=5 python scripts/run_variant.py | tail -2
done
echo "== B=10 two teams"; SHAPE=2 BEAMS=10 OMEGA=3.0 EPS1=1.0 LATENTS=2048 REPS=5 python scripts/run_variant.py | tail -1
IREC_LIB_PATH=$C/variants/exp_B.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "golden or full_size or config4 or ragged" 2>&1 | grep -v amdgpu.ids | tail -2
