#!/bin/bash
# CPU-side sanitizer run (the only one this pool allows): the C oracle built with AddressSanitizer + UBSan, its own tests,
# the importance-sampler tests and the OpenMP batch entry run against it.  Log -> profiles/sanitizer_oracle.log
set -u
cd "$(dirname "$0")/.."
OUT=profiles/sanitizer_oracle.log
TMP=$(mktemp -d)
gcc -O1 -g -mfma -ffp-contract=off -fno-fast-math -fPIC -Wall -Wextra -std=gnu11 -fopenmp -fsanitize=address,undefined \
    -fno-omit-frame-pointer -shared -o $TMP/libirec_oracle.so oracle/irec_oracle.c -lm || exit 1
export IREC_ORACLE_LIB_PATH=$TMP/libirec_oracle.so    # loaded by path: the library in oracle/ is never touched
trap 'rm -rf $TMP' EXIT
{
echo "# $(date -u +%FT%TZ)  gcc $(gcc -dumpversion)  -fsanitize=address,undefined  (oracle/irec_oracle.c)"
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  python -m pytest tests/test_oracle.py tests/test_importance_sampler.py tests/test_oracle_crosscheck.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -8
echo "# OpenMP batch entry"
LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  python - <<'PY' 2>&1 | tail -3
import sys, numpy as np
sys.path.insert(0, ".")
from oracle import oracle as O
lat = [O.synthetic_latent(i, 2048) for i in range(4)]
q = [np.stack([l[j] for l in lat]) for j in range(4)]
idx, samp, used = O.encode_tensors_omp(*q, 42, 3.0, 36, 20, 1000)
ri, rs = O.encode_tensor(*(a[2] for a in q), 42, 3.0, 36, 20, block_size=1000)
print("omp threads", used, "parity", ri == idx[2] and np.array_equal(rs, samp[2]))
PY
} | tee $OUT
