#!/bin/bash
# AddressSanitizer + UBSan over the host-only part of the C ABI (csrc/irec_io.cpp: arithmetic coder, .rec containers, the
# batched container calls on host threads): the container tests -- golden files, random structures, damaged files -- run
# against a sanitized build of that one source file (GPU sanitizers are not available on this pool; this code never touches
# the GPU).  The product library is not touched: the build goes to a temporary directory and is loaded by path (scripts/with_lib.py).
# Usage: scripts/sanitize_io.sh [log]      (CPU only)
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$ROOT/profiles/sanitizer_io.log}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
g++ -O1 -g -std=c++17 -shared -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -I"$ROOT/include" "$ROOT/relative-entropy-coding_amd/csrc/irec_io.cpp" -o "$TMP/libirec_io_asan.so" -pthread
{
  echo "# $(g++ --version | head -1); -fsanitize=address,undefined -fno-sanitize-recover=undefined; $(date -u +%F)"
  cd "$ROOT"
  # (test_arithmetic_coder_rejects... also calls irec_n_samples, which lives in irec_host.cpp: its coder part is covered by the
  #  hostile-model cases of the container fuzz below and by the golden coder vectors)
  LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 \
    python scripts/with_lib.py "$TMP/libirec_io_asan.so" -m pytest tests/test_rec_io.py -q -m "not gpu" -k "not arithmetic_coder_rejects" 2>&1 | tail -4
} | tee "$LOG"
