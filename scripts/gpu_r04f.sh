#!/bin/bash
# Round 4, step f: the chunked encoder (blocks of more than 1024 dims): parity, then its rate next to the generic kernel's.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${TAG:-r04f}
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_models_shim.py -m gpu -x -q -k "chunk or 1024 or large_block or plan_names or decoder_variants or models or shim or round_trip or compress" > $OUT/pytest_chunk.log 2>&1; rc=$?
tail -15 $OUT/pytest_chunk.log
[ $rc -ne 0 ] && { echo "chunk parity failed rc=$rc"; exit $rc; }
timeout 600 python scripts/generic_rate.py 2>&1 | grep -v amdgpu.ids | tee $OUT/chunk_rate.log
