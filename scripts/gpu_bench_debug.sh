#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
nproc; python -c "import os; print(os.cpu_count(), len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for L in 64 512; do
  echo "== bench L=$L no cpu"
  timeout 300 python bench.py --steps 2 --warmup 1 --latents $L --no-cpu-baseline 2>&1 | tail -4 | tee gpurun_out/bench_L$L.log
done
echo "== bench with cpu baseline"
timeout 600 python bench.py --steps 2 --warmup 1 --latents 512 --cpu-ref-latents 2 --cpu-opt-latents 8 2>&1 | tail -8 | tee gpurun_out/bench_cpu.log
