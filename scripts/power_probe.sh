#!/bin/bash
# power, clocks and temperature of the card while a kernel runs in a loop (diagnostics): RUN = environment of scripts/run_variant.py
(env ${RUN:-LATENTS=8192 BEAMS=20 EPS1=1.2} IREC_VARIANT=auto REPS=${REPS:-700} timeout -k 10 300 python scripts/run_variant.py > gpurun_out/power_run.log 2>&1) &
PID=$!
for i in $(seq 1 ${SAMPLES:-25}); do
  sleep 2
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "Package Power\|sclk\|junction" | sed 's/.*: //' | tr "\n" " "; echo
done
wait $PID
tail -1 gpurun_out/power_run.log
