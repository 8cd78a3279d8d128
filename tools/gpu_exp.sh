#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence):  gpurun --timeout 900 -- 'bash tools/gpu_exp.sh > gpurun_out/expN.txt 2>&1'
# (this version: shader clock and package power of the microbenchmark's instruction mixes, each sustained for 8 s - profiles/r4zz_power_clocks.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
poll() { for i in $(seq 1 $1); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)\|Socket" | sed 's/ \+/ /g; s/GPU\[0\] : //g' | tr '\n' ' '; echo; sleep 1; done; }
for m in 0 2 6 8 12 10; do
  timeout 60 tools/micro/mfmabench sustain $m 8 > gpurun_out/mm.txt 2>&1 &
  sleep 3; poll 3; wait
  cut -c1-150 gpurun_out/mm.txt
done
