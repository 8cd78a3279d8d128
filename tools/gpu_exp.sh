#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence):  gpurun --timeout 900 -- 'bash tools/gpu_exp.sh > gpurun_out/expN.txt 2>&1'
# (this version: shader clock and package power polled beside a long run of the headline bench - profiles/r4zz_power_clocks.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
poll() { for i in $(seq 1 $1); do /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)\|Socket" | sed 's/ \+/ /g; s/GPU\[0\] : //g' | tr '\n' ' '; echo; sleep 2; done; }
timeout 300 python bench.py --steps 300 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/b.txt 2>&1 &
sleep 15; poll 24; wait
tail -1 gpurun_out/b.txt | cut -c1-160
