#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence):  gpurun --timeout 900 -- 'bash tools/gpu_exp.sh > gpurun_out/expN.txt 2>&1'
# (this version, round 5: the persistent ResBlock kernel with c1 reading an image as well - RVC_X3Q_BOTH=1 - against the shipped pair; profiles/r5_x3q_image_in.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
for c in 128 64; do
  for rep in 1 2; do
    for both in 0 1; do echo "== PAIR_C=$c RVC_X3Q_BOTH=$both (run $rep)"; PAIR_C=$c RVC_X3Q_BOTH=$both timeout 200 python tools/bench_pair.py 2>&1 | grep split; done
  done
done
