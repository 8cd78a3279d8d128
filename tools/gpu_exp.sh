#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_models.py tests/test_hip_pipeline.py -q -x -k "synth or pipeline or c3 or c4" 2>&1 | tail -3
for v in 2 0; do
  rm -rf gpurun_out/att_prof
  RVC_X3_WIDE_K3=$v timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/att_prof -o att -- python3 bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - $v <<'PY'
import csv,glob,sys
rows=list(csv.DictReader(open(glob.glob('gpurun_out/att_prof/**/*kernel_stats.csv',recursive=True)[0])))
n=int([r for r in rows if 'gru_scan' in r['Name']][0]['Calls'])
tot=sum(int(r['TotalDurationNs']) for r in rows)/1e6/n
k3=sum(int(r['TotalDurationNs']) for r in rows if 'conv_x3p_kernel<2, 2, 3' in r['Name'] or 'conv_x3p_kernel<2, 4, 3' in r['Name'])/1e6/n
gen=sum(int(r['TotalDurationNs']) for r in rows if 'conv_x3p' in r['Name'])/1e6/n
print('wide_k3', sys.argv[1], 'ms/clip', round(tot,2), 'k3 classes ms', round(k3,3), 'x3p family ms', round(gen,3))
PY
done
