#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_hip_ops.py -q -x -k "split_resident_operands" 2>&1 | tail -3
RVC_HIP_LIB=comfy-rvc_amd/csrc/variants/librvc_hip_timing.so timeout 300 python tools/time_att.py 2>&1 | tail -3
for x in 1 0; do
rm -rf gpurun_out/att_prof
RVC_X3_XCD=$x timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/att_prof -o att -- python3 bench.py --lanes 1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - $x <<'PY'
import csv,glob,os,sys
f=glob.glob('gpurun_out/att_prof/**/*kernel_stats.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
tot=sum(int(r['TotalDurationNs']) for r in rows)
nclip=int([r for r in rows if 'gru_scan' in r['Name']][0]['Calls'])
print('xcd',sys.argv[1],'ms/clip', round(tot/1e6/nclip,2))
for r in rows:
    if 'attention' in r['Name']: print(r['Name'][:80], int(r['Calls'])/nclip, r['AverageNs'])
PY
done
