#!/bin/bash
# scratch driver for one-off GPU experiments (edited per experiment; not part of the evidence)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
echo "== mdx tests"
timeout 1200 python -m pytest tests/test_hip_mdx23c.py -x -q 2>&1 | tail -15
echo "== uvr bench"
timeout 900 python bench.py --variant uvr_48k_v2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
echo "== uvr bench, shortcuts unfused"
RVC_MDX_FUSE_SC=0 timeout 900 python bench.py --variant uvr_48k_v2 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
echo "== uvr profile"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_uvr -o uvr -- python3 $GRAFT_REPO_ROOT/bench.py --variant uvr_48k_v2 --steps 2 --warmup 1 --lanes 1 --clips 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/prof_uvr -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/uvr_kernel_stats.csv
rm -rf gpurun_out/prof_uvr
