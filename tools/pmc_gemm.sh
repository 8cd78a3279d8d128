# SQ counters of the split-resident GEMM kernel under tools/bench_gemm.py:   bash tools/pmc_gemm.sh <shape filter> <tag> [ENV=VAL ...]  -> gpurun_out/pmcgemm_<tag>.txt
sel="$1"; tag="$2"; shift 2
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do export "$v"; done
export BENCH_QUICK=1
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" ; do
  i=$((i+1)); d=gpurun_out/pmcgemm_${tag}_$i; rm -rf $d
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $d -o p -- python3 tools/bench_gemm.py "$sel" > $d.out 2>&1
done
python3 tools/pmc_conv_summary.py gpurun_out/pmcgemm_${tag}_ > gpurun_out/pmcgemm_${tag}.txt
cat gpurun_out/pmcgemm_${tag}.txt
find gpurun_out -path "*pmcgemm_${tag}_*" -name "*.csv" -size +2M -delete
