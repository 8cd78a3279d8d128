# SQ / L2 counters of one kernel under tools/run_ops.py (separate rocprofv3 passes of 8 counters each, --pmc only):
#   bash tools/pmc_kernels.sh <run_ops argument> <kernel name substring> <tag>   -> gpurun_out/pmck_<tag>.txt
what="$1"; kern="$2"; tag="$3"
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" \
 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" ; do
  i=$((i+1)); d=gpurun_out/pmck_${tag}_$i; rm -rf $d
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $d -o p -- python3 tools/run_ops.py "$what" > $d.out 2>&1
done
PMC_KERNELS="$kern" python3 tools/pmc_conv_summary.py gpurun_out/pmck_${tag}_ > gpurun_out/pmck_${tag}.txt
cat gpurun_out/pmck_${tag}.txt
find gpurun_out -path "*pmck_${tag}_*" -name "*.csv" -size +2M -delete
