# SQ counter passes over the all-users top-K pass, one rocprofv3 run per pass (arguments: output tag, then
# options for tools/microbench.py topk, e.g. --trained 6 --order).
set -e
TAG=${1:-topk_sq}
shift || true
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM"
P3="SQ_INSTS_MFMA SQ_ACTIVE_INST_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAVES GRBM_GUI_ACTIVE GRBM_COUNT"
i=1
for P in "$P1" "$P2" "$P3"; do
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $O/pass$i -o p -- python3 $R/tools/microbench.py topk --reps 2 "$@" > $O/pass$i.log 2>&1
  i=$((i+1))
done
echo done
