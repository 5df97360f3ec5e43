# Vector-L1 (TCP) and texture-addresser (TA) counters of one half-sweep side or the top-K pass, one rocprofv3 run per
# pass (arguments: user|item|topk, output tag).
set -e
WHAT=${1:-user}; TAG=${2:-tcp}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
if [ "$WHAT" = topk ]; then CMD="python3 $R/tools/microbench.py topk --reps 2"; else CMD="python3 $R/tools/microbench.py sweep --side $WHAT"; fi
P1="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
P2="TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
i=1
for P in "$P1" "$P2"; do
  timeout -s KILL 120 rocprofv3 --pmc $P --output-format csv -d $O/${WHAT}_pass$i -o p -- $CMD > $O/${WHAT}_pass$i.log 2>&1
  i=$((i+1))
done
echo done
