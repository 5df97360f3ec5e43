# One capture of everything the committed profile summaries come from (argument: tag, e.g. r01f).
# Runs on the GPU box from the repository root; every GPU step has its own time limit.
set -e
TAG=${1:-prof}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 400 python3 bench.py > $O/bench_line.json 2> $O/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/trace -o r -- python3 bench.py --cpu-seconds 0 > $O/trace.log 2>&1
DB=$(ls $O/trace/*.db $O/trace/*/*.db 2>/dev/null | head -1)
python3 tools/rocpd_stats.py $DB $O/kernel_stats.csv > $O/kernel_stats.txt
cd /tmp
for side in user item; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/${side}_$c -o p -- python3 $R/tools/microbench.py sweep --side $side > $O/${side}_$c.log 2>&1
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/topk_$c -o p -- python3 $R/tools/microbench.py topk --reps 2 --trained 6 --order > $O/topk_$c.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $O/pmc_traffic.json $O/user_FETCH_SIZE $O/user_WRITE_SIZE $O/item_FETCH_SIZE $O/item_WRITE_SIZE $O/topk_FETCH_SIZE $O/topk_WRITE_SIZE
echo done
