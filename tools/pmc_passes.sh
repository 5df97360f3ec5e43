set -e
R=/root/repo; O=$R/gpurun_out/pmc_r01c; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for side in user item; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/${side}_$c -o p -- python3 $R/tools/microbench.py sweep --side $side > $O/${side}_$c.log 2>&1
  done
done
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $O/topk_$c -o p -- python3 $R/tools/microbench.py topk --reps 2 > $O/topk_$c.log 2>&1
done
echo done
