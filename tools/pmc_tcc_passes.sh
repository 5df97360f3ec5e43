# L2 (TCC) hit / miss / request counters of the ALS half-sweeps and the top-K pass, one rocprofv3 run per pass.
set -e
TAG=${1:-tcc}
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for side in user item; do
  timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/${side}_tcc -o p -- python3 $R/tools/microbench.py sweep --side $side > $O/${side}_tcc.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_ATOMIC_sum --output-format csv -d $O/${side}_ea -o p -- python3 $R/tools/microbench.py sweep --side $side > $O/${side}_ea.log 2>&1 || true
done
echo done
