#!/bin/bash
# GPU box: many full-size runs with the sparse root factorised on a stream of its own (the configuration of which round 4 saw one run in
# about two dozen not finish), each through the normal exit under bench.py's watchdog.  usage: stress_async.sh <runs> [chain blocks]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-100}; CH=${2:-256}
O=$R/gpurun_out/stress_async
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ok=0; bad=0; t0=$(date +%s)
for i in $(seq 1 $N); do
  PIPS_HIP_SPARSE_ROOT_ASYNC=1 PIPS_BENCH_WATCHDOG=200 timeout -k 5 260 python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --chain-blocks $CH \
     --steps 6 --warmup 2 --no-cpu-baseline > $O/run_$i.json 2> $O/run_$i.err
  rc=$?
  if [ $rc -eq 0 ] && grep -q '^{' $O/run_$i.json; then ok=$((ok+1)); rm -f $O/run_$i.json $O/run_$i.err; else bad=$((bad+1)); echo "run $i: exit $rc"; tail -30 $O/run_$i.err; fi
done
echo "async root, chain $CH: $ok of $N runs ended normally, $bad did not; $(( $(date +%s) - t0 )) s" | tee -a $O/summary.txt
