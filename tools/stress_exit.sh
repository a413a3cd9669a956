#!/bin/bash
# GPU box: does a bench process ever fail to END?  Many short runs of the time-coupled family through the interpreter's normal exit
# (PIPS_BENCH_NORMAL_EXIT=1: no os._exit) - default streams, and the sparse root factorised on a stream of its own
# (PIPS_HIP_SPARSE_ROOT_ASYNC=1) - each under bench.py's watchdog (all Python stacks on stderr after 90 s) and an outer timeout.
# usage: stress_exit.sh <runs per variant> [blocks] [n] [watchdog seconds]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-60}; NB=${2:-16}; NI=${3:-20000}; WD=${4:-90}
O=$R/gpurun_out/stress_exit
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for variant in default async; do
  ok=0; bad=0; t0=$(date +%s)
  for i in $(seq 1 $N); do
    if [ $variant = async ]; then export PIPS_HIP_SPARSE_ROOT_ASYNC=1; else unset PIPS_HIP_SPARSE_ROOT_ASYNC; fi
    PIPS_BENCH_NORMAL_EXIT=1 PIPS_BENCH_WATCHDOG=$WD timeout -k 5 $((WD + 60)) python3 $R/bench.py --family time-coupled --blocks-per-gpu $NB --n $NI --chain-blocks 256 \
       --steps 6 --warmup 2 --no-cpu-baseline > $O/${variant}_$i.json 2> $O/${variant}_$i.err
    rc=$?
    if [ $rc -eq 0 ] && grep -q '^{' $O/${variant}_$i.json; then ok=$((ok+1)); rm -f $O/${variant}_$i.json $O/${variant}_$i.err; else bad=$((bad+1)); echo "$variant run $i: exit $rc"; tail -30 $O/${variant}_$i.err; fi
  done
  echo "$variant: $ok of $N runs ended normally, $bad did not; $(( $(date +%s) - t0 )) s" | tee -a $O/summary.txt
done
