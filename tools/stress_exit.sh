#!/bin/bash
# GPU box: does a bench process ever fail to END?  Many short runs of the time-coupled family through the interpreter's normal exit (the
# only exit bench.py has since round 5) - the sparse root factorised on the main stream (PIPS_HIP_SPARSE_ROOT_ASYNC=0: "sync"), and on a
# stream of its own (the default since round 5: "async") - each under bench.py's watchdog (all Python stacks on stderr after 90 s) and
# an outer timeout.  (profiles/r5_stress_exit.txt was made before the default flipped: its "default" rows are the main-stream root.)
# usage: stress_exit.sh <runs per variant> [blocks] [n] [watchdog seconds]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-60}; NB=${2:-16}; NI=${3:-20000}; WD=${4:-90}
O=$R/gpurun_out/stress_exit
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for variant in sync async; do
  ok=0; bad=0; t0=$(date +%s)
  for i in $(seq 1 $N); do
    if [ $variant = async ]; then unset PIPS_HIP_SPARSE_ROOT_ASYNC; else export PIPS_HIP_SPARSE_ROOT_ASYNC=0; fi
    PIPS_BENCH_WATCHDOG=$WD timeout -k 5 $((WD + 60)) python3 $R/bench.py --family time-coupled --blocks-per-gpu $NB --n $NI --chain-blocks 256 \
       --steps 6 --warmup 2 --no-cpu-baseline > $O/${variant}_$i.json 2> $O/${variant}_$i.err
    rc=$?
    if [ $rc -eq 0 ] && grep -q '^{' $O/${variant}_$i.json; then ok=$((ok+1)); rm -f $O/${variant}_$i.json $O/${variant}_$i.err; else bad=$((bad+1)); echo "$variant run $i: exit $rc"; tail -30 $O/${variant}_$i.err; fi
  done
  echo "$variant: $ok of $N runs ended normally, $bad did not; $(( $(date +%s) - t0 )) s" | tee -a $O/summary.txt
done
