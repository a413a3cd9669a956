#!/bin/bash
# GPU box: repeated bench runs, each under its own timeout - looking for runs that do not finish
R=${GRAFT_REPO_ROOT:-/root/repo}
n_ok=0; n_bad=0
run() { local tag=$1; shift; timeout 300 "$@" > $R/gpurun_out/stress_$tag.json 2> $R/gpurun_out/stress_$tag.err; local rc=$?; if [ $rc -eq 0 ] && tail -1 $R/gpurun_out/stress_$tag.json | grep -q '^{'; then n_ok=$((n_ok+1)); else n_bad=$((n_bad+1)); echo "RUN $tag exit $rc"; tail -3 $R/gpurun_out/stress_$tag.err; fi; }
for i in 1 2 3 4 5 6; do run c3_$i python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --steps 6 --warmup 2 --no-cpu-baseline; done
for i in 1 2 3; do PIPS_HIP_AUG_WITNESS=0 run c3w0_$i python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --steps 6 --warmup 2 --no-cpu-baseline; done
for i in 1 2 3; do run c1_$i python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 2; done
echo "finished: $n_ok ok, $n_bad not"
