#!/bin/bash
# GPU box: the single-launch root under repetition - the configs[4] share (S = 16 000, the root on its own stream beside the first Lsolve),
# configs[1] with its end-to-end IPM (48 root factorisations each) and the root alone, every run under its own timeout.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
n_ok=0; n_bad=0
run() { local tag=$1; shift; timeout 300 "$@" > $R/gpurun_out/stress_$tag.json 2> $R/gpurun_out/stress_$tag.err; local rc=$?; if [ $rc -eq 0 ] && tail -1 $R/gpurun_out/stress_$tag.json | grep -q "^{\|^S="; then n_ok=$((n_ok+1)); else n_bad=$((n_bad+1)); echo "RUN $tag exit $rc"; tail -3 $R/gpurun_out/stress_$tag.err; fi; }
for i in 1 2 3 4 5 6 7 8; do run c4_$i python3 bench.py --blocks-per-gpu 32 --n 2000 --schur-dim 16000 --rho 0.005 --steps 5 --warmup 1 --no-cpu-baseline --no-ipm; done
for i in 1 2 3 4; do run c1_$i python3 bench.py --no-cpu-baseline --steps 3 --warmup 1; done
for i in 1 2 3 4 5 6; do run root_$i python3 tools/root_probe.py 16000; done
for i in 1 2 3 4; do run c0_$i python3 bench.py --blocks-per-gpu 4 --n 1000 --schur-dim 200 --rho 0.01 --steps 200 --warmup 3 --no-cpu-baseline; done
echo "finished: $n_ok ok, $n_bad not"
