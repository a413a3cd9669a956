R=${GRAFT_REPO_ROOT:-/root/repo}
export PIPS_HIP_SPARSE_ROOT_ASYNC=1 PIPS_HIP_AUG_WITNESS=0
for i in 1 2 3 4 5 6 7 8; do
  timeout 200 python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/sa_$i.json 2> $R/gpurun_out/sa_$i.err; echo "run $i exit $?"
done
