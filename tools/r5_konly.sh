# round 5: fronts on the rows of K only (PIPS_HIP_MF_KONLY, default on) against the border split of round 4 - tests of everything the head touches, then A/B bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/konly; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_leaf_gpu.py tests/test_aug_sweeps_gpu.py tests/test_configs_gpu.py tests/test_fuzz_gpu.py tests/test_sparse_root_gpu.py tests/test_plugin_batch_gpu.py tests/test_ipm_gpu.py -q -m gpu -k "not configs1_matches" 2>&1 | tail -40 > $O/tests.txt
cat $O/tests.txt
cd /tmp && export TMPDIR=/tmp
: > $O/ab.jsonl
for k in 1 0; do   # (the mode is opt-in: 1 = fronts on the rows of K only, 0 = the default)
  PIPS_HIP_MF_KONLY=$k python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --no-cpu-baseline --steps 6 --warmup 2 2>$O/err_$k.txt | grep '^{' >> $O/ab.jsonl
  PIPS_HIP_MF_KONLY=$k python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --chain-blocks 256 --no-cpu-baseline --no-ipm --steps 6 --warmup 2 2>>$O/err_$k.txt | grep '^{' >> $O/ab.jsonl
done
python3 - <<PY
import json
for l in open("$O/ab.jsonl"):
    d=json.loads(l); i=d.get("ipm_end_to_end") or {}
    print(d["config"]["workload"][:70], "|", d["ms_per_step"], "| leaf_factor", d["phase_ms"]["leaf_factor"], "| ipm", i.get("iterations"), i.get("status"), i.get("objective"), i.get("seconds"))
PY
tail -5 $O/err_1.txt
