# round 5: deterministic mode with the sweeps of the augmented factor (Engine::forward_augmented_det) against its refined path (PIPS_HIP_AUG_SWEEPS=0)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/det_aug; mkdir -p $O
cd $R; timeout 1200 python3 -m pytest tests/test_deterministic_gpu.py tests/test_aug_sweeps_gpu.py tests/test_bench_contract_gpu.py -q -m gpu 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
: > $O/ab.jsonl
for v in 1 0; do
  e=""; [ $v = 0 ] && e="PIPS_HIP_AUG_SWEEPS=0"
  env PIPS_HIP_DETERMINISTIC=1 $e python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | grep '^{' >> $O/ab.jsonl
  env PIPS_HIP_DETERMINISTIC=1 $e python3 $R/bench.py --no-cpu-baseline --no-ipm --steps 3 --warmup 1 2>/dev/null | grep '^{' >> $O/ab.jsonl
done
python3 - <<PY
import json
for l in open("$O/ab.jsonl"):
    d=json.loads(l); i=d.get("ipm_end_to_end") or {}
    print(d["config"]["workload"][:40], "| det", d["config"].get("deterministic"), "|", d["ms_per_step"], "| paths", d.get("solve_paths_last_step"), "| step", d["phase_ms"]["step"], "| ipm", i.get("iterations"), i.get("seconds"), i.get("objective"))
PY
