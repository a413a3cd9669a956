# bench lines of the other BASELINE configurations (per-GPU shares) -> gpurun_out/other.jsonl
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/other.jsonl
: > $O
cd /tmp && export TMPDIR=/tmp
run() { python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' >> $O; }
run --blocks-per-gpu 4 --n 1000 --schur-dim 200 --rho 0.01 --steps 20 --warmup 3
run --blocks-per-gpu 64 --n 10000 --schur-dim 4000 --rho 0.001 --steps 3 --warmup 1
run --blocks-per-gpu 32 --n 2000 --schur-dim 16000 --rho 0.005 --steps 3 --warmup 1
run --blocks-per-gpu 256 --n 2000 --schur-dim 4000 --rho 0.005 --steps 3 --warmup 1
PIPS_HIP_DETERMINISTIC=1 python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | grep '^{' >> $O
PIPS_HIP_DETERMINISTIC=1 python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --no-cpu-baseline --no-ipm --steps 3 --warmup 1 2>/dev/null | grep '^{' >> $O
python3 - <<PY
import json
for l in open("$O"):
    d=json.loads(l); r=d["roofline"]; i=d.get("ipm_end_to_end") or {}
    print(d["config"]["workload"][:100], "|", d["ms_per_step"], d["value"], "|", r["group"], "frac", r["frac"], {k: v for k, v in d["phase_ms"]["step"].items()}, "| ipm", i.get("iterations"), round(i.get("seconds",0),2), i.get("status"))
PY
