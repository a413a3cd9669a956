# bench lines of the other BASELINE configurations (per-GPU shares) -> gpurun_out/other.jsonl
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/other.jsonl
: > $O
cd /tmp && export TMPDIR=/tmp
run() { python3 $R/bench.py --no-cpu-baseline "$@" 2>/dev/null | grep '^{' >> $O; }
run --blocks-per-gpu 4 --n 1000 --schur-dim 200 --rho 0.01 --steps 20 --warmup 3
run --blocks-per-gpu 64 --n 10000 --schur-dim 4000 --rho 0.001 --steps 3 --warmup 1
run --blocks-per-gpu 32 --n 2000 --schur-dim 16000 --rho 0.005 --steps 3 --warmup 1
# BASELINE configs[4] as a whole on one GPU (256 x 2000, S = 16 000: it names no GPU count and fits)
run --blocks-per-gpu 256 --n 2000 --schur-dim 16000 --rho 0.005 --steps 3 --warmup 1 --no-ipm
run --blocks-per-gpu 256 --n 2000 --schur-dim 4000 --rho 0.005 --steps 3 --warmup 1
PIPS_HIP_DETERMINISTIC=1 python3 $R/bench.py --no-cpu-baseline --no-ipm --steps 3 --warmup 1 2>/dev/null | grep '^{' >> $O
PIPS_HIP_DETERMINISTIC=1 python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --no-cpu-baseline --no-ipm --steps 3 --warmup 1 2>/dev/null | grep '^{' >> $O
# round 5: the 256-block chain rounds 3-4 measured (31 linking rows per pair), for continuity with their numbers; and without the per-solve measure
run --family time-coupled --blocks-per-gpu 256 --n 50000 --chain-blocks 256 --no-ipm --steps 6 --warmup 2
run --family time-coupled --blocks-per-gpu 256 --n 50000 --chain-blocks 256 --no-ipm --steps 6 --warmup 2 --solve-check-every 0
run --family time-coupled --blocks-per-gpu 256 --n 50000 --no-ipm --steps 6 --warmup 2 --solve-check-every 0
python3 - <<PY
import json
for l in open("$O"):
    d=json.loads(l); r=d["roofline"]; i=d.get("ipm_end_to_end") or {}
    ph=d["phase_ms"]
    print(d["config"]["workload"][:100], "|", d["ms_per_step"], d["value"], "|", r["group"], "frac", r["frac"], "| accounted", ph["accounted"], "of instrumented step", ph["instrumented_step_wall"], {k: v for k, v in ph["step"].items()}, "| ipm", i.get("iterations"), round(i.get("seconds",0),2), i.get("status"))
PY
# round 4: the largest point of the SURVEY 8d random generator (10 non-zeros per row) that 256 blocks per GPU reach in 288 GB
# (tools/config3_random_limit.py: n_i = 10 000 fits, 15 000 does not), S = 8000 as configs[3] has it
python3 $R/bench.py --family random --blocks-per-gpu 256 --n 10000 --schur-dim 8000 --rho 0.001 --steps 2 --warmup 1 --no-cpu-baseline --no-ipm 2>$R/gpurun_out/other_random256.err | grep '^{' > $R/gpurun_out/other_random256.json
python3 - <<PY
import json
try:
    d=json.load(open("$R/gpurun_out/other_random256.json")); print("random 256 x 10000, S = 8000:", d["ms_per_step"], d["value"], d["roofline"]["group"], d["roofline"]["frac"], d["phase_ms"]["step"])
except Exception as e:
    print("random 256 x 10000 failed:", e); print(open("$R/gpurun_out/other_random256.err").read()[-800:])
PY
