# GPU box: profiles/r6_plugin_path_level1.json - INTEGRATION.md level 1 (adapters only) on configs[1] with its parts timed, level 1.5 beside it,
# and the device time of solve(nrhs) on one block.  usage: bash tools/r6_plugin_profile.sh   (writes gpurun_out/r6_plugin_path_level1.json)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
O=gpurun_out/_plugin; rm -rf $O; mkdir -p $O
timeout 600 python tools/plugin_path_bench.py --level 1 --blocks 4 2>/dev/null | tail -1 > $O/a.json
timeout 600 python tools/plugin_path_bench.py --level 1 --blocks 4 --dense-rhs 2>/dev/null | tail -1 > $O/b.json
timeout 600 python tools/plugin_path_bench.py --level 1 --blocks 4 --chunk 40 2>/dev/null | tail -1 > $O/c.json
timeout 600 python tools/plugin_path_bench.py --level 1.5 --blocks 4 2>/dev/null | tail -1 > $O/d.json
timeout 300 python tools/multi_rhs_probe.py 40 160 2>/dev/null | grep "rhs:" > $O/probe.txt
python - $O <<'PY'
import json, sys
o = sys.argv[1]
out = {"what": "INTEGRATION.md level 1 (adapters only) on BASELINE configs[1], 4 of 64 blocks run and scaled (tools/plugin_path_bench.py --level 1 --blocks 4 "
               "[--chunk 40] [--dense-rhs]); level 1.5 beside it; device time of solve(nrhs) on one block (tools/multi_rhs_probe.py).  The host loop keeps one "
               "dense buffer per leaf like the reference (colsBlockDense, DistributedLinearSystem.C:840-853); the adapters refine adaptively (at most 2 steps, "
               "backward error 1e-15: iparm[7] = 2)",
       "round5": {"level1_chunk160_seconds_per_unit": 16.58, "solve_calls": 10.76, "device_ms_per_solve_160rhs": "~11 (per-right-hand-side sweeps)"},
       "lines": [json.loads(open(f"{o}/{k}.json").read()) for k in "abcd"],
       "multi_rhs_probe": [l.strip() for l in open(f"{o}/probe.txt")]}
json.dump(out, open("gpurun_out/r6_plugin_path_level1.json", "w"), indent=1)
for l in out["lines"]:
    print(l["path"][:60], l["chunk_columns"], round(l["seconds_per_unit"]["total"], 2), {k: round(v, 2) for k, v in l["schur_term_parts_per_unit"].items()})
print("\n".join(out["multi_rhs_probe"]))
PY
