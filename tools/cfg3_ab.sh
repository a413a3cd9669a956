#!/bin/bash
# GPU box: A/B runs of the time-coupled share under environment settings.  usage: cfg3_ab.sh "<tag>:<VAR=val,VAR=val>" ...   (prints head / factor / step ms)
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="--family time-coupled --blocks-per-gpu 256 --n 50000 --no-ipm --no-cpu-baseline --steps 5 --warmup 1"
for spec in "$@"; do
   tag=${spec%%:*}; envs=${spec#*:}
   ( for kv in ${envs//,/ }; do [ -n "$kv" ] && export "$kv"; done
     python3 $R/bench.py $ARGS 2> $R/gpurun_out/ab_$tag.err | tail -1 > $R/gpurun_out/ab_$tag.json )
   python3 - $R/gpurun_out/ab_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    lf, st = d["phase_ms"]["leaf_factor"], d["phase_ms"]["step"]
    print(f"{sys.argv[2]:>16}: step {d['ms_per_step']:.2f} ms, leaf_factor {st['leaf_factor']:.2f}, head {lf['head']:.2f}, scatter {lf['scatter']:.2f}, lsolve {st['lsolve_leaf']:.2f}, ltsolve {st['ltsolve']:.2f}, nnzL {d['config']['nnzL_per_gpu']}")
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
