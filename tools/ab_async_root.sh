#!/bin/bash
# GPU box: the sparse root on the main stream (default) against a stream of its own (PIPS_HIP_SPARSE_ROOT_ASYNC=1), alternating on one box:
# step time of the factorise / solve loop and the end-to-end IPM.  usage: ab_async_root.sh [chain blocks]
R=${GRAFT_REPO_ROOT:-/root/repo}
CH=${1:-2048}
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
  for v in 0 1; do
    if [ $v = 1 ]; then export PIPS_HIP_SPARSE_ROOT_ASYNC=1; else unset PIPS_HIP_SPARSE_ROOT_ASYNC; fi
    timeout 300 python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --chain-blocks $CH --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); i=d['ipm_end_to_end']
print('chain $CH async $v: step', d['ms_per_step'], 'ms; IPM', i['seconds'], 's', i['iterations'], 'iterations; root_wait', d['phase_ms']['step']['root_wait'], 'root on main', d['phase_ms']['step']['root_factor_main_stream'])"
  done
done
