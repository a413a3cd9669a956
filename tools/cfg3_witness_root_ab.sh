#!/bin/bash
# GPU box: the time-coupled share under (witness kind) x (root stream) - step time and the end-to-end IPM; every run under its own timeout
R=${GRAFT_REPO_ROOT:-/root/repo}
for cfg in ${@:-"0:0" "1:0" "0:1" "1:1"}; do
  w=${cfg%%:*}; rs=${cfg#*:}
  export PIPS_HIP_AUG_WITNESS=$w
  if [ "$rs" = "1" ]; then export PIPS_HIP_ROOT_SYNC=1; else unset PIPS_HIP_ROOT_SYNC; fi
  timeout 240 python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/wr_$w$rs.json 2> $R/gpurun_out/wr_$w$rs.err
  echo "witness $w rootsync $rs exit $?"
  tail -1 $R/gpurun_out/wr_$w$rs.json | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); i=d['ipm_end_to_end']; s=d['phase_ms']['step']; print('   ipm s', i['seconds'], 'step ms', d['ms_per_step'], 'lsolve', s['lsolve_leaf'], 'dsolve', s['dsolve'], 'ltsolve', s['ltsolve'], 'root', s['root_factor'])
except Exception as e: print('   no line:', e)"
  tail -3 $R/gpurun_out/wr_$w$rs.err
done
