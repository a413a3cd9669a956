R=${GRAFT_REPO_ROOT:-/root/repo}
for rs in 0 1 0 1; do
  if [ "$rs" = "1" ]; then export PIPS_HIP_ROOT_SYNC=1; else unset PIPS_HIP_ROOT_SYNC; fi
  timeout 300 python3 $R/bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); i=d['ipm_end_to_end']; print('rootsync $rs step', d['ms_per_step'], 'ipm', i['seconds'], i['iterations'])"
done
