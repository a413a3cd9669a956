R=${GRAFT_REPO_ROOT:-/root/repo}
for p in 2 3 4 6; do
  PIPS_HIP_ROOT_PANEL=$p timeout 300 python3 $R/bench.py --blocks-per-gpu 32 --n 2000 --schur-dim 16000 --rho 0.005 --steps 4 --warmup 1 --no-cpu-baseline --no-ipm 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['phase_ms']['step']; print('panel $p step', d['ms_per_step'], 'root_factor', s['root_factor'], 'exposed', d['phase_ms'].get('root_factor_exposed'))"
done
