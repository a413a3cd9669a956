# scan of the head / tail cut cost model on config 2 (bench line per setting)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for hc in 8.0e-11 4.0e-11 1.6e-10 3.2e-10 8.0e-11; do
  PIPS_HIP_HEAD_COST=$hc python3 $R/bench.py --no-cpu-baseline --no-ipm 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']; print('head_cost $hc', d['value'], d['ms_per_step'], 'm', d['config']['tail_dim_avg'], r['phase_ms'])"
done
