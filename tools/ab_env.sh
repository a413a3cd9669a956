# A/B of one environment switch on one box: tools/ab_env.sh VAR  (bench.py without the CPU baseline and the IPM leg, alternating)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=$1
for v in on off on off; do
  if [ $v = on ]; then export $V=1; else unset $V; fi
  python3 $R/bench.py --no-cpu-baseline --no-ipm 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']; print('$V $v', d['value'], d['ms_per_step'], 'frac', r['frac'], r['phase_ms'])"
done
