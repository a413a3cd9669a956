# A/B of VAR=1 against the default on one box: bench line, alternating
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=$1
for v in default on default on; do
  if [ $v = on ]; then export $V=1; else unset $V; fi
  python3 $R/bench.py --no-cpu-baseline --no-ipm 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']; print('$V $v', d['value'], d['ms_per_step'], 'factor', r['phase_ms']['total'])"
done
