# A/B of an alternative library build (tools/ab/lib<NAME>.so) against the default one: bench line + dense root factor times
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in default $1 default $1; do
  if [ $v = default ]; then unset PIPS_HIP_LIBRARY; else export PIPS_HIP_LIBRARY=$R/tools/ab/lib$v.so; fi
  python3 $R/bench.py --no-cpu-baseline --no-ipm 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); r=d['roofline']; print('$v', d['value'], d['ms_per_step'], 'frac', r['frac'], r['phase_ms'])"
  python3 $R/tools/root_probe.py 4000 16000 2>/dev/null | grep "S="
done
