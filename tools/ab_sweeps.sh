# A/B of the single-launch tail sweeps against the launch-per-column kernels on one box (run through gpurun)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in rows launches rows launches; do
  if [ $v = rows ]; then unset PIPS_HIP_SWEEP_LAUNCHES; else export PIPS_HIP_SWEEP_LAUNCHES=1; fi
  python3 $R/bench.py --no-cpu-baseline --no-ipm 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v', d['value'], d['ms_per_step'], d['roofline']['phase_ms']['total'])"
done
unset PIPS_HIP_SWEEP_LAUNCHES
rm -rf $R/gpurun_out/qb; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/qb -o qb -- python3 $R/tools/quick_bench.py --reps 6 2>&1 | grep "solve(" | tail -2
python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/qb/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_tail" in r["Name"]: print(r["Name"][:30], r["Calls"], r["AverageNs"])
PY
