cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/cfg4prof; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o c4 -- python3 $R/bench.py --blocks-per-gpu 32 --n 2000 --schur-dim 16000 --rho 0.005 --steps 3 --warmup 1 --no-cpu-baseline --no-ipm > $O/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:22]: print(f"{r['Name'][:80]:80s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:8.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us")
PY
find $O -name "*kernel_trace.csv" -delete
