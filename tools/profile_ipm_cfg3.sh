#!/bin/bash
# GPU box: kernel statistics of the end-to-end IPM run on the time-coupled share (the bench line's ipm_end_to_end leg dominates the process)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ipmprof; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o ipm -- python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --steps 1 --warmup 0 --no-cpu-baseline > $O/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f))); rows.sort(key=lambda r:-float(r['TotalDurationNs']))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:45]: print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms avg {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
tail -c 400 $O/log.txt
find $O -name "*kernel_trace.csv" -delete
