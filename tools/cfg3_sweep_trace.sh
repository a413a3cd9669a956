#!/bin/bash
# GPU box: per-launch durations of the head sweep kernels of one solveCompressed on the time-coupled share (kernel trace, the last forward sweep of the augmented factor)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sweeptrace; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O -o sw -- python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --steps 1 --warmup 1 --no-cpu-baseline --no-ipm > $O/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def short(n): return n.replace('void pips::','').replace('pips::','').split('(')[0][:34]
# the last forward_augmented: find the last k_border_collect and walk back to the preceding k_permute_in
idx=[i for i,r in enumerate(rows) if 'k_border_collect' in r['Kernel_Name']]
end=idx[-1]; beg=end
while beg>0 and 'k_permute_in' not in rows[beg]['Kernel_Name']: beg-=1
t0=int(rows[beg]['Start_Timestamp'])
print("forward sweep of the augmented factor: launches", end-beg+1, "span ms", (int(rows[end]['End_Timestamp'])-t0)/1e6)
for r in rows[beg:end+1]:
    print(f"  +{(int(r['Start_Timestamp'])-t0)/1e3:8.1f} us  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} us  grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>9}  {short(r['Kernel_Name'])}")
PY
find $O -name "*kernel_trace.csv" -delete
