R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/konly; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -o k -- python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --no-cpu-baseline --no-ipm --steps 4 --warmup 1 > $O/prof_bench.txt 2>&1
python3 - <<PY
import csv,glob,re
fs=glob.glob('/tmp/kp/**/*kernel_stats.csv', recursive=True)
if not fs: raise SystemExit('no stats file')
for r in list(csv.DictReader(open(fs[0])))[:16]:
    m = re.search(r"k_[a-z_0-9]+(<[^>]*>)?", r["Name"])
    print((m.group(0) if m else r["Name"][:30]).ljust(36), r["Calls"].rjust(6), ("%.3f ms" % (int(r["TotalDurationNs"]) / 1e6)).rjust(12), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(12))
f=glob.glob('/tmp/kp/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
br=[r for r in rows if 'k_border_rows' in r['Kernel_Name'] and 'dense' not in r['Kernel_Name']]
print(len(br), 'k_border_rows launches')
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in br]
g=[int(r.get('Grid_Size_X') or r.get('Grid_Size') or 0) for r in br]
n=len(br)//5
for i in range(0, n, max(1,n//50)): print(i, g[i]//64, round(d[i],1))
print('sum per step us', sum(d[:n]), 'launches per step', n)
PY
