cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/ipmtrace; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O -o ipm -- python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --steps 2 --warmup 1 --no-cpu-baseline > $O/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
kf=[(int(r['Start_Timestamp']),int(r['End_Timestamp'])-int(r['Start_Timestamp'])) for r in rows if r['Kernel_Name'].startswith('void pips::k_front<256, 16, false>')]
n=len(kf); per=33
print("launches", n, "factorisations", n/per)
for g in range(0, n, per):
    chunk=kf[g:g+per]
    print(f"fact {g//per:3d}: sum {sum(d for _,d in chunk)/1e6:7.3f} ms, first start {chunk[0][0]/1e9:.3f} s")
PY
find $O -name "*kernel_trace.csv" -delete
