# GPU box: the launches of one forward and one backward head sweep on the configs[3] shape, level by level (supernodes, microseconds)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kp -o k -- python3 $R/bench.py --family time-coupled --blocks-per-gpu 256 --n 50000 --no-cpu-baseline --no-ipm --steps 2 --warmup 1 > /dev/null 2>&1
python3 - <<PY
import csv,glob,re
f=glob.glob('/tmp/kp/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for name in ('k_head_fwd_chain','k_head_bwd_chain'):
    r=[x for x in rows if name in x['Kernel_Name']]
    r=r[-20:]
    print(name, 'last sweep:', ' '.join(f"{int(x['Grid_Size_X'])//64}:{(int(x['End_Timestamp'])-int(x['Start_Timestamp']))/1e3:.0f}" for x in r), 'sum', sum((int(x['End_Timestamp'])-int(x['Start_Timestamp']))/1e3 for x in r))
# gaps between consecutive kernels in the last forward sweep
r=[x for x in rows]
idx=[i for i,x in enumerate(r) if 'k_head_fwd_chain' in x['Kernel_Name']][-20:]
a,b=idx[0]-3,idx[-1]+6
prev=None
for x in r[a:b]:
    m=re.search(r"k_[a-z_0-9]+", x['Kernel_Name']); s,e=int(x['Start_Timestamp']),int(x['End_Timestamp'])
    print(m.group(0) if m else x['Kernel_Name'][:20], int(x['Grid_Size_X'])//int(x['Workgroup_Size_X']), f"{(e-s)/1e3:.1f} us", f"gap {(s-prev)/1e3:.1f}" if prev else "")
    prev=e
PY
