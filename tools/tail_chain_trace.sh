# GPU box: the tail factorisation of configs[1] launch by launch (one factorisation): update / diagonal / trsm kernels with start offsets, durations and the idle gaps of the main stream
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kp -o k -- python3 $R/bench.py --no-cpu-baseline --no-ipm --steps 1 --warmup 1 > /dev/null 2>&1
python3 - <<PY
import csv,glob,re
f=glob.glob('/tmp/kp/**/*kernel_trace.csv', recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r['Start_Timestamp']))
# last factorisation: from the last k_arena_clear
i0=[i for i,r in enumerate(rows) if 'k_arena_clear' in r['Kernel_Name']][-1]
i1=[i for i,r in enumerate(rows) if i>i0 and 'k_tile_gemm_bal<2>' in r['Kernel_Name'].replace('(int)','')]
seg=rows[i0:(i1[0]+1 if i1 else i0+400)]
t0=int(seg[0]['Start_Timestamp'])
busy_end=0; gaps=0.0; out=[]
for r in seg:
    m=re.search(r"k_[a-z_0-9]+(<[^>]*>)?", r['Kernel_Name']); n=m.group(0) if m else r['Kernel_Name'][:20]
    s,e=int(r['Start_Timestamp'])-t0,int(r['End_Timestamp'])-t0
    if not n.startswith('k_tile'): continue
    out.append((n, int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), s/1e3, (e-s)/1e3, r['Stream_Id'] if 'Stream_Id' in r else r.get('Queue_Id')))
# idle time of the device between tile kernels (no tile kernel running)
ev=sorted([(o[2],o[2]+o[3]) for o in out])
cur=ev[0][1]; idle=0
for s,e in ev[1:]:
    if s>cur: idle+=s-cur
    cur=max(cur,e)
print('tile kernels', len(out), 'span us', ev[-1][1]-ev[0][0] if ev else 0, 'device idle between them us', round(idle,1))
for o in out[40:100]: print(o[0].ljust(22), str(o[1]).rjust(6), 'start %.0f' % o[2], 'dur %.1f' % o[3], 'q', o[4])
PY
