import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_trace.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
import collections
for name in ("k_tail_fwd","k_tail_bwd"):
    sel=[r for r in rows if name in r['Kernel_Name']]
    bygrid=collections.defaultdict(list)
    for r in sel: bygrid[int(r['Grid_Size_X'])//256].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    tot=sum(sum(v) for v in bygrid.values())/ (len(sel)/35)
    ks=sorted(bygrid)
    print(name,"per sweep us",round(tot,1)," tasks:dur ", " ".join(f"{k}:{sum(bygrid[k])/len(bygrid[k]):.1f}" for k in ks[::4]))
