"""Timeline of one dense-root factorisation from a rocprofv3 kernel trace (tools/root_probe.py <S> under --kernel-trace):
per kernel kind: launches, busy time, and the gaps on the critical chain diag -> trsm -> next-column update -> diag."""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    n = r["Kernel_Name"]
    for k in ("k_tile_diag", "k_tile_gemm<1>", "k_tile_gemm<3>", "k_tile_gemm<4>", "k_tile_gemm_bal<0>", "k_tile_gemm<0>"):
        if k in n: return k
    return n.split("(")[0][-30:]
# last factorisation = last run of 'k_tile_diag' launches: take the last N diag launches where N = count/number_of_factorisations
diags = [i for i, r in enumerate(rows) if nm(r) == "k_tile_diag"]
nfac = int(sys.argv[2]) if len(sys.argv) > 2 else 4
per = len(diags) // nfac
first = diags[-per]
seg = rows[first:]
t0 = int(seg[0]["Start_Timestamp"])
tend = max(int(r["End_Timestamp"]) for r in seg if nm(r).startswith("k_tile"))
print("factorisation span %.2f ms, %d diag launches" % ((tend - t0) / 1e6, per))
busy = collections.defaultdict(lambda: [0, 0.0])
for r in seg:
    k = nm(r)
    if not k.startswith("k_tile"): continue
    busy[k][0] += 1; busy[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, v in busy.items(): print(f"  {k:22s} {v[0]:5d} launches  {v[1]/1e3:8.2f} ms busy  avg {v[1]/v[0]:8.1f} us")
# chain: for consecutive diag launches: diag duration, time from diag end to next diag start
dl = [r for r in seg if nm(r) == "k_tile_diag"]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in dl]
gap = [(int(dl[i + 1]["Start_Timestamp"]) - int(dl[i]["End_Timestamp"])) / 1e3 for i in range(len(dl) - 1)]
import statistics as st
print("diag duration us: mean %.1f  min %.1f  max %.1f ; first 5 %s ; last 5 %s" % (st.mean(dur), min(dur), max(dur), [round(x) for x in dur[:5]], [round(x) for x in dur[-5:]]))
print("diag end -> next diag start us: mean %.1f  first 5 %s  mid 5 %s  last 5 %s" % (st.mean(gap), [round(x) for x in gap[:5]], [round(x) for x in gap[60:65]], [round(x) for x in gap[-5:]]))
# what runs between: for a middle column list the kernels between diag k and diag k+1
k = len(dl) // 2
a, b = int(dl[k]["Start_Timestamp"]), int(dl[k + 1]["Start_Timestamp"])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > a and s < b + 1 and nm(r).startswith("k_tile"):
        print(f"    {nm(r):20s} start {(s - a)/1e3:8.1f} us  dur {(e - s)/1e3:8.1f} us  grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])} wgs queue {r['Queue_Id']}")
