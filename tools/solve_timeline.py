"""Timeline of the last solveCompressed of a bench run from a rocprofv3 kernel trace: kernels, durations, gaps (development aid)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r): return r["Kernel_Name"].split("(")[0].replace("void ", "").replace("pips::", "")[:34]
# last k_tail_rows_fwd marks the last Lsolve; start of that solve_compressed = the k_permute_in before it
idx = [i for i, r in enumerate(rows) if "k_tail_rows_fwd" in r["Kernel_Name"]]
last = idx[-1]
i0 = last
while i0 > 0 and "k_permute_in" not in rows[i0]["Kernel_Name"]: i0 -= 1
i1 = last
while i1 + 1 < len(rows) and "k_tile_gemm" not in rows[i1 + 1]["Kernel_Name"] and "k_arena_clear" not in rows[i1 + 1]["Kernel_Name"]: i1 += 1
seg = rows[max(i0 - 2, 0):i1 + 1]
t0 = int(seg[0]["Start_Timestamp"]); prev = t0; busy = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{nm(r):34s} start {(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:6.1f}  queue {r['Queue_Id']}")
    busy += e - s; prev = max(prev, e)
print("span %.1f us, kernel time %.1f us" % ((prev - t0) / 1e3, busy / 1e3))
