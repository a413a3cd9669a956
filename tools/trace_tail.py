"""Timeline of the main-stream launches of one leaf factorisation (config-2-like blocks) from a rocprofv3 kernel trace (csv) of
tools/quick_bench.py: start, duration and the gap to the previous launch of the same queue:  trace_tail.py <trace.csv>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
clears = [i for i, r in enumerate(rows) if "k_arena_clear" in r["Kernel_Name"]]
seg = rows[clears[-1]:]
t0 = int(seg[0]["Start_Timestamp"])
last_end = {}
tot_gap = 0.0
for r in seg:
    m = re.search(r"k_[a-z_0-9]+(<[^>]*>)?", r["Kernel_Name"])
    n = m.group(0) if m else r["Kernel_Name"][:30]
    q = r.get("Queue_Id", "0")
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = a - last_end.get(q, a)
    last_end[q] = b
    if n.startswith("k_tail_rows") or n.startswith("k_permute"):
        break
    if n in ("k_tile_gemm_bal<0>", "k_tile_gemm<1>", "k_tile_gemm_bal<2>"):
        tot_gap += max(gap, 0)
    print(f"{a / 1e3:10.1f} us  dur {(b - a) / 1e3:8.1f}  gap {gap / 1e3:7.1f}  q{q}  {n}")
print(f"sum of gaps before update / trsm / SYRK launches on their queue: {tot_gap / 1e3:.1f} us")
