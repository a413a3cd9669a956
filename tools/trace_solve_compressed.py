"""Kernel sequence of ONE solveCompressed from a rocprofv3 kernel trace (csv) of tools/config3_probe.py: everything between the second
and the third k_border_tmult (one per Lsolve), consecutive launches of one kernel merged:  trace_solve_compressed.py <trace.csv>"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_border_tmult" in r["Kernel_Name"]]
if len(marks) < 3:
    sys.exit("fewer than three solveCompressed calls in the trace")
seg = rows[marks[1] + 1:marks[2] + 1]
t0 = int(seg[0]["Start_Timestamp"])
out = []
for r in seg:
    m = re.search(r"k_[a-z_0-9]+(<[^>]*>)?", r["Kernel_Name"])
    n = m.group(0) if m else r["Kernel_Name"][:32]
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if out and out[-1][0] == n:
        out[-1][2] = b; out[-1][3] += b - a; out[-1][4] += 1
    else:
        out.append([n, a, b, b - a, 1])
prev = 0
for n, a, b, busy, cnt in out:
    print(f"{a / 1e3:9.1f} us  gap {max(a - prev, 0) / 1e3:6.1f}  busy {busy / 1e3:8.1f} us  x{cnt:<3d} {n}")
    prev = b
print(f"span {prev / 1e3:.1f} us")
