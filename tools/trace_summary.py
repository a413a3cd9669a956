"""Busy time per kernel and phase boundaries of ONE factorize + ONE solveCompressed from a rocprofv3 kernel trace (csv):
trace_summary.py <trace.csv> [nth factorize]"""
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
seen, sel = -1, []
for r in rows:
    if "k_block_absmax_init" in r["Kernel_Name"] and (not sel or "k_block_absmax_init" not in sel[-1]["Kernel_Name"]):
        pass
    if "k_arena_clear" in r["Kernel_Name"]:
        seen += 1
    if seen == which:
        sel.append(r)
if not sel:
    sys.exit("no such factorisation in the trace")
t0 = int(sel[0]["Start_Timestamp"])
tot = collections.OrderedDict()
first, last = {}, {}
for r in sel:
    m = re.search(r"k_[a-z_0-9]+(<[^>]*>)?", r["Kernel_Name"])
    n = m.group(0) if m else r["Kernel_Name"][:30]
    a, b = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    e = tot.setdefault(n, [0, 0])
    e[0] += b - a; e[1] += 1
    first.setdefault(n, a); last[n] = b
print(f"{'kernel':40s} {'calls':>6s} {'busy ms':>9s} {'first us':>10s} {'last us':>10s}")
for n, (d, c) in tot.items():
    print(f"{n:40s} {c:6d} {d / 1e6:9.3f} {first[n] / 1e3:10.1f} {last[n] / 1e3:10.1f}")
