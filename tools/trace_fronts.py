"""Per-launch durations of the head kernels of ONE factorisation from a rocprofv3 kernel trace (csv): trace_fronts.py <trace.csv> [nth factorize]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
seen, out = -1, []
for r in rows:
    name = r["Kernel_Name"]
    if "k_arena_clear" in name:
        seen += 1
    if seen == which:
        m = re.search(r"k_[a-z_0-9]+(<[^>]*>)?", name)
        out.append((m.group(0) if m else name[:30], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("LDS_Block_Size", "?")))
t0 = out[0][1] if out else 0
for n, a, b, g, wg, lds in out:
    print(f"{(a - t0) / 1e3:9.1f} us  +{(b - a) / 1e3:8.1f} us  {n:34s} grid {g:>9s} wg {wg:>5s} lds {lds}")
    if "k_pref_tail" in n:
        break
