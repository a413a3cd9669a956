"""Per-launch durations of the level kernels of ONE leaf solve pass from a rocprofv3 kernel trace (csv):
trace_solve_levels.py <trace.csv> - the first k_permute_in .. k_permute_out span"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
on = False
for r in rows:
    n = r["Kernel_Name"]
    if "k_permute_in" in n:
        on = True; t0 = int(r["Start_Timestamp"])
    if on:
        m = re.search(r"k_[a-z_0-9]+(<[^>]*>)?", n)
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  "
              f"grid {int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):8d} x {r['Workgroup_Size_X']:>4s}  {m.group(0) if m else n[:30]}")
    if on and "k_permute_out" in n:
        break
