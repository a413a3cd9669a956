"""Summarise a PIPS_HIP_TAIL_TRACE file (per-task clocks of one single-launch factorisation of the leaf tails, 100 MHz): how long the task kinds
wait and work, how busy the 512 slots are, and the same per group of tile columns.  python tools/tail_trace.py <file>"""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.int64)
lst, kind, blk, ti, tj, pad, t0, t1, t2 = (d[:, k] for k in range(1, 10))
ok = t2 > 0
base = t0[ok].min()
us = lambda x: (x - base) / 100.0
span = us(t2[ok].max())
print(f"{len(d)} tasks ({int((~ok).sum())} without clocks), span {span/1e3:.2f} ms")
for k, name in ((0, "UPD"), (1, "TRSM"), (2, "DIAG")):
    m = (kind == k) & ok
    if not m.any():
        continue
    wait, work = (t1[m] - t0[m]) / 100.0, (t2[m] - t1[m]) / 100.0
    extra = ""
    if k == 0:
        depth = (pad[m] >> 16) - (pad[m] & 0xffff)
        extra = f", mean depth {depth.mean():.1f}, us per K step {(work.sum() / depth.sum()):.1f}; tasks of depth 1-3: {int((depth <= 3).sum())} with work mean {work[depth <= 3].mean():.1f} us"
    print(f"{name}: {m.sum()} tasks, wait mean {wait.mean():.1f} us (sum {wait.sum()/1e3:.1f} ms), work mean {work.mean():.1f} us (sum {work.sum()/1e3:.1f} ms){extra}")
print(f"slot occupancy by work (512 slots): {((t2[ok] - t1[ok]).sum() / 100.0) / (span * 512):.2f}; by work + wait: {((t2[ok] - t0[ok]).sum() / 100.0) / (span * 512):.2f}")
# per tile column of the task: first draw, last end, work and wait inside
ntc = int(tj.max()) + 1
print("col   first draw us   last end us   work ms   wait ms   (by the column of the task's tile)")
for j in range(ntc):
    m = (tj == j) & ok
    if m.any():
        print(f"{j:3d} {us(t0[m].min()):12.0f} {us(t2[m].max()):12.0f} {(t2[m]-t1[m]).sum()/1e5:9.2f} {(t1[m]-t0[m]).sum()/1e5:9.2f}")
