"""Summarise a PIPS_HIP_ROOT_TRACE file (per-task clocks of one single-launch root factorisation, 100 MHz): where the chain of
diagonal tiles spends its time, how long the task kinds take, how busy the slots are.  python tools/root_trace.py <file>"""
import sys
import numpy as np
d = np.loadtxt(sys.argv[1], dtype=np.int64)
kind, ti, tj, pad, t0, t1, t2 = d[:, 1], d[:, 2], d[:, 3], d[:, 4], d[:, 5], d[:, 6], d[:, 7]
base = t0.min()
us = lambda x: (x - base) / 100.0
span = us(t2.max())
print(f"{len(d)} tasks, span {span/1e3:.2f} ms")
for k, name in ((0, "UPD"), (1, "TRSM"), (2, "DIAG")):
    m = kind == k
    if not m.any():
        continue
    wait, work = (t1[m] - t0[m]) / 100.0, (t2[m] - t1[m]) / 100.0
    extra = ""
    if k == 0:
        depth = (pad[m] >> 16) - (pad[m] & 0xffff)
        extra = f", mean depth {depth.mean():.1f}, us per K step {(work.sum() / depth.sum()):.1f}"
    print(f"{name}: {m.sum()} tasks, wait mean {wait.mean():.1f} us (sum {wait.sum()/1e3:.1f} ms), work mean {work.mean():.1f} us (sum {work.sum()/1e3:.1f} ms){extra}")
busy = ((t2 - t1).sum() / 100.0) / (span * 512)
print(f"slot occupancy by work (512 slots): {busy:.2f}")
# the chain: DIAG j end -> TRSM (j+1, j) start/end -> last UPD of (j+1, j+1) start/end -> DIAG j+1 start
diag = {int(tj[i]): i for i in np.nonzero(kind == 2)[0]}
trsm = {(int(ti[i]), int(tj[i])): i for i in np.nonzero(kind == 1)[0]}
lastupd = {}
for i in np.nonzero(kind == 0)[0]:
    if ti[i] == tj[i] and (pad[i] >> 16) == tj[i]:
        lastupd[int(tj[i])] = i
ntc = max(diag) + 1
rows = []
for j in range(ntc - 1):
    a, b, c, e = diag[j], trsm[(j + 1, j)], lastupd[j + 1], diag[j + 1]
    rows.append((us(t1[a]), (t2[a] - t1[a]) / 100.0, (t1[b] - t2[a]) / 100.0, (t2[b] - t1[b]) / 100.0, (t1[c] - t2[b]) / 100.0,
                 (t2[c] - t1[c]) / 100.0, (t1[e] - t2[c]) / 100.0, (pad[c] >> 16) - (pad[c] & 0xffff)))
r = np.array(rows)
print("chain step (us): diag | gap | trsm(j+1,j) | gap | last upd (j+1,j+1) [depth] | gap  -- means over thirds of the columns")
for lo, hi in ((0, ntc // 3), (ntc // 3, 2 * ntc // 3), (2 * ntc // 3, ntc - 1)):
    s = r[lo:hi]
    if len(s):
        print(f"  columns {lo:3d}-{hi:3d}: {s[:,1].mean():6.1f} | {s[:,2].mean():6.1f} | {s[:,3].mean():6.1f} | {s[:,4].mean():6.1f} | {s[:,5].mean():6.1f} [{s[:,7].mean():.1f}] | {s[:,6].mean():6.1f}   sum {s[:,1:7].sum(axis=1).mean():.1f}")
