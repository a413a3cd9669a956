"""Design aid for the single-launch dense root (DenseLdl, csrc/rootplan.cpp): list-schedule the tile tasks of a left-looking tiled
LDL^T of ntc tile columns on W workgroup slots with a cost model, print the makespan, the task count and the depth histogram.
The C++ plan builder restates the policy chosen here.  Usage: python tools/root_schedule_sim.py 125 [qmin] [diag_us]"""
import heapq
import sys


def simulate(ntc, W=512, qmin=4, t_step=36.0, t0=8.0, t_trsm=40.0, t_diag=70.0, urgent=1, verbose=False):
    prog = [[0] * (i + 1) for i in range(ntc)]          # tile columns applied to tile (i, j)
    busy = [[False] * (i + 1) for i in range(ntc)]      # a task on the tile is running
    rowdone = [0] * ntc                                 # L(i, k) final for k < rowdone[i]
    dready = [False] * ntc
    trsm_done = [[False] * (i + 1) for i in range(ntc)]
    events = []                                         # (time, seq, kind, i, j, k1)
    free = W
    t = 0.0
    seq = 0
    tasks = []
    chain = 0                                           # first column whose diagonal tile is not factorised yet
    depth_hist = {}
    first_col = 0                                       # first column with unfinished tiles

    def avail(i, j):
        return min(rowdone[i], rowdone[j], j)

    def pick():
        # 1. diagonal tile of the chain column
        nonlocal chain
        j = chain
        if j < ntc and not busy[j][j] and not dready[j] and prog[j][j] == j:
            return ("D", j, j, j)
        # 2. trsm, rows closest to the diagonal first, columns left to right
        for jj in range(first_col, min(chain + 1, ntc)):
            if not dready[jj]:
                continue
            for i in range(jj + 1, ntc):
                if not trsm_done[i][jj] and not busy[i][jj] and prog[i][jj] == jj:
                    return ("T", i, jj, jj)
        # 3. updates: nearest column first; eligible when final, deep enough or urgent
        for jj in range(first_col, ntc):
            for i in range(jj, ntc):
                if busy[i][jj] or prog[i][jj] == jj:
                    continue
                a = avail(i, jj)
                q = a - prog[i][jj]
                if q <= 0:
                    continue
                if a == jj or q >= qmin or (jj <= chain + urgent and i <= jj + urgent):
                    return ("U", i, jj, a)
        return None

    n_left = ntc * (ntc + 1) // 2 + ntc * (ntc - 1) // 2  # diag + trsm completions ... loop ends when all diagonal tiles are done
    while chain < ntc or events:
        started = False
        while free > 0:
            p = pick()
            if p is None:
                break
            kind, i, j, k1 = p
            busy[i][j] = True
            if kind == "D":
                dur = t_diag
            elif kind == "T":
                dur = t_trsm
            else:
                q = k1 - prog[i][j]
                dur = t0 + q * t_step
                depth_hist[q] = depth_hist.get(q, 0) + 1
            tasks.append((t, kind, i, j, prog[i][j], k1))
            seq += 1
            heapq.heappush(events, (t + dur, seq, kind, i, j, k1))
            free -= 1
            started = True
        if not events:
            break
        t, _, kind, i, j, k1 = heapq.heappop(events)
        free += 1
        busy[i][j] = False
        if kind == "D":
            dready[j] = True
            rowdone[j] = max(rowdone[j], j)   # (row j's off-diagonal tiles were final before)
            chain = j + 1
        elif kind == "T":
            trsm_done[i][j] = True
            rowdone[i] = j + 1
        else:
            prog[i][j] = k1
        while first_col < ntc and dready[first_col] and all(trsm_done[i][first_col] for i in range(first_col + 1, ntc)):
            first_col += 1
    return t, tasks, depth_hist


if __name__ == "__main__":
    ntc = int(sys.argv[1]) if len(sys.argv) > 1 else 125
    qmin = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    t_diag = float(sys.argv[3]) if len(sys.argv) > 3 else 70.0
    urgent = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    T, tasks, hist = simulate(ntc, qmin=qmin, t_diag=t_diag, urgent=urgent)
    n_upd = sum(1 for x in tasks if x[1] == "U")
    steps = sum(k * v for k, v in hist.items())
    S = ntc * 128
    print(f"ntc {ntc} qmin {qmin} diag {t_diag} us: makespan {T/1e3:.2f} ms = {S**3/3/(T*1e-6)/1e12:.1f} TFLOP/s; {len(tasks)} tasks, {n_upd} updates, "
          f"mean depth {steps/max(n_upd,1):.1f}; work-bound {(steps*36.0 + n_upd*8.0)/512/1e3:.2f} ms")
    print("depth histogram:", sorted(hist.items())[:12], "...")
