"""Block-angular SpMV of the device harness (harness.hip k_spmv, SURVEY 8 f-2: DistributedMatrix::mult / transpose_mult) on the LP of the
time-coupled share (256 x 50 000): J x and J^T y, ten times each.  Run under tools/kstats.sh: the kernel statistics give the average duration of
k_spmv<*>, this script prints the bytes one product moves (values 8 + column index 4 per entry, row pointers, input and output vectors)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pips_ipmpp_amd as pa
import families

c3 = families.CONFIG3_SHARE
N, n_i = int(os.environ.get("SPMV_BLOCKS", "256")), 50000
blocks, F0, my_i, myl = families.time_coupled_blocks(N, n_i, c3["L"], c3["n0"], c3["bw"], c3["nnz_row"], c3["seed"])
n0 = c3["n0"]
rng = np.random.default_rng(0)
c = rng.uniform(0.5, 1.5, n0 + N * n_i)
b = rng.uniform(0.5, 1.5, myl + N * my_i)
ipm = pa.IpmSolver(n0, myl, blocks, F0, c, b)
nnz = F0.rowptr[-1] + sum(int(W.rowptr[-1] + T.rowptr[-1] + F.rowptr[-1]) for (W, T, F) in blocks)
nx, ny = ipm.nx, ipm.ny + getattr(ipm, 'nzr', 0)
x, y = rng.standard_normal(nx), rng.standard_normal(ny)
import ctypes as C
lib = pa.capi.lib
ox, oy = np.zeros(nx), np.zeros(ny)
for _ in range(10):     # (IpmSolver has no mult of its own: the C entry serves every harness handle)
    assert lib.pips_ipm_mult(ipm._h, C.c_int(0), x.ctypes.data_as(C.c_void_p), oy.ctypes.data_as(C.c_void_p)) == 0
    assert lib.pips_ipm_mult(ipm._h, C.c_int(1), y.ctypes.data_as(C.c_void_p), ox.ctypes.data_as(C.c_void_p)) == 0
out = {"nnz_J": int(nnz), "nx": int(nx), "rows": int(ny),
       "bytes_J_x": int(12 * nnz + 4 * (ny + 1) + 8 * (nx + ny)), "bytes_Jt_y": int(12 * nnz + 4 * (nx + 1) + 8 * (nx + ny))}
print(json.dumps(out))
ipm.close()
