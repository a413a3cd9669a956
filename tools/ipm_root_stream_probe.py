"""Development aid: end-to-end IPM (bench.ipm_end_to_end) of N random blocks with the dense root on its own stream or on the main stream
(PIPS_HIP_ROOT_SYNC=1).  usage: python tools/ipm_root_stream_probe.py [N n_i S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pips_ipmpp_amd as pa
import bench
N, n_i, S = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (16, 10000, 2000)
r = bench.ipm_end_to_end(pa, 5, N, n_i, n_i // 2, S // 2, S // 2, 0.001)
print(f"ROOT_SYNC={os.environ.get('PIPS_HIP_ROOT_SYNC')}: {r['iterations']} iterations, {r['seconds']:.3f} s, {1e3 * r['seconds'] / r['iterations']:.2f} ms per iteration, "
      f"{r['factorizations']} factorisations, {r['solve_compressed']} solveCompressed", flush=True)
