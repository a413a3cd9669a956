"""Per-kernel HBM bytes of the time-coupled bench run from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_cfg3.sh, joined
with the kernel statistics; writes <tag>_cfg3_hbm_by_kernel.json and the head-traffic file bench.py's roofline object reads.
Counter units and gfx950 corrections as in tools/profile_summarise.py (MI355X_MICROARCH.md: counters in KiB; streaming reads of
8 / 16 bytes per lane report half their bytes: x 2048 per counter KiB; scattered sector reads, writes and atomics exactly: x 1024)
- the calibration kernels of tools/pmc_calib measured those factors on this pool in round 2 (profiles/r2_bench_hbm_by_kernel.json).
The chunked 8-byte reads of the solve sweeps (a panel of a few kilobytes per wave) take the x 2048 too: tools/pmc_calib, calib_chunk8,
round 5 (2035 / 1981 bytes per counter KiB for 128-byte / 16-byte aligned pieces).
usage: profile_cfg3_summarise.py <gpurun_out/cfg3_<tag> dir> <tag> [blocks per GPU] [n_i] [chain blocks]"""
import csv, glob, json, os, re, sys

out, tag = sys.argv[1], sys.argv[2]
nb, ni, chain = (int(sys.argv[3]) if len(sys.argv) > 3 else 256), (int(sys.argv[4]) if len(sys.argv) > 4 else 50000), (int(sys.argv[5]) if len(sys.argv) > 5 else 2048)
shape = {"family": "time-coupled", "blocks_per_gpu": nb, "n": ni, "schur_dim": 8000, "chain_blocks": chain}     # bench.py shape_key


def short(name):
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?)", name)
    return m.group(1) if m else name[:40]


def counter_by_kernel(sub, counter):
    res = {}
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                k = short(row.get("Kernel_Name", ""))
                e = res.setdefault(k, [0.0, set()])
                e[0] += float(row["Counter_Value"])
                e[1].add(row.get("Dispatch_Id"))
    return {k: (v[0], len(v[1])) for k, v in res.items()}


def stats():
    res = {}
    for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                res[short(row["Name"])] = (int(row["Calls"]), float(row["AverageNs"]))
    return res


fetch, write, st = counter_by_kernel("fetch", "FETCH_SIZE"), counter_by_kernel("write", "WRITE_SIZE"), stats()
# the PMC passes run warmup 0 + 1 timed step + 1 instrumented step = 2 factorizations and 8 solveCompressed
N_FACT = 2
rows, head_bytes, solve_bytes = [], 0.0, 0.0
# the kernels of a leaf solve pass (bench.py's "leaf solve sweeps" group)
SOLVE_KERNELS = {"k_permute_in", "k_permute_out", "k_leaf_fwd_gather", "k_leaf_bwd", "k_head_fwd_chain", "k_head_bwd_chain", "k_head_fwd",
                 "k_head_bwd", "k_head_solve_simple", "k_head_dscale", "k_tail_rows_fwd", "k_tail_rows_bwd", "k_tail_fwd", "k_tail_bwd",
                 "k_leaf_border", "k_tail_border_fwd", "k_border_collect", "k_border_fill"}
for k in sorted(set(fetch) | set(write)):
    fk, nd = fetch.get(k, (0.0, 0))
    wk, nw = write.get(k, (0.0, 0))
    rd, wr = fk * 2048.0, wk * 1024.0
    calls, avg_ns = st.get(k, (0, 0.0))
    n = max(nd, nw, 1)
    rows.append({"kernel": k, "dispatches_in_pmc_pass": n, "hbm_read_bytes_per_launch": rd / n, "hbm_write_bytes_per_launch": wr / n,
                 "avg_launch_us": avg_ns / 1e3, "achieved_TBps": ((rd + wr) / n) / (avg_ns * 1e-9) / 1e12 if avg_ns > 0 else None})
    if k.startswith("k_front") or k.startswith("k_head_factor") or k.startswith("k_border_schur") or k.startswith("k_root_assemble"):
        head_bytes += rd + wr
    if k.split("<")[0] in SOLVE_KERNELS:
        solve_bytes += rd + wr
rows.sort(key=lambda r: -(r["hbm_read_bytes_per_launch"] + r["hbm_write_bytes_per_launch"]) * r["dispatches_in_pmc_pass"])
json.dump({"note": "FETCH_SIZE x 2048, WRITE_SIZE x 1024 bytes per counter KiB (see docstring); separate --pmc passes", "kernels": rows},
          open(os.path.join(out, f"{tag}_cfg3_hbm_by_kernel.json"), "w"), indent=1)
json.dump({"shape": shape, "hbm_bytes_per_factorize": head_bytes / N_FACT, "kernels": "k_front<*> + k_head_factor_simple + k_border_schur + k_root_assemble",
           "solve_hbm_bytes_per_step": solve_bytes / N_FACT, "solve_kernels": sorted(SOLVE_KERNELS),
           "source": f"{tag}_cfg3_hbm_by_kernel.json", "factorizations_in_pass": N_FACT},
          open(os.path.join(out, f"{tag}_cfg3_head_traffic.json"), "w"), indent=1)
print(json.dumps({"head_hbm_bytes_per_factorize": head_bytes / N_FACT, "top": rows[:12]}, indent=1))
