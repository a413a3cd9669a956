"""Summarise the rocprofv3 outputs of tools/profile_bench.sh into the files kept under profiles/:
  * per-kernel HBM bytes from the FETCH_SIZE / WRITE_SIZE passes (counters in KiB; gfx950 correction factors measured with
    tools/pmc_calib on this very box, as /opt/skills/guides/MI355X_MICROARCH.md asks for access widths other than 16 B / lane),
  * joined with the kernel-trace statistics (calls, average duration) -> achieved TB/s per kernel,
  * the update-kernel traffic file bench.py's roofline object reads.
usage: profile_summarise.py <gpurun_out/prof dir> <round tag>"""
import csv, glob, json, os, re, sys

out, tag = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r2")


def short(name):
    m = re.search(r"(k_[a-z_0-9]+(<[^>]*>)?|calib_[a-z0-9]+)", name)
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


def stats(sub):
    res = {}
    for f in glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                res[short(row["Name"])] = (int(row["Calls"]), float(row["AverageNs"]))
    return res


# ---- calibration: bytes really moved / counter value
calib = {}
cf, cw = counter_by_kernel("calib_fetch", "FETCH_SIZE"), counter_by_kernel("calib_write", "WRITE_SIZE")
N8 = (1 << 28) * 8
moved = {"calib_read8": N8, "calib_read16": N8, "calib_gather8": (1 << 28) // 8 * 64, "calib_write8": N8, "calib_write16": N8, "calib_atomic8": N8}
for k, b in moved.items():
    src = cf if "read" in k or "gather" in k else cw
    if k in src and src[k][0] > 0:
        calib[k] = {"bytes_moved": b, "counter_KiB": src[k][0], "bytes_per_counter_KiB": b / src[k][0]}
f16 = calib.get("calib_read16", {}).get("bytes_per_counter_KiB", 2048.0)
f8 = calib.get("calib_read8", {}).get("bytes_per_counter_KiB", 2048.0)
w16 = calib.get("calib_write16", {}).get("bytes_per_counter_KiB", 1024.0)
w8 = calib.get("calib_write8", {}).get("bytes_per_counter_KiB", 1024.0)
wa = calib.get("calib_atomic8", {}).get("bytes_per_counter_KiB", 1024.0)

fetch, write, st = counter_by_kernel("fetch", "FETCH_SIZE"), counter_by_kernel("write", "WRITE_SIZE"), stats("stats")
# which width a kernel's global traffic has: the tile kernels stream 16 B / lane (LDS-DMA), everything else 8-byte accesses
WIDE = ("k_tile_gemm", "k_arena_clear")
UPDATE = ("k_tile_gemm_bal<0>", "k_tile_gemm<0>")   # the tail update kernel: balanced variant (default) or static shares
rows = []
for k in sorted(set(fetch) | set(write)):
    if k.startswith("calib_"):
        continue
    wide = k.startswith(WIDE)
    fb = fetch.get(k, (0, 0))[0] * (f16 if wide else f8)
    atom = k in ("k_tile_gemm<2>", "k_tile_gemm_bal<2>") or k.startswith("k_head_factor")
    wb = write.get(k, (0, 0))[0] * (wa if atom else (w16 if wide and k not in UPDATE else w8))
    n_pmc = max(fetch.get(k, (0, 0))[1], write.get(k, (0, 0))[1], 1)
    calls, avg_ns = st.get(k, (0, 0.0))
    per_launch = (fb + wb) / n_pmc
    rows.append({"kernel": k, "launches_in_pmc_run": n_pmc, "hbm_read_bytes_per_launch": fb / n_pmc, "hbm_write_bytes_per_launch": wb / n_pmc,
                 "avg_launch_us": avg_ns / 1e3, "achieved_TBps": (per_launch / (avg_ns * 1e-9) / 1e12) if avg_ns > 0 else None})
n_fact = 2   # bench.py --steps 1 --warmup 0 runs 2 factorizations (the timed step + the instrumented one for the roofline object)
upd = next((r for r in rows if r["kernel"] in UPDATE), None)
res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) on `python3 bench.py --steps 1 "
                 "--warmup 0 --no-cpu-baseline --no-ipm`, durations from the --kernel-trace --stats pass, MI355X (tools/profile_bench.sh)",
       "calibration": calib, "factorizations_in_pmc_run": n_fact, "kernels": rows}
json.dump(res, open(os.path.join(out, f"{tag}_bench_hbm_by_kernel.json"), "w"), indent=1)
if upd:
    tot = (upd["hbm_read_bytes_per_launch"] + upd["hbm_write_bytes_per_launch"]) * upd["launches_in_pmc_run"]
    json.dump({"kernel": upd["kernel"], "source": res["source"], "correction": f"FETCH_SIZE KiB x {f16:.0f}, WRITE_SIZE KiB x {w8:.0f} (tools/pmc_calib on the same box)",
               "factorizations_in_run": n_fact, "launches_in_run": upd["launches_in_pmc_run"],
               "hbm_read_bytes_per_factorize": upd["hbm_read_bytes_per_launch"] * upd["launches_in_pmc_run"] / n_fact,
               "hbm_write_bytes_per_factorize": upd["hbm_write_bytes_per_launch"] * upd["launches_in_pmc_run"] / n_fact,
               "hbm_bytes_per_factorize": tot / n_fact, "hbm_bytes_per_launch": tot / upd["launches_in_pmc_run"]},
              open(os.path.join(out, f"{tag}_bench_update_traffic.json"), "w"), indent=1)
print(json.dumps({"calibration": {k: round(v["bytes_per_counter_KiB"], 1) for k, v in calib.items()},
                  "kernels": [(r["kernel"], round((r["hbm_read_bytes_per_launch"] + r["hbm_write_bytes_per_launch"]) / 1e6, 1), round(r["avg_launch_us"], 1),
                               None if r["achieved_TBps"] is None else round(r["achieved_TBps"], 2)) for r in rows]}, indent=1))
