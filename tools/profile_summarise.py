"""Summarise the rocprofv3 outputs of tools/profile_bench.sh: HBM bytes of the tile-update kernel per factorize / per
launch from the FETCH_SIZE / WRITE_SIZE passes (units and the gfx950 correction as /opt/skills/guides/MI355X_MICROARCH.md
prescribes: counters are in KiB, streamed reads are reported at half their size)."""
import csv, glob, json, os, sys

out = sys.argv[1]
KERNEL = "k_tile_gemm<0>"


def counter_sum(sub, counter):
    files = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    total, launches = 0.0, set()
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if KERNEL in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    total += float(row["Counter_Value"])
                    launches.add(row.get("Dispatch_Id"))
    return total, len(launches)


fetch, n1 = counter_sum("fetch", "FETCH_SIZE")
write, n2 = counter_sum("write", "WRITE_SIZE")
# bench.py --steps 1 --warmup 0 runs 2 factorizations (the timed step + the instrumented one for the roofline object)
n_fact = 2
res = {
    "kernel": KERNEL,
    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, with --kernel-trace only) on "
              "`python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline`, MI355X (tools/profile_bench.sh)",
    "correction": "FETCH_SIZE*1024*2 (gfx950 reports half of streamed reads), WRITE_SIZE*1024",
    "factorizations_in_run": n_fact, "launches_in_run": n1,
    "hbm_read_bytes_per_factorize": fetch * 1024 * 2 / n_fact,
    "hbm_write_bytes_per_factorize": write * 1024 / n_fact,
}
res["hbm_bytes_per_factorize"] = res["hbm_read_bytes_per_factorize"] + res["hbm_write_bytes_per_factorize"]
res["hbm_bytes_per_launch"] = res["hbm_bytes_per_factorize"] * n_fact / max(n1, 1)
print(json.dumps(res, indent=1))
