#!/bin/bash
# GPU box, round 5 first trip: the GPU test suite, the default bench line, the configs[3] chain (2048-block chain share, 256-block chain),
# and the counter calibration with the chunked-read pattern of the solve sweeps
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5_first
rm -rf $O; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -5 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py --steps 10 --warmup 2 > $O/bench_c1.json 2> $O/bench_c1.err
timeout 600 python3 $R/bench.py --steps 10 --warmup 2 --solve-check-every 0 --no-cpu-baseline --no-ipm > $O/bench_c1_nocheck.json 2> $O/bench_c1_nocheck.err
C3="--family time-coupled --blocks-per-gpu 256 --n 50000 --no-cpu-baseline --no-ipm --steps 6 --warmup 2"
timeout 900 python3 $R/bench.py $C3 > $O/bench_c3.json 2> $O/bench_c3.err
timeout 900 python3 $R/bench.py $C3 --chain-blocks 256 > $O/bench_c3_chain256.json 2> $O/bench_c3_chain256.err
timeout 900 python3 $R/bench.py $C3 --chain-blocks 256 --solve-check-every 0 > $O/bench_c3_chain256_nocheck.json 2> $O/bench_c3_chain256_nocheck.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/calib_fetch -o calib -- $R/tools/pmc_calib > $O/calib.log 2>&1
python3 - <<'PY' > $O/calib_summary.txt 2>&1
import csv, glob, os, re
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/r5_first"
tot = {}
for f in glob.glob(O + "/calib_fetch/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if row.get("Counter_Name") == "FETCH_SIZE":
            m = re.search(r"(calib_[a-z0-9]+(<[^>]*>)?)", row["Kernel_Name"])
            if m:
                tot[m.group(1)] = tot.get(m.group(1), 0.0) + float(row["Counter_Value"])
moved = {"calib_read8": 2**31, "calib_read16": 2**31, "calib_gather8": 2**25 * 64, "calib_chunk8<528>": 2**19 * 528 * 8, "calib_chunk8<529>": 2**19 * 529 * 8}
for k, v in sorted(tot.items()):
    print(k, "counter KiB", v, "bytes moved", moved.get(k), "bytes per counter KiB", (moved[k] / v if k in moved and v else None))
PY
cat $O/calib_summary.txt
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
for f in bench_c1 bench_c1_nocheck bench_c3 bench_c3_chain256 bench_c3_chain256_nocheck; do python3 -c "
import json,sys
try:
    d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('$f', d['ms_per_step'], d['value'], d['config']['solve_paths_last_step'], d['config'].get('solve_checks'), d['phase_ms']['step'])
except Exception as e: print('$f', 'FAILED', e)
"; done
