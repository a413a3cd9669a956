#!/bin/bash
# GPU box: kernel statistics (rocprofv3 --kernel-trace --stats) + the per-front phase clocks of the time-coupled share.  usage: profile_cfg3_quick.sh <tag> [env assignments...]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-q}; shift
for kv in "$@"; do export "$kv"; done
OUT=$R/gpurun_out/cfg3q_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="--family time-coupled --blocks-per-gpu 256 --n 50000 --no-ipm --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o cfg3 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
rm -rf $OUT/stats
PIPS_HIP_MF_CLOCKS=1 timeout 600 python3 $R/bench.py $ARGS --steps 2 --warmup 1 > $OUT/clocks.json 2> $OUT/clocks.err
python3 - $OUT <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1] + "/kernel_stats.csv")))
for r in rows[:28]:
    m = re.search(r"k_[a-z_0-9]+(<[^>]*>)?", r["Name"])
    print((m.group(0) if m else r["Name"][:30]).ljust(36), r["Calls"].rjust(6), ("%.3f ms" % (int(r["TotalDurationNs"]) / 1e6)).rjust(12), ("%.1f us" % (float(r["AverageNs"]) / 1e3)).rjust(12))
PY
grep "mf clocks" $OUT/clocks.err | tail -70
