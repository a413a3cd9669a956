#!/bin/bash
# Runs on the GPU box (through gpurun): bench line, rocprofv3 kernel statistics, the two PMC passes the roofline object cites,
# and the counter calibration.  Outputs land in gpurun_out/prof/; tools/profile_summarise.py turns them into the files under profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r2}
OUT=$R/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-ipm > $OUT/stats.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ipm > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o bench -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-ipm > $OUT/write.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/calib_fetch -o calib -- $R/tools/pmc_calib > $OUT/calib.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/calib_write -o calib -- $R/tools/pmc_calib >> $OUT/calib.log 2>&1
python3 $R/tools/profile_summarise.py $OUT $TAG > $OUT/summary.json 2> $OUT/summary.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_bench_kernel_stats.csv 2>/dev/null
# counter files are large: keep only the per-kernel sums
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
ls -la $OUT | head -30
