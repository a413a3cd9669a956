#!/bin/bash
# GPU box: kernel statistics of factorize / solveCompressed on the time-coupled family (BASELINE configs[3] per-GPU share).
# usage: profile_cfg3.sh <tag> [blocks] [n_i]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r3}; NB=${2:-256}; NI=${3:-50000}
OUT=$R/gpurun_out/cfg3_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 $R/tools/config3_probe.py $NB $NI > $OUT/probe.txt 2> $OUT/probe.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o cfg3 -- python3 $R/tools/config3_probe.py $NB $NI > $OUT/stats.log 2>&1
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_cfg3_kernel_stats.csv 2>/dev/null
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/probe.txt; head -40 $OUT/${TAG}_cfg3_kernel_stats.csv
