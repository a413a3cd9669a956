#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel statistics of the end-to-end IPM on config 2 (tools/ipm_run.py).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_ipm
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o ipm -- python3 $R/tools/ipm_run.py > $OUT/ipm.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
tail -2 $OUT/ipm.log
ls $OUT/stats
