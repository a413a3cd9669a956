# GPU box: kernel trace of the time-coupled probe; per-launch list of the head of one factorisation + per-kernel busy times
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace1; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $R/tools/config3_probe.py ${1:-64} 50000 > $OUT/log.txt 2>&1
T=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_fronts.py $T 1 > $OUT/fronts.txt
python3 $R/tools/trace_summary.py $T 1 > $OUT/summary.txt
# the last factorisation of the probe and everything after it: 3 solveCompressed, 3 leaf solves
python3 $R/tools/trace_summary.py $T 2 > $OUT/summary_last.txt
python3 $R/tools/trace_solve_levels.py $T > $OUT/solve_levels.txt
python3 $R/tools/trace_solve_compressed.py $T > $OUT/solve_compressed.txt
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/summary_last.txt
