#!/bin/bash
# GPU box: bench line, rocprofv3 kernel statistics and the two PMC passes (HBM bytes of the sparse-head kernels) of the time-coupled
# family (BASELINE configs[3]: blocks 0..255 of the 2048-block chain on one GPU; chain = 256: the 256-block chain of rounds 3-4).
# usage: profile_cfg3.sh <tag> [blocks] [n_i] [chain blocks]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r3}; NB=${2:-256}; NI=${3:-50000}; CHAIN=${4:-2048}
OUT=$R/gpurun_out/cfg3_$TAG
rm -rf $OUT; mkdir -p $OUT
ARGS="--family time-coupled --blocks-per-gpu $NB --n $NI --chain-blocks $CHAIN --no-ipm"
cd /tmp && export TMPDIR=/tmp
# (the bench line proper also carries the end-to-end IPM of the share: 12.8 M variables at 256 x 50 000)
timeout 1800 python3 $R/bench.py ${ARGS/ --no-ipm/} > $OUT/bench.json 2> $OUT/bench.err
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o cfg3 -- python3 $R/bench.py $ARGS --steps 3 --warmup 1 --no-cpu-baseline > $OUT/stats.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o cfg3 -- python3 $R/bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/fetch.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o cfg3 -- python3 $R/bench.py $ARGS --steps 1 --warmup 0 --no-cpu-baseline > $OUT/write.log 2>&1
python3 $R/tools/profile_cfg3_summarise.py $OUT $TAG $NB $NI $CHAIN > $OUT/summary.json 2> $OUT/summary.err
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_cfg3_kernel_stats.csv 2>/dev/null
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*kernel_trace.csv" -delete
tail -c 600 $OUT/bench.json; cat $OUT/summary.err | tail -5; head -c 1500 $OUT/summary.json
