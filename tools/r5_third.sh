#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5_third
rm -rf $O; mkdir -p $O
cd $R
timeout 1200 python3 -m pytest tests/test_deterministic_gpu.py tests/test_sweeps_gpu.py tests/test_aug_sweeps_gpu.py tests/test_sparse_root_gpu.py -q -m gpu --maxfail=20 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -30
cd /tmp && export TMPDIR=/tmp
C3="--family time-coupled --blocks-per-gpu 256 --n 50000 --no-cpu-baseline --no-ipm --steps 6 --warmup 2"
PIPS_HIP_DUMP_LEVELS=1 timeout 900 python3 $R/bench.py $C3 > $O/bench_c3.json 2> $O/bench_c3.err
timeout 900 python3 $R/bench.py $C3 --chain-blocks 256 > $O/bench_c3_chain256.json 2> $O/bench_c3_chain256.err
grep "^level" $O/bench_c3.err | head -60
for f in bench_c3 bench_c3_chain256; do python3 -c "
import json,sys
try:
    d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('$f', d['ms_per_step'], d['value'], d['phase_ms']['step'], d['phase_ms']['leaf_solves'])
except Exception as e: print('$f', 'FAILED', e); print(open('$O/$f.err').read()[-1500:])
"; done
