#!/bin/bash
# GPU box, round 5: the GPU test suite, then the bench lines of configs[1], the configs[3] shape (2048-block chain share) and the 256-block chain
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5_second
rm -rf $O; mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests -q -m gpu --maxfail=20 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -30
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.err
C3="--family time-coupled --blocks-per-gpu 256 --n 50000 --no-cpu-baseline --no-ipm --steps 6 --warmup 2"
timeout 900 python3 $R/bench.py $C3 > $O/bench_c3.json 2> $O/bench_c3.err
timeout 900 python3 $R/bench.py $C3 --chain-blocks 256 > $O/bench_c3_chain256.json 2> $O/bench_c3_chain256.err
timeout 300 $R/tools/mb2 > $O/mb2.txt 2>&1
tail -3 $O/mb2.txt
for f in bench_c1 bench_c3 bench_c3_chain256; do python3 -c "
import json,sys
try:
    d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('$f', d['ms_per_step'], d['value'], d['config']['solve_paths_last_step'], d['phase_ms']['accounted'], d['phase_ms']['instrumented_step_wall'], d['phase_ms']['step'], d['phase_ms']['leaf_solves'], (d.get('ipm_end_to_end') or {}).get('cpu_pardiso_path'))
except Exception as e: print('$f', 'FAILED', e); print(open('$O/$f.err').read()[-1500:])
"; done
