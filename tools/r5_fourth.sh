#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5_fourth
rm -rf $O; mkdir -p $O
cd $R
timeout 1800 python3 -m pytest tests -q -m gpu --maxfail=20 --deselect tests/test_ipm_gpu.py::test_configs1_matches_the_cpu_pardiso_path > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -30
cd /tmp && export TMPDIR=/tmp
timeout 300 $R/tools/mb2 > $O/mb2.txt 2>&1
tail -6 $O/mb2.txt
C3="--family time-coupled --blocks-per-gpu 256 --n 50000 --no-cpu-baseline --no-ipm --steps 6 --warmup 2"
timeout 900 python3 $R/bench.py $C3 > $O/bench_c3.json 2> $O/bench_c3.err
PIPS_HIP_DETERMINISTIC=1 timeout 900 python3 $R/bench.py $C3 --steps 3 --warmup 1 > $O/bench_c3_det.json 2> $O/bench_c3_det.err
PIPS_HIP_DETERMINISTIC=1 timeout 900 python3 $R/bench.py --no-cpu-baseline --no-ipm --steps 3 --warmup 1 > $O/bench_c1_det.json 2> $O/bench_c1_det.err
for f in bench_c3 bench_c3_det bench_c1_det; do python3 -c "
import json,sys
try:
    d=json.loads([l for l in open('$O/$f.json') if l.startswith('{')][-1]); print('$f', d['ms_per_step'], d['value'], d['phase_ms']['accounted'], d['phase_ms']['instrumented_step_wall'], d['phase_ms']['leaf_factor'], d['phase_ms']['step'])
except Exception as e: print('$f', 'FAILED', e); print(open('$O/$f.err').read()[-1500:])
"; done
