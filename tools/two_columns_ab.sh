# GPU box: the left-looking tail update with two tile columns per launch (PIPS_HIP_TWO_COLUMNS=1) against one - tests, then configs[1] A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; PIPS_HIP_TWO_COLUMNS=1 timeout 900 python3 -m pytest tests/test_leaf_gpu.py tests/test_golden.py tests/test_kkt_gpu.py tests/test_fuzz_gpu.py tests/test_configs_gpu.py -q -m gpu -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for v in 1 0 1 0; do   # (1 is the default)
  PIPS_HIP_TWO_COLUMNS=$v python3 $R/bench.py --no-cpu-baseline --no-ipm --steps 8 --warmup 2 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('two_columns=$v', d['ms_per_step'], d['phase_ms']['leaf_factor'], d['roofline']['frac'])"
done
