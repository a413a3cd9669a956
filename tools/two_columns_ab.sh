# GPU box: the left-looking tail update with P tile columns per launch (PIPS_HIP_TWO_COLUMNS=P; default 4, 1 = one column per launch) - tests at P, then configs[1] A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
P=${1:-4}
cd $R; PIPS_HIP_TWO_COLUMNS=$P timeout 900 python3 -m pytest tests/test_leaf_gpu.py tests/test_golden.py tests/test_kkt_gpu.py tests/test_fuzz_gpu.py -q -m gpu -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for v in $P 4 6 1 $P 4; do
  PIPS_HIP_TWO_COLUMNS=$v python3 $R/bench.py --no-cpu-baseline --no-ipm --steps 8 --warmup 2 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('columns per launch $v:', d['ms_per_step'], d['phase_ms']['leaf_factor']['tail_update'], d['roofline']['frac'])"
done
