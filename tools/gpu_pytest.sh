#!/bin/bash
# GPU box: the GPU test suite (all failures, not only the first).  usage: gpu_pytest.sh [pytest args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pytest
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -q -m gpu --maxfail=40 "$@" > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
grep -E "^(FAILED|ERROR)|passed|failed" $O/pytest.log | tail -50
