#!/bin/bash
# GPU box, round 5, last trip: the GPU test suite as the driver runs it, then every committed profile of the round with the final code
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5_final
rm -rf $O; mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -4 $O/pytest.log
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
bash tools/r5_profiles.sh > $O/profiles.log 2>&1
tail -40 $O/profiles.log | cut -c1-300
