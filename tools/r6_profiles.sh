#!/bin/bash
# GPU box, round 6: every committed profile of the round in one trip - configs[1] with PMC passes and calibration, the configs[3] shape
# (blocks 0..255 of the 2048-block chain) with PMC passes, the other per-GPU shares and configs[4] whole, the dense root at three sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
make -C tools pmc_calib > /dev/null 2>&1
bash tools/profile_bench.sh r6 > $R/gpurun_out/profile_bench_r6.log 2>&1
bash tools/profile_cfg3.sh r6 256 50000 2048 > $R/gpurun_out/profile_cfg3_r6.log 2>&1
bash tools/other_configs.sh > $R/gpurun_out/other_configs_r6.log 2>&1
bash tools/r6_root_profile.sh > $R/gpurun_out/root_profile_r6.log 2>&1
tail -5 $R/gpurun_out/profile_bench_r6.log; tail -3 $R/gpurun_out/profile_cfg3_r6.log; tail -30 $R/gpurun_out/other_configs_r6.log | cut -c1-400
