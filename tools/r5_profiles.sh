#!/bin/bash
# GPU box, round 5: every committed profile of the round in one trip (configs[1] with PMC passes and calibration; the configs[3] shape - blocks
# 0..255 of the 2048-block chain - and the 256-block chain of rounds 3-4 with PMC passes; the other per-GPU shares)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/profile_bench.sh r5 > $R/gpurun_out/profile_bench_r5.log 2>&1
bash tools/profile_cfg3.sh r5 256 50000 2048 > $R/gpurun_out/profile_cfg3_r5.log 2>&1
bash tools/profile_cfg3.sh r5chain256 256 50000 256 > $R/gpurun_out/profile_cfg3_r5chain256.log 2>&1
bash tools/other_configs.sh > $R/gpurun_out/other_configs_r5.log 2>&1
tail -5 $R/gpurun_out/profile_bench_r5.log; tail -3 $R/gpurun_out/profile_cfg3_r5.log; tail -30 $R/gpurun_out/other_configs_r5.log | cut -c1-400
