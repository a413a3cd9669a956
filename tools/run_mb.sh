#!/bin/bash
# GPU box: the update kernel alone (tools/microbench.hip: k_tile_gemm<0> on one block's deep-K column update).  usage: run_mb.sh [tile columns] [rows]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/tools
timeout 120 ./microbench ${1:-32} ${2:-2048} | grep -v "mfma_f64 peak"
