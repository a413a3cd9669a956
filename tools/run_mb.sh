#!/bin/bash
# GPU box: GEMM micro-benchmark variants (tools/Makefile)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/tools
for v in microbench microbench_BUILTIN microbench_Q0 microbench_NO_DMA microbench_NO_BARRIER microbench_CLOCK; do
  echo "== $v"; timeout 120 ./$v 32 2048 | grep -v "mfma_f64 peak"
done
