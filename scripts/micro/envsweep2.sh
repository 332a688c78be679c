#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for e in "DL3P_GEMM_LONG_BN=0" "DL3P_GEMM_LONG_BN=1" "DL3P_GEMM_LONG_BN=1 DL3P_GEMM_LONG_NT=5" "DL3P_GEMM_LONG_BN=0" "DL3P_GEMM_LONG_BN=1" "DL3P_GEMM_LONG_ROWS=1000000000"; do
  echo -n "$e: "; env $e python3 bench.py --no-cpu-baseline --steps 50 --warmup 10 2>&1 | tail -1 | cut -c165-195
done
