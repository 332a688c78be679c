#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1100 bash scripts/pmc_ops.sh r03 > gpurun_out/j_pmc_ops.log 2>&1
timeout 700 bash scripts/step_table.sh mobilenetv2 > /dev/null 2>&1
timeout 700 bash scripts/step_table.sh xception > /dev/null 2>&1
DL3P_ST_N=1 DL3P_ST_H=1024 DL3P_ST_W=2048 DL3P_ST_C=19 DL3P_ST_DTYPE=bf16 timeout 700 bash scripts/step_table.sh mobilenetv3large > /dev/null 2>&1
mv gpurun_out/step_table_mobilenetv3large.txt gpurun_out/step_table_mobilenetv3large_bf16.txt
timeout 900 bash scripts/bench_table.sh > /dev/null 2>&1
timeout 300 python3 scripts/micro/sb_gemm.py > gpurun_out/split_gemm.txt 2>&1
timeout 600 python -m pytest tests/test_split_model_gpu.py tests/test_split_gemm_gpu.py -q --timeout 500 > gpurun_out/j_split.log 2>&1; echo "split rc=$?" > gpurun_out/j_rc.txt
cat gpurun_out/j_rc.txt; tail -n 3 gpurun_out/j_split.log
tail -n 30 gpurun_out/j_pmc_ops.log | cut -c1-200
cat gpurun_out/bench_table.txt
head -3 gpurun_out/step_table_*.txt
