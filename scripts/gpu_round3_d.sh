#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_split_gemm_gpu.py -q -x --timeout 1200 > gpurun_out/d_split.log 2>&1; echo "split rc=$?" > gpurun_out/d_rc.txt
python scripts/micro/sb_gemm.py > gpurun_out/d_sb_gemm.txt 2>&1; echo "sbgemm rc=$?" >> gpurun_out/d_rc.txt
rm -f gpurun_out/bf16_backward_parity.jsonl
python -m pytest tests/test_bf16_gpu.py -q --timeout 1800 -k "train_step_bf16" > gpurun_out/d_bf16.log 2>&1; echo "bf16 rc=$?" >> gpurun_out/d_rc.txt
cat gpurun_out/d_rc.txt
tail -n 4 gpurun_out/d_split.log; tail -n 3 gpurun_out/d_bf16.log
cat gpurun_out/d_sb_gemm.txt
