#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_split_gemm_gpu.py -q -x --timeout 1200 > gpurun_out/b_split.log 2>&1; echo "split rc=$?" > gpurun_out/b_rc.txt
python scripts/micro/sb_gemm.py > gpurun_out/b_sb_gemm.txt 2>&1; echo "sbgemm rc=$?" >> gpurun_out/b_rc.txt
rm -f gpurun_out/bf16_backward_parity.jsonl
python -m pytest tests/test_bf16_gpu.py -q --timeout 1800 -k "train_step_bf16" > gpurun_out/b_bf16.log 2>&1; echo "bf16 rc=$?" >> gpurun_out/b_rc.txt
python -m pytest tests/test_ops_gpu.py tests/test_production_shapes_gpu.py -q --timeout 1800 -k "dwconv or depthwise" > gpurun_out/b_dw.log 2>&1; echo "dw rc=$?" >> gpurun_out/b_rc.txt
python bench.py --no-cpu-baseline --steps 50 > gpurun_out/b_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/b_rc.txt
cat gpurun_out/b_rc.txt
tail -n 4 gpurun_out/b_split.log; tail -n 3 gpurun_out/b_bf16.log; tail -n 3 gpurun_out/b_dw.log
cat gpurun_out/b_sb_gemm.txt
tail -n 1 gpurun_out/b_bench.log | cut -c1-2500
