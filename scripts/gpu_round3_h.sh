#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
timeout 300 python scripts/micro/sb_gemm.py > gpurun_out/h_sb_gemm.txt 2>&1; echo "sbgemm rc=$?" > gpurun_out/h_rc.txt
timeout 600 python -m pytest tests/test_split_gemm_gpu.py -q -x --timeout 300 > gpurun_out/h_split.log 2>&1; echo "split rc=$?" >> gpurun_out/h_rc.txt
timeout 900 python -m pytest tests/test_split_model_gpu.py -q --timeout 800 > gpurun_out/h_splitmodel.log 2>&1; echo "splitmodel rc=$?" >> gpurun_out/h_rc.txt
rm -f gpurun_out/bf16_backward_parity.jsonl
timeout 600 python -m pytest tests/test_bf16_gpu.py tests/test_augment.py -q --timeout 500 > gpurun_out/h_bf16.log 2>&1; echo "bf16+aug rc=$?" >> gpurun_out/h_rc.txt
for sg in 0 1; do
timeout 300 python bench.py --no-cpu-baseline --steps 50 --no-other-configs --no-streaming --split-gemm $sg > gpurun_out/h_bench_mnv2_$sg.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --steps 30 --no-other-configs --model xception --batch 4 --split-gemm $sg > gpurun_out/h_bench_xc_$sg.log 2>&1
done
cat gpurun_out/h_rc.txt
tail -n 3 gpurun_out/h_split.log; tail -n 3 gpurun_out/h_splitmodel.log; tail -n 3 gpurun_out/h_bf16.log
grep -E "^fwd|^dgbn" gpurun_out/h_sb_gemm.txt | cut -c1-260
for f in gpurun_out/h_bench_*.log; do echo $f; tail -n 1 $f | cut -c90-200; done
