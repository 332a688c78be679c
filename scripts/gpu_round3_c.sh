#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_split_model_gpu.py -q --timeout 2400 > gpurun_out/c_splitmodel.log 2>&1; echo "splitmodel rc=$?" > gpurun_out/c_rc.txt
rm -f gpurun_out/bf16_backward_parity.jsonl
python -m pytest tests/test_bf16_gpu.py -q --timeout 1800 -k "train_step_bf16" > gpurun_out/c_bf16.log 2>&1; echo "bf16 rc=$?" >> gpurun_out/c_rc.txt
for sg in 0 1; do
python bench.py --no-cpu-baseline --steps 50 --no-other-configs --no-streaming --split-gemm $sg > gpurun_out/c_bench_mnv2_$sg.log 2>&1
python bench.py --no-cpu-baseline --steps 30 --no-other-configs --model xception --batch 4 --split-gemm $sg > gpurun_out/c_bench_xc_$sg.log 2>&1
python bench.py --no-cpu-baseline --steps 20 --no-other-configs --model resnet50 --split-gemm $sg > gpurun_out/c_bench_rn_$sg.log 2>&1
done
cat gpurun_out/c_rc.txt
tail -n 5 gpurun_out/c_splitmodel.log; tail -n 3 gpurun_out/c_bf16.log
for f in gpurun_out/c_bench_*.log; do echo $f; tail -n 1 $f | cut -c1-330; done
