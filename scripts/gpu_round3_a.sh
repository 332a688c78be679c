#!/bin/bash
# first GPU pass of round 3: the new parity tests, then the bench line with configs[2..4]
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
python -m pytest tests/test_tuned_tables_gpu.py -q -m gpu -x --timeout 1200 > gpurun_out/a_tuned.log 2>&1; echo "tuned rc=$?" > gpurun_out/a_rc.txt
python -m pytest tests/test_dist_gpu.py -q -m gpu --timeout 1800 > gpurun_out/a_dist.log 2>&1; echo "dist rc=$?" >> gpurun_out/a_rc.txt
python -m pytest tests/test_bf16_gpu.py -q -m gpu --timeout 1800 > gpurun_out/a_bf16.log 2>&1; echo "bf16 rc=$?" >> gpurun_out/a_rc.txt
python -m pytest tests/test_production_shapes_gpu.py -q -m gpu --timeout 2400 > gpurun_out/a_prod.log 2>&1; echo "prod rc=$?" >> gpurun_out/a_rc.txt
python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -q -m gpu --timeout 1800 -k "moving or dropout_stream or batchnorm_fwd_bwd" > gpurun_out/a_misc.log 2>&1; echo "misc rc=$?" >> gpurun_out/a_rc.txt
python bench.py > gpurun_out/a_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/a_rc.txt
cat gpurun_out/a_rc.txt
tail -3 gpurun_out/a_tuned.log gpurun_out/a_dist.log gpurun_out/a_bf16.log gpurun_out/a_prod.log gpurun_out/a_misc.log
tail -1 gpurun_out/a_bench.log | cut -c1-3000
