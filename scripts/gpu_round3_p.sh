#!/bin/bash
# split-bf16 GEMMs after the unconditional-prefetch change: parity (op + whole model), step time with the switch on
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_split_gemm_gpu.py tests/test_split_model_gpu.py -x -q -m gpu > gpurun_out/p_test.log 2>&1; echo "test rc=$?"
tail -3 gpurun_out/p_test.log
timeout 900 python3 bench.py --steps 60 --warmup 15 --no-other-configs > gpurun_out/p_bench0.log 2>&1; echo "bench rc=$?"
timeout 900 python3 bench.py --steps 60 --warmup 15 --no-other-configs --split-gemm 1 > gpurun_out/p_bench1.log 2>&1; echo "bench split rc=$?"
python3 - <<'PY'
import json
for f in ('gpurun_out/p_bench0.log','gpurun_out/p_bench1.log'):
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); print(f, d['value'], d['ms_per_step'], d['config'].get('gemm'))
PY
