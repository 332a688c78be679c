#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/bf16_backward_parity.jsonl
( time python -m pytest tests -q -m gpu --timeout 3000 -x ) > gpurun_out/f_all.log 2>&1; echo "all rc=$?" > gpurun_out/f_rc.txt
python bench.py --no-cpu-baseline --steps 50 > gpurun_out/f_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/f_rc.txt
cat gpurun_out/f_rc.txt
tail -n 12 gpurun_out/f_all.log
tail -n 1 gpurun_out/f_bench.log | cut -c1-6000
