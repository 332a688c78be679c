#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/bf16_backward_parity.jsonl
python -m pytest tests/test_bf16_gpu.py -q --timeout 1800 -k "train_step_bf16" > gpurun_out/g_bf16.log 2>&1; echo "bf16 rc=$?" > gpurun_out/g_rc.txt
for i in 1 2; do
for v in new old; do
  if [ $v = old ]; then export DL3P_LIB_OVERRIDE=$ROOT/scripts/micro/libdl3p_finold.so; else unset DL3P_LIB_OVERRIDE; fi
  python bench.py --no-cpu-baseline --steps 60 --no-other-configs --no-streaming > gpurun_out/g_bench_$v$i.log 2>&1
  echo "$v$i $(tail -n 1 gpurun_out/g_bench_$v$i.log | cut -c100-190)"
done
done
unset DL3P_LIB_OVERRIDE
cat gpurun_out/g_rc.txt; tail -n 3 gpurun_out/g_bf16.log
