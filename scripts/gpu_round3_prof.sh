#!/bin/bash
# profiles of round 3: headline (kernel stats, FETCH / WRITE, MFMA counters), Xception configs[2], MobileNetV3-Large bf16 configs[4],
# per-launch step tables, the bench table
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
export TMPDIR=/tmp
bash scripts/collect_profiles.sh r03 > gpurun_out/prof_r03.log 2>&1
bash scripts/collect_profiles.sh r03 xception --model xception --batch 4 > gpurun_out/prof_r03_xc.log 2>&1
bash scripts/collect_profiles.sh r03 mobilenetv3large_bf16 --model mobilenetv3large --size 1024 --width 2048 --classes 19 --batch 1 --dtype bf16 > gpurun_out/prof_r03_bf16.log 2>&1
bash scripts/step_table.sh mobilenetv2 > /dev/null 2>&1
bash scripts/step_table.sh xception > /dev/null 2>&1
DL3P_ST_N=1 DL3P_ST_H=1024 DL3P_ST_W=2048 DL3P_ST_C=19 DL3P_ST_DTYPE=bf16 bash scripts/step_table.sh mobilenetv3large > /dev/null 2>&1
mv gpurun_out/step_table_mobilenetv3large.txt gpurun_out/step_table_mobilenetv3large_bf16.txt
bash scripts/bench_table.sh > /dev/null 2>&1
python3 scripts/micro/sb_gemm.py > gpurun_out/split_gemm.txt 2>&1
ls -la gpurun_out/profiles_r03* gpurun_out/*.txt | head -60
cat gpurun_out/bench_table.txt
