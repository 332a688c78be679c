#!/bin/bash
# everything profiles/ holds for a round, in one GPU call:  bash scripts/collect_round.sh r04
#   headline bench + kernel stats + FETCH / WRITE / MFMA counters (collect_profiles.sh), step tables of configs[1] / [3] / [4]
#   (rocprofv3 kernel trace of an eagerly replayed step), op-level counters of configs[2] / [3] / [4] and the headline's decoder GEMMs
set -uo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
R=${1:-r04}
bash scripts/collect_profiles.sh $R > gpurun_out/collect_$R.log 2>&1
bash scripts/step_table.sh mobilenetv2 > /dev/null 2>&1; cp gpurun_out/step_table_mobilenetv2.txt gpurun_out/${R}_step_table_mobilenetv2.txt
DL3P_ST_N=2 DL3P_ST_H=769 DL3P_ST_C=19 DL3P_ST_OS=8 bash scripts/step_table.sh xception > /dev/null 2>&1; cp gpurun_out/step_table_xception.txt gpurun_out/${R}_step_table_xception769_os8.txt
DL3P_ST_N=4 bash scripts/step_table.sh xception > /dev/null 2>&1; cp gpurun_out/step_table_xception.txt gpurun_out/${R}_step_table_xception.txt
DL3P_ST_N=1 DL3P_ST_H=1024 DL3P_ST_W=2048 DL3P_ST_C=19 DL3P_ST_DTYPE=bf16 bash scripts/step_table.sh mobilenetv3large > /dev/null 2>&1; cp gpurun_out/step_table_mobilenetv3large.txt gpurun_out/${R}_step_table_mobilenetv3large_bf16.txt
bash scripts/pmc_ops.sh $R > gpurun_out/pmc_ops_$R.log 2>&1
ls -la gpurun_out/profiles_$R gpurun_out/pmc_ops_$R gpurun_out/${R}_step_table_*
