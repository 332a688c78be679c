#!/bin/bash
set -euo pipefail
# usage (GPU box, repo root): bash scripts/step_table.sh [model]  -> gpurun_out/step_table_<model>.txt
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
export TMPDIR=/tmp
cd "$ROOT"
M=${1:-mobilenetv2}
rm -rf gpurun_out/st
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/st -- python3 scripts/step_table.py run $M > gpurun_out/st_run.log 2>&1
python3 scripts/step_table.py parse gpurun_out/st > gpurun_out/step_table_$M.txt 2> gpurun_out/st_parse.log
rm -rf gpurun_out/st
tail -40 gpurun_out/step_table_$M.txt; tail -5 gpurun_out/st_parse.log
