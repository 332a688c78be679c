#!/bin/bash
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/dist_kt
timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-streaming --steps 50 > gpurun_out/o_plain.log 2>&1
DL3P_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --no-other-configs --no-streaming --steps 50 > gpurun_out/o_dist.log 2>&1
DL3P_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dist_kt -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --no-streaming > gpurun_out/o_dist_kt.log 2>&1
python3 scripts/prof_summary.py gpurun_out/dist_kt 14 45 > gpurun_out/o_dist_summary.txt 2>&1
rm -rf gpurun_out/dist_kt
tail -n 1 gpurun_out/o_plain.log | cut -c90-200; tail -n 1 gpurun_out/o_dist.log | cut -c90-200
head -50 gpurun_out/o_dist_summary.txt
