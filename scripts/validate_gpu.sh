#!/bin/bash
# what the driver runs at round end: the whole GPU suite, smoke(), the default bench line
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$ROOT"
mkdir -p gpurun_out
[ "$1" = "--release" ] && export DL3P_RELEASE_TESTS=1
( time timeout 2400 python -m pytest tests -q -m gpu ) > gpurun_out/final_tests.log 2>&1; echo "tests rc=$?" > gpurun_out/final_rc.txt
timeout 300 python __graft_entry__.py smoke > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/final_rc.txt
( time timeout 900 python bench.py ) > gpurun_out/final_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/final_rc.txt
cat gpurun_out/final_rc.txt; tail -n 8 gpurun_out/final_tests.log; tail -n 2 gpurun_out/final_smoke.log; tail -n 6 gpurun_out/final_bench.log | cut -c1-4000
