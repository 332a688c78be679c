#!/bin/bash
set -euo pipefail
# tile rows (64 * MI) x widest column block (16 * NT) of the tiled GEMM, all four roles, one box
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd "$ROOT"
for shape in "266256 304 256" "266256 256 256" "17424 960 320" "17424 576 96"; do
  for cfg in "0 8" "1 8" "1 4" "1 6" "2 4" "2 6" "0 8"; do
    set -- $cfg
    echo -n "MI=$1 NT_MAX=$2: "; DL3P_GEMM_MI=$1 DL3P_GEMM_NT_MAX=$2 python3 scripts/micro/gemm_shape.py $shape 2>&1 | tail -1
  done
done
