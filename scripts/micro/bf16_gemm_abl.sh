#!/bin/bash
set -euo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
cd "$ROOT"
for dbg in 0 1 2 4 6 7; do
  export DL3P_BF16_DBG=$dbg
  for shape in "131072 304 256 fwd" "131072 256 256 dgrad" "524288 16 64 fwd"; do
    echo -n "DBG=$dbg $shape: "
    bash scripts/ktrace.sh pwb_ -- scripts/micro/bf16_gemm.py $shape 10 | tr '\n' ' '
    echo
  done
done
