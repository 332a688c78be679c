#!/bin/bash
set -euo pipefail
# mean kernel time of the bf16 GEMM family on the configs[4] shapes, for DL3P_BF16_MI = 1 and 2
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
export TMPDIR=/tmp
cd "$ROOT"
for mi in 1 2; do
  export DL3P_BF16_MI=$mi
  for shape in "131072 304 256" "131072 256 256" "8192 672 112" "8192 112 672" "8192 1280 256" "524288 16 64" "32768 120 40"; do
    for mode in fwd dgrad wgrad; do
      echo -n "MI=$mi $shape $mode: "
      bash scripts/ktrace.sh pwb_ -- scripts/micro/bf16_gemm.py $shape $mode 10 | tr '\n' ' '
      echo
    done
  done
done
