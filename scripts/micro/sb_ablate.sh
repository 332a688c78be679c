#!/bin/bash
# what bounds the split-bf16 GEMM?  a build with -DDL3P_SB_ABLATE whose launches drop one ingredient (results wrong, time only):
# 1 split arithmetic, 2 MFMAs, 3 A global loads, 4 epilogue, 5 B LDS stores, 6 A LDS stores, 7 B global loads
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
cd "$ROOT"
P=tf-keras-deeplabv3p-model-set_amd
mkdir -p /tmp/sbab
for f in $(cd $P/csrc && ls *.hip | sed 's/\.hip$//'); do
  if [ $f = pw_split ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -DDL3P_SB_ABLATE -c $P/csrc/$f.hip -o /tmp/sbab/$f.o || exit 1
  else
    cp $P/build/$f.o /tmp/sbab/$f.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/sbab/libdl3p_ab.so /tmp/sbab/*.o || exit 1
for m in 0 1 2 3 4 5 6 7; do
  echo "== ablate $m"
  DL3P_LIB_OVERRIDE=/tmp/sbab/libdl3p_ab.so DL3P_SB_ABLATE=$m python3 scripts/micro/sb_gemm.py ${1:-8} ${2:-2} ${3:-0} 2>&1 | grep -E "^fwd" | head -3 | cut -c1-140
done
