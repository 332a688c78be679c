#!/bin/bash
# GPU box: production pointwise GEMM next to its compile-time ablations (build_variant.sh abl100/101/102 first)
for v in "" abl102 abl100; do
  echo "== variant '${v:-production}'"
  for s in "266256 304 256" "266256 256 304" "4356 728 728" "17424 960 160" "17424 1280 256"; do
    DL3P_LIB_VARIANT=$v python3 scripts/micro/gemm_shape.py $s
  done
done
