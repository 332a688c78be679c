#!/bin/bash
# dev: the row-stationary split GEMM with one ingredient dropped per run (results wrong, time only); needs scripts/micro/libdl3p_abl.so
# (pw_split_rs.hip + pw_split.hip compiled with -DDL3P_SB_ABLATE, linked with the other objects)
#   1 no statistics  2 no MFMAs (fragment reads stay)  3 no staging arithmetic after the first row tile  4 no B stream after the first slots
#   5 no epilogue (stores + statistics)  6 no multiply loop at all  7 no A loads after the first row tile
cd "$(dirname "$0")/../.."
for a in 0 1 2 3 4 5 6 7; do
  echo "== ablate $a"
  DL3P_SB_ABLATE=$a DL3P_LIB_VARIANT=abl SB_SHAPES=${SB_SHAPES:-266256x256x256} timeout 120 python3 scripts/micro/sb_rs.py 2>&1 | grep "^fwd" | sed 's/.*| rs/rs/'
done
