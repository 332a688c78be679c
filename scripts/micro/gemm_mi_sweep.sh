#!/bin/bash
# tile rows (64 * MI) x resident workgroups per CU of the tiled GEMM on the decoder shapes, one box
cd "$GRAFT_REPO_ROOT"
for shape in "266256 304 256" "266256 256 256" "17424 960 320"; do
  for cfg in "0 0" "1 3" "1 4" "1 5" "2 2" "0 0"; do
    set -- $cfg
    echo -n "MI=$1 PER_CU=$2: "; DL3P_GEMM_MI=$1 DL3P_GEMM_PER_CU=$2 python3 scripts/micro/gemm_shape.py $shape 2>&1 | tail -1
  done
done
for e in "X=0" "DL3P_GEMM_PER_CU=4" "DL3P_GEMM_LONG_ROWS=1000000000" "X=0"; do echo -n "bench $e: "; env $e python3 bench.py --no-cpu-baseline --steps 50 --warmup 10 2>&1 | tail -1 | cut -c165-195; done
