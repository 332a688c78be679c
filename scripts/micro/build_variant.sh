#!/bin/bash
# dev: a second copy of libdl3p built with extra compiler flags, for A/B runs next to the in-tree library
#   bash scripts/micro/build_variant.sh nodma -DDL3P_GEMM_BDMA=0     -> scripts/micro/libdl3p_nodma.so
# scripts that honour DL3P_LIB_VARIANT=<name> (micro/gemm_shape.py) load it instead of the in-tree build.
cd "$(dirname "$0")/../.."
NAME=$1; shift
P=tf-keras-deeplabv3p-model-set_amd
mkdir -p /tmp/dl3p_$NAME
for f in $(cd $P/csrc && ls *.hip | sed 's/\.hip$//'); do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off "$@" -c $P/csrc/$f.hip -o /tmp/dl3p_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/micro/libdl3p_$NAME.so /tmp/dl3p_$NAME/*.o
ls -la scripts/micro/libdl3p_$NAME.so
