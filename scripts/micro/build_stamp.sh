#!/bin/bash
# dev instrument: libdl3p with in-kernel phase stamps in the pointwise GEMM (-DDL3P_STAMP)
cd "$(dirname "$0")/../.."
P=tf-keras-deeplabv3p-model-set_amd
mkdir -p /tmp/dl3p_stamp
for f in $(cd $P/csrc && ls *.hip | sed 's/\.hip$//'); do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off -DDL3P_STAMP -c $P/csrc/$f.hip -o /tmp/dl3p_stamp/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/micro/libdl3p_stamp.so /tmp/dl3p_stamp/*.o
ls -la scripts/micro/libdl3p_stamp.so
