#!/bin/bash
# dev: a copy of libdl3p with ONE source rebuilt under extra flags (the other objects are the in-tree build's)
#   bash scripts/micro/build_one_variant.sh imm pw_split3 -DS3_ABL_IMMEDIATE   -> scripts/micro/libdl3p_imm.so  (DL3P_LIB_OVERRIDE=that path)
cd "$(dirname "$0")/../.."
NAME=$1; SRC=$2; shift 2
P=tf-keras-deeplabv3p-model-set_amd
mkdir -p /tmp/dl3p_one_$NAME
EXTRA=""
case $SRC in irb_fwd|irb_bwd) EXTRA="-mllvm -amdgpu-mfma-vgpr-form";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $EXTRA "$@" -c $P/csrc/$SRC.hip -o /tmp/dl3p_one_$NAME/$SRC.o || exit 1
OBJS=$(ls $P/build/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/micro/libdl3p_$NAME.so $OBJS /tmp/dl3p_one_$NAME/$SRC.o
ls -la scripts/micro/libdl3p_$NAME.so
