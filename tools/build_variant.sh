#!/bin/bash
# Build a variant of libseigen_hip.so with extra -D flags for kernels_mfma.hip (timing experiments):
#   tools/build_variant.sh <name> [-DSG_FLA=6 ...]   ->  build_tools/libseigen_hip_<name>.so   (use with SEIGEN_HIP_LIB=...)
# SRC=kernels_lane tools/build_variant.sh ... rebuilds that kernel file instead.
set -e
NAME=$1; shift
SRC=${SRC:-kernels_mfma}
cd "$(dirname "$0")/../seigen_amd/csrc"
mkdir -p ../../build_tools
# SRC=comm (or another .cpp of the library): pass -x hip among the flags
EXT=hip; [ -f $SRC.hip ] || EXT=cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $SRC.$EXT -o /tmp/kernels_mfma_$NAME.o
OBJS=$(ls *.o | grep -v $SRC.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_tools/libseigen_hip_$NAME.so $OBJS /tmp/kernels_mfma_$NAME.o -ldl
echo built build_tools/libseigen_hip_$NAME.so
