#!/bin/bash
# Build experimental variants of ONE source file into side-by-side libraries (A/B on one box through SODT_LIB_PATH).
#   tools/exp/ab_build.sh gemm3 "-DSODT_EXP_PRIO=1" v1      -> small-object-detection-transformers_amd/libsodt_hip_v1.so
set -e
F=$1; FLAGS=$2; TAG=$3
P=small-object-detection-transformers_amd
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wno-unused-value -Wno-inline-asm $FLAGS -c $P/csrc/$F.hip -o $P/build/${F}_$TAG.o
OBJS=$(for f in $P/csrc/*.hip; do n=$(basename $f .hip); if [ "$n" != "$F" ]; then echo $P/build/$n.o; fi; done)
hipcc -shared -fPIC --offload-arch=gfx950 $OBJS $P/build/${F}_$TAG.o -o $P/libsodt_hip_$TAG.so
echo built $P/libsodt_hip_$TAG.so
