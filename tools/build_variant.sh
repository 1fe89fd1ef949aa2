#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "-DFLAG=.. ..."  → tools/ab/libaukit_NAME.so: the current library with ONE translation unit rebuilt
# with extra flags (A/B on one GPU box: AUKIT_LIB=tools/ab/libaukit_NAME.so python bench.py ...).  Needs aukit_amd/build/*.o (run build() first).
set -e
cd "$(dirname "$0")/.."
NAME=$1; FILE=$2; FLAGS=$3
mkdir -p tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $FLAGS -c aukit_amd/csrc/$FILE -o tools/ab/${FILE%.hip}_$NAME.o
OBJS=$(ls aukit_amd/build/*.o | grep -v "/${FILE%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libaukit_$NAME.so $OBJS tools/ab/${FILE%.hip}_$NAME.o
rm -f tools/ab/${FILE%.hip}_$NAME.o
ls -la tools/ab/libaukit_$NAME.so
