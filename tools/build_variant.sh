#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "-DFLAG=.. ..."  → tools/variants/libaukit_NAME.so: the current library with ONE translation unit rebuilt
# with extra flags (A/B on one GPU box: AUKIT_LIB=tools/variants/libaukit_NAME.so python bench.py ...).  Needs aukit_amd/build/*.o (run build() first).
set -e
cd "$(dirname "$0")/.."
NAME=$1; FILE=$2; FLAGS=$3
mkdir -p tools/variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $FLAGS -c aukit_amd/csrc/$FILE -o tools/variants/${FILE%.hip}_$NAME.o
OBJS=$(ls aukit_amd/build/*.o | grep -v "/${FILE%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/libaukit_$NAME.so $OBJS tools/variants/${FILE%.hip}_$NAME.o
rm -f tools/variants/${FILE%.hip}_$NAME.o
ls -la tools/variants/libaukit_$NAME.so
