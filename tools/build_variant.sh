#!/bin/bash
# usage: [SRC=decode] build_variant.sh <name> [extra hipcc flags ...]
# Builds jpeg-encoder_amd/libjpegenc_amd_<name>.so: the library with $SRC.hip (default fdct) compiled with the extra flags
# (-D switches of experiments, -fno-slp-vectorize, ...) and every other object as built by the Makefile.
# tools/ab_k1.py times all such builds against the real one in one process; delete them afterwards.
set -e
cd "$(dirname "$0")/.."
name="$1"; shift
C=jpeg-encoder_amd/csrc
make -s -C $C
mkdir -p $C/build/variants
SRC="${SRC:-fdct}"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function "$@" -c $C/$SRC.hip -o $C/build/variants/${SRC}_$name.o
objs=""
for o in fdct entropy decode blocks capi; do
  if [ "$o" = "$SRC" ]; then objs="$objs $C/build/variants/${SRC}_$name.o"; else objs="$objs $C/build/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o jpeg-encoder_amd/libjpegenc_amd_$name.so
echo built jpeg-encoder_amd/libjpegenc_amd_$name.so
