#!/bin/bash
# usage: build_variant.sh <name> [extra hipcc flags for fdct.hip ...]
# Builds jpeg-encoder_amd/libjpegenc_amd_<name>.so: the library with fdct.hip compiled with the extra flags
# (-D switches of experiments, -fno-slp-vectorize, ...) and every other object as built by the Makefile.
# tools/ab_k1.py times all such builds against the real one in one process; delete them afterwards.
set -e
cd "$(dirname "$0")/.."
name="$1"; shift
C=jpeg-encoder_amd/csrc
make -s -C $C
mkdir -p $C/build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -Wno-unused-function "$@" -c $C/fdct.hip -o $C/build/variants/fdct_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC $C/build/variants/fdct_$name.o $C/build/entropy.o $C/build/decode.o $C/build/blocks.o $C/build/capi.o -o jpeg-encoder_amd/libjpegenc_amd_$name.so
echo built jpeg-encoder_amd/libjpegenc_amd_$name.so
