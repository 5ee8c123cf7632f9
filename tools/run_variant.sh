#!/bin/bash
# usage: run_variant.sh <variant .so> <python script> [args...]; runs the script with the variant in place of the built library
set -e
cp jpeg-encoder_amd/libjpegenc_amd.so /tmp/lib_orig.so
cp "$1" jpeg-encoder_amd/libjpegenc_amd.so
shift
python "$@" 2>/dev/null || true
cp /tmp/lib_orig.so jpeg-encoder_amd/libjpegenc_amd.so
