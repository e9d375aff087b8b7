#!/bin/bash
# Probe build of the library with one source recompiled under extra macros (the other objects are the product's):
#   bash tools/build_variant.sh <tag> <source.hip> [-DMACRO=..]...   ->  tools/bin/libgdmcf_<tag>.so
# tools/fused_probe.py and friends load it with GDMCF_PROBE_LIB=tools/bin/libgdmcf_<tag>.so (probe scripts only; the product
# always loads gdmcf_amd/csrc/libgdmcf_hip.so).
set -e
cd "$(dirname "$0")/.."
TAG=$1; SRC=$2; shift 2
mkdir -p tools/bin
OBJS=""
for f in gdmcf_amd/csrc/*.o; do
  [ "$(basename $f .o)" = "$(basename $SRC .hip)" ] || OBJS="$OBJS $f"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c gdmcf_amd/csrc/$SRC -o /tmp/variant_$TAG.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libgdmcf_$TAG.so $OBJS /tmp/variant_$TAG.o
echo built tools/bin/libgdmcf_$TAG.so
