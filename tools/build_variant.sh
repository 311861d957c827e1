#!/bin/bash
# Build an A/B variant of the engine: tools/build_variant.sh <tag> <file.hip> [-DFLAG=..]...  -> retrieval-scaling_amd/csrc/librsx_<tag>.so
# (the named kernel file recompiled with the extra flags, every other object taken from the regular build; loaded with RSX_LIB=<path>)
set -e
cd "$(dirname "$0")/../retrieval-scaling_amd/csrc"
tag=$1; src=$2; shift 2
make -s -j8
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed "$@" -c "$src" -o "/tmp/${src%.hip}.$tag.o"
objs=""
for o in k_select.o k_misc.o k_gemm.o k_pq.o k_pq_rot.o api_build.o api_search.o api_sharded.o rsx_api.o; do
  if [ "$o" = "${src%.hip}.o" ]; then objs="$objs /tmp/${src%.hip}.$tag.o"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "librsx_$tag.so" $objs
echo "built librsx_$tag.so"
