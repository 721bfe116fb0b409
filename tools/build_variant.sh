#!/bin/bash
# usage: tools/build_variant.sh <name> <file.hip> "<extra flags>"  -> 3dhumangan_amd/csrc/libh3d_<name>.so (development experiments)
set -e
cd "$(dirname "$0")/../3dhumangan_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -I../../include $3 -c $2 -o /tmp/variant_$1.o
objs=$(ls *.o | grep -v "^${2%.hip}.o$")
hipcc -shared -fPIC --offload-arch=gfx950 -o libh3d_$1.so $objs /tmp/variant_$1.o
echo built libh3d_$1.so
