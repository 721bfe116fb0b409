#!/bin/bash
# usage: tools/build_x3_variant2.sh <name> "<extra flags for field_x3.hip>" "<extra flags for synthesis_x3.hip>"
#   -> 3dhumangan_amd/csrc/libh3d_<name>.so (development experiments: H3D_LIB=... python bench.py); an empty flag string reuses the
#   object of the regular build
set -e
cd "$(dirname "$0")/../3dhumangan_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -I../../include -Wno-inline-asm"
fo=field_x3.o; so=synthesis_x3.o
if [ -n "$2" ]; then fo=/tmp/v_$1_field_x3.o; hipcc $F $2 -c field_x3.hip -o $fo & fi
if [ -n "$3" ]; then so=/tmp/v_$1_synthesis_x3.o; hipcc $F $3 -c synthesis_x3.hip -o $so & fi
wait
objs=$(ls *.o | grep -v "^field_x3.o$" | grep -v "^synthesis_x3.o$")
hipcc -shared -fPIC --offload-arch=gfx950 -o libh3d_$1.so $objs $fo $so
echo built libh3d_$1.so
