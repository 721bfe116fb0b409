#!/bin/bash
# usage: tools/build_x3t_variant.sh <name> "<extra flags>"  -> 3dhumangan_amd/csrc/libh3d_<name>.so with field_x3t.hip and
# synthesis_x3t.hip recompiled under the extra flags (development experiments: H3D_LIB=... python bench.py)
set -e
cd "$(dirname "$0")/../3dhumangan_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -I../../include -Wno-inline-asm"
hipcc $F $2 -c field_x3t.hip -o /tmp/v_$1_field_x3t.o &
hipcc $F $2 -c synthesis_x3t.hip -o /tmp/v_$1_synthesis_x3t.o &
wait
objs=$(ls *.o | grep -v "^field_x3t.o$" | grep -v "^synthesis_x3t.o$")
hipcc -shared -fPIC --offload-arch=gfx950 -o libh3d_$1.so $objs /tmp/v_$1_field_x3t.o /tmp/v_$1_synthesis_x3t.o
echo built libh3d_$1.so
