#!/bin/bash
# Build an experimental variant of the library for profiles/ab.sh: profiles/build_variant.sh NAME [-DFLAG ...]
# -> videoloop3d_amd/lib/ab/NAME.so (vl3d_render.hip recompiled with the flags, the other objects reused).
set -e
cd "$(dirname "$0")/.."
N=$1; shift
mkdir -p videoloop3d_amd/lib/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude -Ivideoloop3d_amd/csrc "$@" \
    -c videoloop3d_amd/csrc/vl3d_render.hip -o /tmp/vl3d_render_$N.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/vl3d_render_$N.o videoloop3d_amd/lib/vl3d_loss.o videoloop3d_amd/lib/vl3d_ops.o \
    -o videoloop3d_amd/lib/ab/$N.so
echo built videoloop3d_amd/lib/ab/$N.so
