#!/bin/bash
# Build a MEASUREMENT variant of the library: profiles/build_variant.sh NAME [-DFLAG ...]  ->  videoloop3d_amd/lib/ab/NAME.so
# (every csrc/*.hip recompiled with the flags; select it with VL3D_LIB_PATH).  -DVL3D_VARIANTS enables the timing-only ablation
# switches of desc->variant bits 4-7, which the product build compiles out and refuses.
set -e
cd "$(dirname "$0")/.."
N=$1; shift
mkdir -p videoloop3d_amd/lib/ab /tmp/vl3d_$N
for f in videoloop3d_amd/csrc/*.hip; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude -Ivideoloop3d_amd/csrc "$@" -c $f -o /tmp/vl3d_$N/$(basename $f .hip).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/vl3d_$N/*.o -o videoloop3d_amd/lib/ab/$N.so
echo built videoloop3d_amd/lib/ab/$N.so
