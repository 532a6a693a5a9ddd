#!/bin/bash
# round 6: what each layer-regulariser term costs in INSTRUCTIONS per plane (and frame pair), from hipcc -S of the shipped convention's translation
# unit with the terms compiled out one by one (-DVL3D_REG_ABLATE, measurement only).  CPU only: no GPU needed.
cd "$(dirname "$0")/.."
T=/tmp/vl3d_reg_isa; mkdir -p $T
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Iinclude -Ivideoloop3d_amd/csrc -S --cuda-device-only"
for v in 0 1 3 7 8 24; do
  /opt/rocm/bin/hipcc $F -DVL3D_REG_ABLATE=$v videoloop3d_amd/csrc/vl3d_render_c3_mpv_sig.hip -o $T/c3_$v.s &
done
wait
echo "== forward: render_fwd2x_k (plain, two frames per thread) | render_fwd_reg_k (one frame per thread), main loop = 4 planes per trip (fwd_reg) / 2 planes (fwd2x)"
python profiles/isa_count.py $T/c3_0.s render_fwd2x_kILi1ELi1ELi1ELi1ELi1ELi8ELb0ELb0E loop | head -1
for v in 0 1 3 7; do echo -n "fwd_reg ablate=$v: "; python profiles/isa_count.py $T/c3_$v.s render_fwd_reg_kILi1ELi1ELi1ELi1ELi1ELb0ELb0E loop | head -1; done
echo "== backward: render_bwd_pair_k plain | REG (per plane, two frames), whole kernel body"
python profiles/isa_count.py $T/c3_0.s render_bwd_pair_kILi1ELi1ELi1ELi1ELi1ELb0ELb0ELb0ELi32E loop | head -1
for v in 0 8 24; do echo -n "bwd_pair<REG> ablate=$v: "; python profiles/isa_count.py $T/c3_$v.s render_bwd_pair_kILi1ELi1ELi1ELi1ELi1ELb0ELb1ELb0ELi32E loop | head -1; done
