#!/bin/bash
# A/B of the cfg3 render kernels between libraries of several commits (alternating processes on one box): profiles/r05d_ab_lib.sh lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
for r in ${ROUNDS:-1 2}; do
  for so in "$@"; do
    echo "== round $r $so"
    VL3D_LIB_PATH=$PWD/videoloop3d_amd/lib/$so VL3D_ALLOW_MISSING_SYMBOLS=1 python profiles/ab_inproc.py --variants 0 --rounds 4 --reps 3 2>/dev/null | tail -1
  done
done > $O/ab_lib.txt 2>&1
cat $O/ab_lib.txt
