#!/bin/bash
# Round 5, session 3: tile-culled kernels -- tests, the full-frame legs (profiles/cull_lean.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/cull; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_render.py tests/test_gpu_optim.py tests/test_gpu_render_video.py tests/test_gpu_reference_modules.py -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python profiles/cull_lean.py > $O/cull_lean.txt 2>&1; tail -6 $O/cull_lean.txt
