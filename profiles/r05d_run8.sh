#!/bin/bash
# session 4 of round 5, GPU call 8: frames of a clip rendered in place (vl3d_render_fwd_frames) -- tests and the offline renderer's rates
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_render_video.py tests/test_gpu_mpv.py -x -q > $O/tests_rv.txt 2>&1; tail -3 $O/tests_rv.txt
python examples/render_video.py > $O/render_video.json 2> $O/render_video.err; tail -c 1200 $O/render_video.json
