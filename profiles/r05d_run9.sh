#!/bin/bash
# session 4 of round 5, GPU call 9: the four-texel owner table of a single frame -- tests, cfg2 in process, stage-1 loop
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_render.py tests/test_gpu_mpv.py tests/test_gpu_stage1_driver.py tests/test_gpu_fuzz.py -x -q -n 4 > $O/tests_o4.txt 2>&1; tail -3 $O/tests_o4.txt
python profiles/ab_inproc.py --T 1 --variants 3,0,2,5 --rounds 8 --reps 10 > $O/cfg2_owner4.txt 2>&1; tail -8 $O/cfg2_owner4.txt
for r in 1 2; do python examples/stage1_train.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage1_train', round(d['iters_per_s']), 'dense', round(d['iters_per_s_dense_epochs']), 'sparsified', round(d['iters_per_s_sparsified_epochs']))"; done
