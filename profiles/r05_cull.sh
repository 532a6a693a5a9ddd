#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/cull; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_optim.py -x -q > $O/test_optim.log 2>&1; tail -3 $O/test_optim.log
timeout 600 python profiles/cull_lean.py > $O/cull_lean.txt 2>&1; tail -6 $O/cull_lean.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python profiles/cull_lean.py --rounds 3 > $O/prof.log 2>&1
cp $O/prof/t_kernel_stats.csv $O/kernel_stats_cull.csv; rm -rf $O/prof; head -12 $O/kernel_stats_cull.csv | cut -c1-200
