#!/bin/bash
# kernel sequence of one steady-state iteration of the stage-1 loop and of the tile-culled / dense stage-2 schedules (profiles/iter_sequence.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/seq; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/s1 -o t -- python examples/stage1_train.py > $O/s1.log 2>&1
python profiles/iter_sequence.py $O/s1/t_kernel_trace.csv adam_tiles_k 300 > $O/s1_sequence.txt
rm -rf $O/s1; tail -2 $O/s1_sequence.txt
