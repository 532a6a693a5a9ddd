#!/bin/bash
# kernel sequence of one steady-state iteration of the tile-culled stage-2 schedule (tile-exact layout with $1 = --tile-exact)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trc -o t -- python examples/stage2_schedule.py --sparsify $1 > $O/schedc.log 2>&1
f=$(find $O/trc -name "t_kernel_trace.csv" | head -1)
python profiles/iter_sequence.py $f render_bwd_tile_k > $O/seq_schedc$1.txt 2>&1
rm -rf $O/trc; cut -c1-150 $O/seq_schedc$1.txt
