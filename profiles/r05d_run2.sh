#!/bin/bash
# session 4 of round 5, GPU call 2: kernel sequences / gaps of one steady-state iteration of the tile-culled and dense stage-2 schedules
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trc -o t -- python examples/stage2_schedule.py --sparsify > $O/schedc.log 2>&1
f=$(find $O/trc -name "t_kernel_trace.csv" | head -1)
python profiles/iter_sequence.py $f render_bwd_tile_k > $O/seq_schedc.txt 2>&1
python profiles/gap_analysis.py $f render_bwd_tile_k > $O/gaps_schedc.txt 2>&1
tail -3 $O/schedc.log | cut -c1-600; tail -1 $O/seq_schedc.txt; head -1 $O/gaps_schedc.txt
rm -rf $O/trc
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trd -o t -- python examples/stage2_schedule.py > $O/sched.log 2>&1
f=$(find $O/trd -name "t_kernel_trace.csv" | head -1)
python profiles/iter_sequence.py $f render_bwd_pair_k > $O/seq_sched.txt 2>&1
python profiles/gap_analysis.py $f render_bwd_pair_k > $O/gaps_sched.txt 2>&1
tail -1 $O/seq_sched.txt; head -1 $O/gaps_sched.txt
rm -rf $O/trd
