#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/seq; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/schedc -o t -- python examples/stage2_schedule.py --sparsify --epochs 1 > $O/schedc.log 2>&1
python profiles/iter_sequence.py $O/schedc/t_kernel_trace.csv render_bwd_tile_k > $O/schedc_sequence.txt
python profiles/gap_analysis.py $O/schedc/t_kernel_trace.csv render_bwd_tile_k > $O/schedc_gaps.txt
rm -rf $O/schedc; tail -3 $O/schedc_sequence.txt; head -3 $O/schedc_gaps.txt
