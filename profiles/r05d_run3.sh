#!/bin/bash
# session 4 of round 5, GPU call 3: kernel sequence of one steady-state stage-1 iteration (dense epochs and after the sparsify switch-over)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/s1 -o t -- python examples/stage1_train.py > $O/s1.log 2>&1
f=$(find $O/s1 -name "t_kernel_trace.csv" | head -1)
python profiles/iter_sequence.py $f adam_tiles_k 300 > $O/seq_s1.txt 2>&1
python profiles/gap_analysis.py $f adam_tiles_k > $O/gaps_s1.txt 2>&1
tail -1 $O/seq_s1.txt; head -1 $O/gaps_s1.txt
grep -v "^[EW]2026" $O/s1.log | tail -2 | cut -c1-600
rm -rf $O/s1
