#!/bin/bash
# session 4 of round 5, GPU call 1: cfg2 region-height sweep (bit check + in-process timing), loss iteration traces + gap analysis
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
python profiles/ab_inproc.py --T 1 --variants 3,0,2,7,6,5 --rounds 8 --reps 10 > $O/cfg2_rows.txt 2>&1
tail -12 $O/cfg2_rows.txt
for c in ref other; do
  python profiles/loss_ref_trace.py $c 30 > $O/loss_$c.txt 2>&1; tail -1 $O/loss_$c.txt
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$c -o t -- python profiles/loss_ref_trace.py $c 30 > $O/tr_$c.log 2>&1
  f=$(find $O/tr_$c -name "t_kernel_trace.csv" | head -1)
  python profiles/gap_analysis.py $f patchnn > $O/gaps_$c.txt 2>&1; cat $O/gaps_$c.txt
  python profiles/iter_sequence.py $f patchnn > $O/seq_$c.txt 2>&1 || true
  rm -rf $O/tr_$c
done
