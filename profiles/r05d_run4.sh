#!/bin/bash
# session 4 of round 5, GPU call 4: the fused stage-1 objective -- tests, A/B of the driver loop, kernel sequence
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_stage1_driver.py tests/test_gpu_mpv.py -x -q > $O/tests_s1.txt 2>&1; tail -3 $O/tests_s1.txt
for r in 1 2; do
  python examples/stage1_train.py > $O/s1_fused_$r.json 2>/dev/null; python examples/stage1_train.py --generic-objective > $O/s1_generic_$r.json 2>/dev/null
  python - <<PY
import json
for n in ("fused", "generic"):
    d = json.loads(open("$O/s1_%s_$r.json" % n).read().strip().splitlines()[-1])
    print(n, "it/s", round(d["iters_per_s"]), "dense", round(d["iters_per_s_dense_epochs"]), "sparsified", round(d["iters_per_s_sparsified_epochs"]), "140 epochs", round(d["projected_140_epochs_s"], 2), "s")
PY
done
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/s1 -o t -- python examples/stage1_train.py > $O/s1.log 2>&1
f=$(find $O/s1 -name "t_kernel_trace.csv" | head -1)
python profiles/iter_sequence.py $f adam_tiles_k 300 > $O/seq_s1_fused.txt 2>&1
python profiles/gap_analysis.py $f adam_tiles_k > $O/gaps_s1_fused.txt 2>&1
tail -1 $O/seq_s1_fused.txt; head -1 $O/gaps_s1_fused.txt
rm -rf $O/s1
