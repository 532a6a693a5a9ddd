#!/bin/bash
# session 4 of round 5, GPU call 6: the one-pass Adam skipping untouched texels -- tests, stage-1 loop, bench legs of stage 1
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mpv.py tests/test_gpu_stage1_driver.py tests/test_gpu_optim.py -x -q > $O/tests_adam.txt 2>&1; tail -3 $O/tests_adam.txt
for r in 1 2; do
  python examples/stage1_train.py > $O/s1_skip_$r.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open("$O/s1_skip_$r.json").read().strip().splitlines()[-1])
print("zero-skip Adam", "it/s", round(d["iters_per_s"]), "dense", round(d["iters_per_s_dense_epochs"]), "sparsified", round(d["iters_per_s_sparsified_epochs"]), "140 epochs", round(d["projected_140_epochs_s"], 2), "s")
PY
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s1k -o t -- python examples/stage1_train.py > $O/s1k.log 2>&1
f=$(find $O/s1k -name "t_kernel_stats.csv" | head -1); head -8 $f | cut -c1-200; cp $f $O/kernel_stats_s1train_skip.csv; rm -rf $O/s1k
