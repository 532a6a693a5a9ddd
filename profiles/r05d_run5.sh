#!/bin/bash
# session 4 of round 5, GPU call 5: stage-1 loop with the crop-aware optimiser now that the head is fused
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
for r in 1 2; do
 for m in off on; do
  python examples/stage1_train.py --crop-aware-adam $m > $O/s1_ca_${m}_$r.json 2>$O/s1_ca_${m}_$r.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/s1_ca_${m}_$r.json").read().strip().splitlines()[-1])
    print("crop-aware $m", "it/s", round(d["iters_per_s"]), "dense", round(d["iters_per_s_dense_epochs"]), "sparsified", round(d["iters_per_s_sparsified_epochs"]), "140 epochs", round(d["projected_140_epochs_s"], 2), "s")
except Exception as e:
    print("crop-aware $m failed", e); print(open("$O/s1_ca_${m}_$r.err").read()[-1500:])
PY
 done
done
