#!/bin/bash
# round 6: run the bench's stage-2 legs until a stalled level shows, print its interval diagnostics (device vs host clock)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-10}); do
  VL3D_BENCH_DETAIL=/tmp/detail.json timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-loss > /dev/null 2>&1
  python - <<PY
import json
d = json.load(open("/tmp/detail.json"))["stage2_schedule"]
for k, v in d.items():
    for l in v["levels"]:
        m = l["device_ms_per_iter"]
        if m["max"] > 20 * m["p50"] or m["host_max"] > 100 or m["reserved_changes_in_loop"]:
            print("run", $i, k, l["frame"], round(l["iters_per_s"], 1), {a: (round(b, 2) if isinstance(b, float) else b) for a, b in m.items()}, flush=True)
print("run", $i, "done", {k: round(v["iters_per_s"], 1) for k, v in d.items()}, flush=True)
PY
done
