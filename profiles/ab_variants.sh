#!/bin/bash
# A/B of backward kernel variants inside ONE process-external loop of one gpurun call (boxes differ by a few % between calls):
#   profiles/ab_variants.sh "0 5 3" [rounds]     -> fwd / bwd ms per variant, interleaved rounds
V=${1:-"0 5 3"}; R=${2:-3}
for r in $(seq $R); do
  for v in $V; do
    timeout 300 python bench.py --steps 10 --warmup 3 --variant $v --no-cpu-baseline --no-loss --no-stage2 2>/dev/null | tail -1 > /tmp/_ab.json
    python - "$v" <<'PY'
import json, sys
d = json.load(open('/tmp/_ab.json'))
print(f"variant {sys.argv[1]:>4s} {d['value']:8.1f} Mpix/s  fwd {d['roofline_fwd']['avg_ms']:.3f} ms  bwd {d['roofline_bwd']['avg_ms']:.3f} ms", flush=True)
PY
  done
done
