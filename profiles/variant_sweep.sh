#!/bin/bash
# Usage: profiles/variant_sweep.sh "6 0 6 0"   -> one line per backward variant: variant, Mpix/s, fwd ms, bwd ms
for v in $1; do
  timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-loss --no-stage2 --variant $v 2>/dev/null | tail -1 > /tmp/_vs.json
  python - "$v" <<'PY'
import json, sys
d = json.load(open('/tmp/_vs.json'))
print(sys.argv[1], d['value'], round(d['roofline_fwd']['avg_ms'],3), round(d['roofline_bwd']['avg_ms'],3))
PY
done
