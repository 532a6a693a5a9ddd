#!/bin/bash
# A/B timing of several builds of the library inside ONE gpurun call (boxes differ by a few % between calls).
# Usage: profiles/ab.sh lib/ab/A.so lib/ab/B.so ...   (paths relative to videoloop3d_amd/); 3 interleaved rounds
for r in 1 2 3; do
  for so in "$@"; do
    VL3D_LIB_PATH=$PWD/videoloop3d_amd/$so timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-loss --no-stage2 2>/dev/null | tail -1 > /tmp/_ab.json
    python - "$so" <<'PY'
import json, sys
d = json.load(open('/tmp/_ab.json'))
print(f"{sys.argv[1]:24s} {d['value']:8.1f} Mpix/s  fwd {d['roofline_fwd']['avg_ms']:.3f} ms  bwd {d['roofline_bwd']['avg_ms']:.3f} ms")
PY
  done
done
