#!/bin/bash
# round 6: the full-resolution level of the tile-culled stage-2 schedule, shared-border stack against the tile-exact layout (kernel stats)
R=$GRAFT_REPO_ROOT
cat > /tmp/r06_lvl.py <<PY
import sys; sys.path.insert(0, "$R"); sys.path.insert(0, "$R/examples")
import json, stage2_schedule as S
te = sys.argv[1] == "exact"
r = S.run(levels=1, epochs=1, sparsify=True, tile_exact=te)
print(json.dumps({"mode": sys.argv[1], "it_s": r["iters_per_s"], "frac": r["roofline_iter"]["frac"], "stack": r["levels"][0]["stack"]}))
PY
for mode in lattice exact; do
  TOPN=14 bash $R/profiles/kstats.sh r06_schedc_$mode python /tmp/r06_lvl.py $mode
done
