import sys, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/examples')
import __graft_entry__ as g; g.build()
import stage2_schedule as S
order = sys.argv[1].split(',')
for o in order:
    kw = dict(dense={}, dense2=dict(fused=False), culled=dict(sparsify=True), exact=dict(sparsify=True, tile_exact=True), culled2=dict(sparsify=True, fused=False))[o]
    r = S.run(**kw)
    print(o, round(r['iters_per_s'], 1), [round(l['iters_per_s'], 1) for l in r['levels']], flush=True)
