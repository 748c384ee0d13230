"""Which rays of a fuzz draw make the HIP path and the oracle differ?  Renders the draw ray by ray (global_n_rays = the
draw's ray count, so that every ray's loss terms are scaled as in the whole batch) and lists the rays whose surface /
density / SH gradients differ most: usage diag_draw_rays.py <seed> <draw>."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import hip_util as hip
from oracle import oracle as oracle_mod
oracle_mod.build()
import fuzz_util
seed, want = int(sys.argv[1]), int(sys.argv[2])
(it, sc, o, d, opts, fused, gt, tag), = list(fuzz_util.draws(seed, want + 1, only=want))
Q = o.shape[0]
print(tag[:120], "rays", Q, flush=True)
rgb_o, go, _ = oracle_mod.render_fused(sc, o, d, opts, gt, fused)
rgb, g = hip.render_fused(sc, o, d, opts, gt, fused)
scale = {k: float(np.abs(go[k]).max()) for k in ("sh", "density", "surface")}
print("whole batch:", {k: float(np.abs(g[k].reshape(go[k].shape) - go[k]).max()) / scale[k] for k in scale}, flush=True)
worst = []
for q in range(Q):
    oq, dq, gq = o[q:q + 1], d[q:q + 1], gt[q:q + 1]
    r1, g1 = hip.render_fused(sc, oq, dq, opts, gq, fused, global_n_rays=Q)
    r2, g2, _ = oracle_mod.render_fused(sc, oq, dq, opts, gq, dict(fused, global_n_rays=Q))
    e = {k: float(np.abs(g1[k].reshape(g2[k].shape) - g2[k]).max()) / scale[k] for k in scale}
    worst.append((max(e.values()), q, e, float(np.abs(r1 - r2).max())))
worst.sort(reverse=True)
for w in worst[:6]:
    print(w, flush=True)
q = worst[0][1]
print("ray", q, "o", o[q].tolist(), "d", d[q].tolist())
r1, g1 = hip.render_fused(sc, o[q:q + 1], d[q:q + 1], opts, gt[q:q + 1], fused, global_n_rays=Q)
r2, g2, st = oracle_mod.render_fused(sc, o[q:q + 1], d[q:q + 1], opts, gt[q:q + 1], dict(fused, global_n_rays=Q), max_dump=16)
print('oracle samples: t', st['dump_t'][0], 'alpha', st['dump_alpha'][0], 'weight', st['dump_weight'][0], 'voxel', st['dump_voxel'][0].tolist(), 'flags', st['dump_flags'][0])
for k in ("surface", "density"):
    a, b = g1[k].reshape(-1), g2[k].reshape(-1)
    rows = np.nonzero((a != 0) | (b != 0))[0]
    print(k, "rows touched", len(rows))
    for r in rows[np.argsort(-np.abs(a[rows] - b[rows]))[:12]]:
        print("   row %d hip %.6e oracle %.6e diff %.2e" % (r, a[r], b[r], a[r] - b[r]))
