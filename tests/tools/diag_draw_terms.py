"""One ray of a fuzz draw, the fused loss terms one at a time: which term makes the HIP path and the oracle differ?
usage diag_draw_terms.py <seed> <draw> <ray>   (the ray: tests/tools/diag_draw_rays.py)"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import hip_util as hip
from oracle import oracle as oracle_mod
oracle_mod.build()
import fuzz_util
seed, want, q = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
(it, sc, o, d, opts, fused, gt, tag), = list(fuzz_util.draws(seed, want + 1, only=want))
Q = o.shape[0]
oq, dq, gq = o[q:q + 1], d[q:q + 1], gt[q:q + 1]
def cmp(name, f, op=None):
    op = op or opts
    r1, g1 = hip.render_fused(sc, oq, dq, op, gq, f, global_n_rays=Q)
    r2, g2, _ = oracle_mod.render_fused(sc, oq, dq, op, gq, dict(f, global_n_rays=Q))
    a, b = g1["surface"].reshape(-1), g2["surface"].reshape(-1)
    i = int(np.abs(a - b).argmax())
    print("%-40s max|diff| %.3e at row %d hip %.4e oracle %.4e  (max|oracle| %.3e)" % (name, abs(a[i] - b[i]), i, a[i], b[i], np.abs(b).max()), flush=True)
cmp("all terms", fused)
base = {k: v for k, v in fused.items() if not k.startswith("lambda_") and k not in ("sparsity_loss", "beta_loss")}
cmp("l2 only", dict(base, lambda_l2=1.0))
for k, v in fused.items():
    if k.startswith("lambda_") and k != "lambda_l2":
        cmp("l2 + " + k, dict(base, lambda_l2=1.0, **{k: v}))
        cmp("only " + k, dict(base, lambda_l2=0.0, **{k: v}))
cmp("l2 only, no truncated_vol_render", dict(base, lambda_l2=1.0), dict(opts, truncated_vol_render=False))
cmp("l2 only, both crossings", dict(base, lambda_l2=1.0), dict(opts, only_outward_intersect=False))
