"""A fuzz draw against the oracle's own sensitivity (tests/test_hip_parity.py: oracle_sensitivity): which gradient
elements differ from the oracle by more than the bound, and does the oracle move as much there under perturbations of
1e-6?  usage diag_draw_sensitivity.py <seed> <draw>"""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import hip_util as hip
from oracle import oracle as oracle_mod
oracle_mod.build()
import fuzz_util
import test_hip_parity as T
seed, want = int(sys.argv[1]), int(sys.argv[2])
(it, sc, o, d, opts, fused, gt, tag), = list(fuzz_util.draws(seed, want + 1, only=want))
rgb, g = hip.render_fused(sc, o, d, opts, gt, fused)
rgb_o, go, st = oracle_mod.render_fused(sc, o, d, opts, gt, fused)
moved = T.oracle_sensitivity(oracle_mod, sc, o, d, opts, fused, gt, go)
for k in ("sh", "density", "surface"):
    scale = float(np.abs(go[k]).max())
    err = np.abs(g[k].reshape(go[k].shape) - go[k])
    over = err > 3e-4 * scale
    bad = over & (err > 1.1 * moved[k] + 3e-4 * scale)
    print(k, "scale %.3e over %d bad %d" % (scale, int(over.sum()), int(bad.sum())))
    idx = np.argwhere(bad)[:8]
    for i in idx:
        i = tuple(i)
        print("   ", i, "err %.3e moved %.3e hip %.4e oracle %.4e" % (err[i], moved[k][i], g[k].reshape(go[k].shape)[i], go[k][i]))
