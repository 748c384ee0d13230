"""Replays one draw of tests/test_hip_parity.py::test_random_option_fuzz (same RNG stream) and compares the HIP path with
the oracle under several internal configurations: usage fuzz_draw.py <seed> <draw>."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import hip_util as hip
from alphasurf_amd import lib as _lib, synth
from oracle import oracle as oracle_mod
oracle_mod.build()

import fuzz_util
seed, want = int(sys.argv[1]), int(sys.argv[2])
for it, sc, o, d, opts, fused, gt, tag in fuzz_util.draws(seed, want + 1, only=want):
    print(tag, flush=True)
    rgb_o, go, st = oracle_mod.render_fused(sc, o, d, opts, gt, fused)
    for name, env in (("default", {}), ("no filter", {"ASURF_DEBUG": "32"}), ("one-kernel replay", {"ASURF_NO_SPLIT_BWD": "1"}),
                      ("walking backward", {"ASURF_NO_REPLAY": "1"}), ("no pipeline", {"ASURF_NO_PIPELINE": "1"}),
                      ("k_walk", {"ASURF_WALK": "step"})):
        for k, v in env.items():
            _lib.config_set(k, v)
        rgb, g = hip.render_fused(sc, o, d, opts, gt, fused)
        for k in env:
            _lib.config_set(k)
        line = [name, "rgb %.2e" % float(np.abs(rgb - rgb_o).max())]
        for k in ("sh", "density", "surface"):
            scale = max(float(np.abs(go[k]).max()), 1e-30)
            e = np.abs(g[k] - go[k])
            line.append("%s %.2e (at %d, ref %.3e)" % (k, float(e.max()) / scale, int(e.argmax()), float(go[k].reshape(-1)[e.argmax()])))
        print("  ".join(line), flush=True)
