"""One-off stress: voxel lists of the HIP hierarchical walk (Markstein quotients, brick jumps) against the
oracle's division-based DDA, bit for bit, on millions of rays."""
import os, sys, time
import numpy as np
_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _root)
sys.path.insert(0, os.path.join(_root, "tests"))
from alphasurf_amd import synth
from oracle import oracle as O
import hip_util as hip
O.build()
tot = 0
for seed in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    reso = [96, 128, 160, 72][seed % 4]
    sc = synth.make_sparse_scene(reso, 1, seed=seed, band=2.0, dilate=1)
    n = 400000
    o, d = synth.random_rays(n, 1000 + seed)
    rs = np.random.RandomState(seed)
    # a share of rays from lattice points towards lattice points (ties), and nearly axis-parallel ones
    k = n // 8
    src = rs.randint(-reso, 2 * reso, size=(k, 3)).astype(np.float64)
    tgt = rs.randint(0, reso, size=(k, 3)).astype(np.float64)
    keep = np.linalg.norm(tgt - src, axis=1) > 1
    dd = tgt - src
    dd /= np.maximum(np.linalg.norm(dd, axis=1, keepdims=True), 1e-9)
    o[:k][keep] = ((src[keep] - reso / 2) / (reso / 2)).astype(np.float32)
    d[:k][keep] = dd[keep].astype(np.float32)
    d[k:2 * k, 2] *= 1e-4
    d[k:2 * k] /= np.linalg.norm(d[k:2 * k], axis=1, keepdims=True)
    opts = dict(sigma_thresh=-1e4, stop_thresh=-1e4)
    t0 = time.time()
    a = hip.voxel_lists(sc, o, d, opts, skip_empty=False)
    t1 = time.time()
    b = O.trace_voxel_lists(sc, o, d, opts)
    t2 = time.time()
    ok = all(np.array_equal(x, y) for x, y in zip(a, b))
    tot += a[0].shape[0]
    print(f"seed {seed} reso {reso}: {a[0].shape[0]} voxels, equal={ok}, hip {t1 - t0:.1f}s oracle {t2 - t1:.1f}s", flush=True)
    assert ok
print("total voxels compared", tot)
