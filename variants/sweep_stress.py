"""k_sweep against the per-voxel walk on many rays: RGB is a deterministic function of a ray's list of work voxels,
so it must come out bit for bit (gradients: to float atomics' reordering).  Scenes: the 256^3 bench scene and a 130^3
non-cubic crop; rays: random directions from random origins, shuffled pixel rays of 8 cameras, rays through lattice
points.  usage: sweep_stress.py [n_random_rays]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import hip_util
from alphasurf_amd import lib as _lib, synth

n_rand = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
rs = np.random.RandomState(1)
bad = 0
for reso, crop in ((256, None), (130, (125, 111, 68))):
    sc = synth.make_sparse_scene(reso, 4, trunc_a=5.0, **dict(synth.BENCH_SCENE, seed=5 + reso))
    if crop:
        sc = dict(sc, links=np.ascontiguousarray(sc["links"][: crop[0], : crop[1], : crop[2]]))
    o1, d1 = synth.random_rays(n_rand, 3)
    cams = [synth.pinhole_rays(synth.orbit_camera(k), 200, 200) for k in range(8)]
    o2 = np.concatenate([c[0] for c in cams]); d2 = np.concatenate([c[1] for c in cams])
    p = rs.permutation(o2.shape[0]); o2, d2 = o2[p], d2[p]
    n = 100000
    tgt = rs.randint(0, reso, size=(n, 3)).astype(np.float64)
    src = rs.randint(-reso, 2 * reso, size=(n, 3)).astype(np.float64)
    src[n // 2:] += rs.rand(n - n // 2, 3)
    keep = np.linalg.norm(tgt - src, axis=1) > 1
    o3 = ((src[keep] - reso / 2) / (reso / 2)).astype(np.float32)
    d3 = tgt[keep] - src[keep]
    d3 = (d3 / np.linalg.norm(d3, axis=1, keepdims=True)).astype(np.float32)
    o = np.concatenate([o1, o2, o3]).astype(np.float32); d = np.concatenate([d1, d2, d3]).astype(np.float32)
    gt = rs.rand(o.shape[0], 3).astype(np.float32)
    fused = dict(lambda_l2=1.0, lambda_l_entropy=1e-4, lambda_conv_mode_samp=1e-6)
    for extra in ({}, dict(near_clip=0.4), dict(use_spheric_clip=True)):
        opts = dict(synth.SYN_OPTS, **extra)
        _lib.config_set("ASURF_DEBUG", 8); _lib.config_set("ASURF_WALK_SEG", 1)
        rgb_ref, g_ref = hip_util.render_fused(sc, o, d, opts, gt, fused)
        _lib.config_set("ASURF_DEBUG"); _lib.config_set("ASURF_WALK_SEG")
        rgb, g = hip_util.render_fused(sc, o, d, opts, gt, fused)
        # (a ray handed to the one-kernel path -- more candidates than its key buffer holds -- sums its colours in
        # another order than k_comp: an ulp; a ray with a different work-voxel list differs by far more)
        dif = np.abs(rgb - rgb_ref).max(axis=1)
        neq = int((dif > 5e-7).sum())
        n_ulp = int(((dif > 0) & (dif <= 5e-7)).sum())
        mask_eq = bool((g["mask"] == g_ref["mask"]).all())
        gerr = max(float(np.abs(g[k] - g_ref[k]).max() / max(np.abs(g_ref[k]).max(), 1e-20)) for k in ("sh", "density", "surface"))
        print(f"reso {reso} crop {crop} {extra}: {o.shape[0]} rays, rays with different RGB {neq} (+{n_ulp} within an ulp, max {dif.max():.1e}), masks equal {mask_eq}, "
              f"max grad err {gerr:.2e}", flush=True)
        bad += neq + (0 if mask_eq else 1) + (1 if gerr > 1e-4 else 0)
print("STRESS", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
