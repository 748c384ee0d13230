"""Fraction of the grid's rows one 800x800 view of the bench scene touches, and the union over the first N of the bench's
eight poses (the bytes of the exchange keyed by rows)."""
import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import hip_util as H
from alphasurf_amd import csrc_api as _C, synth
sc = synth.make_sparse_scene(256, 9, trunc_a=5.0, **synth.BENCH_SCENE)
g, op = H.grid_spec(sc), H.opt_spec(synth.SYN_OPTS)
F = dict(synth.SYN_FUSED)
args = [F.get(k, dflt) for k, dflt in H.FUSED_ORDER]
masks = []
for k in range(8):
    o, d = synth.pinhole_rays(synth.orbit_camera(k), 800, 800)
    r = H.rays_spec(o, d)
    gr = H.zero_grads(g, with_std=False)
    gt = torch.rand(o.shape[0], 3, device="cuda")
    rgb = torch.zeros_like(gt)
    _C.volume_render_surf_trav_fused(g, r, op, gt, *args, rgb, gr, global_n_rays=8 * o.shape[0])
    masks.append(gr.mask_out.clone())
n = masks[0].numel()
print("rows", n, "one view:", [round(float(m.float().mean()), 3) for m in masks])
for N in (2, 4, 8):
    u = torch.zeros_like(masks[0])
    for m in masks[:N]:
        u |= m
    print(f"union of {N} views: {float(u.float().mean()):.3f}")
