"""Times the cuvol renderer (N4) on the bench scene: one 800x800 view of the 256^3 grid per call.
Densities are the bench scene's raw opacities x60 (sigma ~ 30: sigma * world_step ~ 0.12 per sample)."""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphasurf_amd import csrc_api as _C, synth

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=800)
ap.add_argument("--no-image-width", action="store_true", help="do not tell the calls that the batch is an image (asurf_rays::image_width)")
args = ap.parse_args()
dev = torch.device("cuda:0")
sc = synth.make_sparse_scene(256, 9, trunc_a=5.0, **synth.BENCH_SCENE)
g = _C.SparseGridSpec()
g.density_data = torch.from_numpy(sc["density"] * 60.0).to(dev)
g.sh_data, g.links = torch.from_numpy(sc["sh"]).to(dev), torch.from_numpy(sc["links"]).to(dev)
g._offset, g._scaling = torch.from_numpy(sc["offset"]), torch.from_numpy(sc["scaling"])
g.basis_dim, g.basis_type, g.surface_type = 9, 1, 100
o, d = synth.pinhole_rays(synth.orbit_camera(0), args.width, args.width)
r = _C.RaysSpec()
r.origins, r.dirs = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
r.masks = torch.ones(o.shape[0], dtype=torch.bool, device=dev)
r.image_width = 0 if args.no_image_width else args.width
opt = _C.RenderOptions()
opt.step_size, opt.sigma_thresh, opt.stop_thresh, opt.background_brightness = 0.5, 1e-8, 1e-7, 1.0
gt = torch.rand(o.shape[0], 3, device=dev)
rgb = torch.zeros_like(gt)
gr = _C.GridOutputGrads()
gr.grad_density_out, gr.grad_sh_out = torch.zeros_like(g.density_data), torch.zeros_like(g.sh_data)
gr.mask_out = torch.zeros(g.density_data.shape[0], dtype=torch.bool, device=dev)
cases = {
    "forward": lambda: _C.volume_render_cuvol(g, r, opt),
    "fused fwd+bwd": lambda: _C.volume_render_cuvol_fused(g, r, opt, gt, 0.0, 0.0, rgb, gr),
    "fused fwd+bwd (beta, sparsity)": lambda: _C.volume_render_cuvol_fused(g, r, opt, gt, 0.01, 1e-3, rgb, gr),
}
out = {}
for name, fn in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    out[name] = dict(ms=round(ms, 3), Mrays_per_s=round(o.shape[0] / ms / 1e3, 1))
out["hit_fraction"] = round(float(((rgb - 1.0).abs().amax(dim=1) > 1e-4).float().mean()), 3)
print(json.dumps(out, indent=1))
