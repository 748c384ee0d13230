"""One training iteration of opt.py with surface_cuda_syn.yaml's terms on the bench scene, through the SparseGrid
mirror (Python overhead included): fused render of one 800x800 view (or --rays N shuffled rays), TV on alpha (1 % of
the cells: lambda_tv_alpha 1e-5, tv_sparsity 0.01 -- opt.py:952-957), surface TV, normal loss, alpha sparsify, the
optimizer steps; --eikonal adds the eikonal loss (lambda_surface_eikonal is 0 in that yaml: config_util.py:558).
ms per iteration and rays/s."""
import argparse, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import svox2
from alphasurf_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--rays", type=int, default=0, help="0: the whole 800x800 view in image order; N: N shuffled rays of it")
ap.add_argument("--iters", type=int, default=200,
                help="timed iterations (the first few dozen run behind an empty queue: 30 read 8 % slower than 200)")
ap.add_argument("--eikonal", action="store_true", help="also the eikonal loss (off in surface_cuda_syn.yaml)")
ap.add_argument("--graph-static", action="store_true",
                help="with --graph: replay without copying a batch into the captured buffers (what the copies cost)")
ap.add_argument("--verify", action="store_true",
                help="before the timed iterations: the batch's colours, and the gradients and masks of a sample of its rays "
                     "rendered on their own, against the oracle (CPU restatement)")
ap.add_argument("--graph", action="store_true", help="capture the iteration into one HIP graph (SparseGrid.capture_iteration)")
args = ap.parse_args()
dev = torch.device("cuda:0")
np.random.seed(0)   # (the regularisers draw their runs of cells from numpy's generator, as opt.py does)
sc = synth.make_sparse_scene(256, 9, trunc_a=5.0, **synth.BENCH_SCENE)
grid = svox2.SparseGrid(reso=8, basis_dim=9, device=dev, surface_type=svox2.SURFACE_TYPE_SDF, surface_init="sphere")
n = sc["density"].shape[0]
import torch.nn as nn
grid.links = torch.from_numpy(sc["links"]).to(dev)
grid.density_data = nn.Parameter(torch.from_numpy(sc["density"]).to(dev))
grid.surface_data = nn.Parameter(torch.from_numpy(sc["surface"]).to(dev))
grid.sh_data = nn.Parameter(torch.from_numpy(sc["sh"]).to(dev))
grid.level_set_data = torch.from_numpy(sc["level_set"]).to(dev)
grid.capacity = n
grid.truncated_vol_render_a = 5.0
grid.opt.backend = "surf_trav"
for k, v in synth.SYN_OPTS.items():
    setattr(grid.opt, k, v)
grid.density_rms = grid.sh_rms = grid.surface_rms = None
grid._data_to_links = None
o, d = synth.pinhole_rays(synth.orbit_camera(0), 800, 800)
if args.rays:
    idx = np.random.RandomState(0).permutation(o.shape[0])[: args.rays]
    o, d = o[idx], d[idx]
rays = svox2.Rays(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev))
gt = torch.rand(o.shape[0], 3, device=dev)
grid.requires_grad_(True)
F = synth.SYN_FUSED


def iteration(rays=rays, gt=gt):
    grid.volume_render_fused(rays, gt, lambda_l2=F["lambda_l2"], lambda_l1=F["lambda_l1"],
                             lambda_l_entropy=F["lambda_l_entropy"], lambda_conv_mode_samp=F["lambda_conv_mode_samp"],
                             l_dist_max_sample=F["l_dist_max_sample"], l_di_alpha_thresh=F["l_di_alpha_thresh"],
                             surf_sparse_alpha_thresh=F["surf_sparse_alpha_thresh"],
                             image_width=0 if args.rays else 800)   # (the whole view comes in image order)
    grid.inplace_tv_grad(grid.density_data.grad, scaling=1e-5, sparse_frac=0.01, contiguous=True)
    grid.inplace_tv_surface_grad(grid.surface_data.grad, scaling=1e-3, sparse_frac=1, ignore_edge=True,
                                 edge_value=-1.0, alpha_dependency=True)
    grid.inplace_surface_normal_grad(grid.surface_data.grad, scaling=1e-6, sparse_frac=1, connectivity_check=True,
                                     ignore_empty=False, use_l1=True)
    if args.eikonal:
        grid.inplace_surface_eikonal_grad(grid.surface_data.grad, scaling=1e-4, sparse_frac=1, use_kernel=False)
    grid.inplace_alpha_surf_sparsify_grad(grid.density_data.grad, grid.surface_data.grad, scaling_alpha=1e-9,
                                          scaling_surf=0.0, sparse_frac=0.1)
    grid.optim_density_step(1e-1, beta=0.95, optim="rmsprop")
    grid.optim_surface_step(1e-5, beta=0.95, optim="rmsprop")
    grid.optim_sh_step(1e-2, beta=0.95, optim="rmsprop")


verify = None
if args.verify:
    # the parameters are still the scene's: one fused render of the batch through the mirror against the oracle
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from oracle import oracle as O
    osc = O.make_scene(sc["links"], sc["density"], sc["surface"], sc["sh"], sc["level_set"], sc["offset"], sc["scaling"],
                       9, sc["fake_sample_std"], 5.0)
    n_v = min(1500, o.shape[0])
    vidx = np.linspace(0, o.shape[0] - 1, n_v).astype(np.int64)
    gt_np = gt.cpu().numpy()
    fused_o = dict(lambda_l2=F["lambda_l2"], lambda_l1=F["lambda_l1"], lambda_l_entropy=F["lambda_l_entropy"],
                   lambda_conv_mode_samp=F["lambda_conv_mode_samp"], l_dist_max_sample=F["l_dist_max_sample"],
                   l_di_alpha_thresh=F["l_di_alpha_thresh"], surf_sparse_alpha_thresh=F["surf_sparse_alpha_thresh"])
    rgb_o, go, st_o = O.render_fused(osc, o[vidx], d[vidx], synth.SYN_OPTS, gt_np[vidx], fused_o)
    kw = dict(lambda_l2=F["lambda_l2"], lambda_l1=F["lambda_l1"], lambda_l_entropy=F["lambda_l_entropy"],
              lambda_conv_mode_samp=F["lambda_conv_mode_samp"], l_dist_max_sample=F["l_dist_max_sample"],
              l_di_alpha_thresh=F["l_di_alpha_thresh"], surf_sparse_alpha_thresh=F["surf_sparse_alpha_thresh"])
    rgb_batch = grid.volume_render_fused(rays, gt, **kw)["rgb"][torch.from_numpy(vidx).to(dev)].cpu().numpy()
    for prm in (grid.density_data, grid.surface_data, grid.sh_data):
        prm.grad.zero_()
    vt = torch.from_numpy(vidx).to(dev)
    rv = svox2.Rays(rays.origins[vt].contiguous(), rays.dirs[vt].contiguous())
    rgb_v = grid.volume_render_fused(rv, gt[vt].contiguous(), **kw)["rgb"].cpu().numpy()
    e_grad = {}
    for k, prm in (("sh", grid.sh_data), ("density", grid.density_data), ("surface", grid.surface_data)):
        ref = go[k]
        e_grad[k] = float(np.abs(prm.grad.cpu().numpy().reshape(ref.shape) - ref).max()) / max(float(np.abs(ref).max()), 1e-30)
        prm.grad.zero_()
    m_ok = bool((grid.sparse_grad_indexer.cpu().numpy().astype(bool) == go["mask"].astype(bool)).all())
    e_b, e_v = float(np.abs(rgb_batch - rgb_o).max()), float(np.abs(rgb_v - rgb_o).max())
    verify = dict(ok=bool(e_b <= 2e-5 and e_v <= 2e-5 and max(e_grad.values()) <= 2e-4 and m_ok and int(st_o["n_samples"].sum()) > 0),
                  rays=n_v, samples=int(st_o["n_samples"].sum()), rgb_linf_batch=e_b, rgb_linf=e_v,
                  grad_rel={k: round(v, 8) for k, v in e_grad.items()}, masks_equal=m_ok,
                  checker="oracle (CPU restatement), before the timed iterations")
for _ in range(5):
    iteration()
# (a run of cells that wraps around the end of the volume -- one draw in a hundred -- takes other torch kernels than the
# usual one; their first use loads code for 0.1 s, which is not what an iteration costs)
torch.arange(4, dtype=torch.int64, device=dev).add_(3).remainder_(5).to(torch.int32)
torch.cuda.synchronize()
if args.graph:
    step, _ = grid.capture_iteration(iteration, rays, gt)
    run = (lambda: step()) if args.graph_static else (lambda: step(rays, gt))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
else:
    run = iteration
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import time
t0 = time.perf_counter()
a.record()
for _ in range(args.iters):
    run()
b.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / args.iters * 1e3
ms = a.elapsed_time(b) / args.iters
print(json.dumps(dict(graph=bool(args.graph), graph_static=bool(args.graph_static), rays=o.shape[0], ms_per_iteration=round(ms, 3), wall_ms_per_iteration=round(wall, 3),
                      iterations_per_s=round(1e3 / wall, 1), Mrays_per_s=round(o.shape[0] / wall / 1e3, 2), verify=verify)))
if verify is not None and not verify["ok"]:
    sys.exit("bench_iteration.py: the batch does NOT match the oracle: " + json.dumps(verify))
