"""Times the N2 regulariser kernels on the bench scene with the live config's cell fractions
(opt/configs/surface_cuda_syn.yaml: tv_sparsity 0.01, tv_surface_sparsity 1, norm_surface_sparsity 1).
Prints one JSON object; bytes/cell is the compulsory traffic of the reference kernel:
cell id 4 + 4 links 16 + 4 values 16 + 4 read-modify-write gradients 32 + 4 mask bytes (TV);
32 links + 32 values x 4 cells ... (normal): stated per kernel below."""
import json, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alphasurf_amd import csrc_api as _C, synth

dev = torch.device("cuda:0")
sc = synth.make_sparse_scene(256, 9, trunc_a=5.0, **synth.BENCH_SCENE)
links = torch.from_numpy(sc["links"]).to(dev)
dens = torch.from_numpy(sc["density"]).to(dev)
surf = torch.from_numpy(sc["surface"]).to(dev)
n = dens.shape[0]
mask = torch.zeros(n, dtype=torch.bool, device=dev)
gd, gs = torch.zeros_like(dens), torch.zeros_like(surf)
gsz = links.numel()
block = torch.arange(1000, 1000 + int(0.01 * gsz), dtype=torch.int32, device=dev)
non_empty = torch.where(links.view(-1) >= 0)[0].int()


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


out = {}
cases = {
    "tv_grad_sparse(density, 1% block)": (lambda: _C.tv_grad_sparse(links, dens, block, mask, 0, 1, 1e-5, False, 2.0, False, False, -1, -1, gd), block.numel(), 72),
    "surf_tv_grad_sparse(all non-empty)": (lambda: _C.surf_tv_grad_sparse(links, surf, dens, non_empty, mask, 0, 1, 1e-3, True, -1.0, False, -1, -1, True, gs), non_empty.numel(), 88),
    "surface_normal_grad_sparse(all non-empty, L1)": (lambda: _C.surface_normal_grad_sparse(links, surf, non_empty, mask, 0.0, 0, 1, 1e-6, 0.0, -1, -1, True, False, True, gs), non_empty.numel(), 4 + 4 * 64 + 6 * 8 * 9),
    "alpha_surf_sparsify_grad_sparse(10% non-empty)": (lambda: _C.alpha_surf_sparsify_grad_sparse(links, dens, surf, non_empty[: n // 10].contiguous(), mask, 1e-3, 1e-3, True, 0.01, 1e-6, -1.0, gd, gs), n // 10, 4 + 4 + 8 + 16 + 1),
    "surface_eikonal_grad_sparse(all non-empty)": (lambda: _C.surface_eikonal_grad_sparse(links, surf, non_empty, None, 1e-4, 2.0 / 256, gs), non_empty.numel(), 4 + 16 + 16 + 32),
}
for name, (fn, cells, bpc) in cases.items():
    ms = timeit(fn)
    out[name] = dict(ms=round(ms, 4), cells=int(cells), Gcells_per_s=round(cells / ms / 1e6, 2),
                     model_bytes_per_cell=bpc, model_GBps=round(cells * bpc / ms / 1e6, 1))
print(json.dumps(out, indent=1))
