import os, sys
import numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import svox2
from alphasurf_amd import csrc_api as _C, lib as _lib
_lib.config_set("ASURF_DEBUG", 16)
g = svox2.SparseGrid(reso=32, center=[0.0, 0.0, 0.0], radius=[1.0, 1.0, 1.0], basis_dim=4, device="cuda:0",
                     basis_type=svox2.BASIS_TYPE_SH, surface_type=svox2.SURFACE_TYPE_SDF, surface_init="sphere",
                     use_sphere_bound=False)
g.opt.backend = "surf_trav"; g.opt.step_size = 0.5; g.opt.sigma_thresh = -20.0
g.level_set_data = torch.tensor([0.0], device="cuda:0")
rs = np.random.RandomState(0)
links = g.links.clone(); links[0:14] = -1; g.links = links
with torch.no_grad():
    g.density_data.data[:] = torch.from_numpy(rs.normal(2.0, 0.3, size=(g.capacity, 1)).astype(np.float32)).cuda()
    g.sh_data.data[:] = torch.from_numpy(rs.normal(0, 0.3, size=(g.capacity, 12)).astype(np.float32)).cuda()
    g.surface_data.data += torch.from_numpy((rs.rand(g.capacity, 1).astype(np.float32) - 0.5) * 0.2).cuda()
rr = np.random.RandomState(100)
o = rr.randn(600, 3).astype(np.float32); o = 2.5 * o / np.linalg.norm(o, axis=1, keepdims=True)
t = (rr.rand(600, 3).astype(np.float32) - 0.5) * 0.8
d = t - o; d /= np.linalg.norm(d, axis=1, keepdims=True)
rays = svox2.Rays(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda())
spec = g._to_cpp(); ep = spec.links_epoch
outs = []
for e in (0, 0, 0, ep, ep, ep, 0):
    spec.links_epoch = e
    outs.append(_C.volume_render_surf_trav(spec, rays._to_cpp(), g.opt._to_cpp()))
    torch.cuda.synchronize()
for i, x in enumerate(outs):
    dd = (x - outs[0]).abs().max(dim=1).values
    print(i, int((dd > 0).sum()), float(dd.max()), torch.nonzero(dd > 0).flatten()[:10].tolist())
