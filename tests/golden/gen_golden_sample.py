#!/usr/bin/env python3
"""Golden vectors for point sampling, produced by IMPORTING the reference
(build container only):  python tests/golden/gen_golden_sample.py

The reference's pure-PyTorch branch of SparseGrid.sample (svox2.py:1063-1112; the branch its own
test/test_sample.py compares the CUDA kernel against): sampled density / SH at random world points and
the gradients of a seeded linear functional of them.  Data only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import import_reference  # noqa: E402


def main():
    torch, svox2 = import_reference()
    rs = np.random.RandomState(31)
    grid = svox2.SparseGrid(reso=[16, 12, 20], center=[0.0, 0.1, 0.0], radius=[1.0, 0.8, 1.2], basis_dim=4,
                            use_z_order=False, device="cpu", background_nlayers=0,
                            basis_type=svox2.BASIS_TYPE_SH, surface_type=svox2.SURFACE_TYPE_NONE,
                            use_sphere_bound=True, use_octree=False)
    n = grid.density_data.shape[0]
    density = rs.randn(n, 1).astype(np.float32)
    sh = rs.randn(n, 12).astype(np.float32)
    grid.density_data.data[:] = torch.from_numpy(density)
    grid.sh_data.data[:] = torch.from_numpy(sh)
    pts = rs.uniform(-1.4, 1.4, size=(4000, 3)).astype(np.float32)     # some outside: border padding
    wd = rs.randn(4000, 1).astype(np.float32)
    ws = rs.randn(4000, 12).astype(np.float32)
    grid.requires_grad_(True)
    sd, ss = grid.sample(torch.from_numpy(pts.copy()), use_kernel=False)
    ((sd * torch.from_numpy(wd)).sum() + (ss * torch.from_numpy(ws)).sum()).backward()
    gsz = np.asarray(grid.links.shape, np.float32)
    gpts = rs.uniform(-0.5, 16.0, size=(1000, 3)).astype(np.float32)
    gd, gs = grid.sample(torch.from_numpy(gpts.copy()), use_kernel=False, grid_coords=True)
    np.savez_compressed(
        os.path.join(HERE, "sample_r16.npz"), links=grid.links.numpy(), density=density, sh=sh, points=pts,
        wd=wd, ws=ws, out_density=sd.detach().numpy(), out_sh=ss.detach().numpy(),
        grad_density=grid.density_data.grad.numpy(), grad_sh=grid.sh_data.grad.numpy(),
        offset=(grid._offset.numpy() * gsz).astype(np.float32), scaling=(grid._scaling.numpy() * gsz).astype(np.float32),
        grid_points=gpts, grid_out_density=gd.detach().numpy(), grid_out_sh=gs.detach().numpy())
    print("rows", n, "links<0", int((grid.links.numpy() < 0).sum()), float(np.abs(sd.detach().numpy()).max()))


if __name__ == "__main__":
    main()
