#!/usr/bin/env python3
"""Golden vectors for the Plenoxels `cuvol` renderer, produced by IMPORTING
the reference (build container only):  python tests/golden/gen_golden_cuvol.py

The procedure of the reference's test/test_render_gradcheck.py (:12-48, :82-125: sigma_thresh = 0,
stop_thresh = 0, white background, SH DC 0.5 + N(0, 0.1), an MSE loss against the ground truth) run
through its pure-PyTorch renderer `SparseGrid._volume_render_gradcheck_lerp` (svox2/svox2.py:1215-1351,
reached through volume_render(use_kernel=False) on a surface-less grid) -- the renderer that test
compares the CUDA kernel against.  Densities are signed so that the relu / sigma_thresh gate acts.
Data only."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import import_reference  # noqa: E402

CASES = {
    "cuvol_r16_sh9": dict(reso=[16, 16, 16], radius=[1.0, 1.0, 1.0], center=[0.0, 0.0, 0.0], basis_dim=9,
                          n_rays=384, seed=41, step_size=0.5),
    "cuvol_r20_sh4": dict(reso=[16, 12, 20], radius=[1.0, 0.8, 1.2], center=[0.0, 0.1, 0.0], basis_dim=4,
                          n_rays=320, seed=42, step_size=0.37),
}


def main():
    torch, svox2 = import_reference()
    for name, c in CASES.items():
        rs = np.random.RandomState(c["seed"])
        grid = svox2.SparseGrid(reso=c["reso"], center=c["center"], radius=c["radius"], basis_dim=c["basis_dim"],
                                use_z_order=False, device="cpu", background_nlayers=0,
                                basis_type=svox2.BASIS_TYPE_SH, surface_type=svox2.SURFACE_TYPE_NONE,
                                use_sphere_bound=True, use_octree=False)
        grid.opt.backend = "cuvol"
        grid.opt.sigma_thresh = 0.0
        grid.opt.stop_thresh = 0.0
        grid.opt.background_brightness = 1.0
        grid.opt.step_size = c["step_size"]
        grid.opt.near_clip = 0.0
        n, B = grid.density_data.shape[0], c["basis_dim"]
        density = (rs.randn(n, 1) * 4.0 + 1.0).astype(np.float32)
        sh = (rs.randn(n, 3 * B) * 0.1).astype(np.float32)
        sh[:, ::B] = 0.5 + rs.randn(n, 3).astype(np.float32) * 0.3
        grid.density_data.data[:] = torch.from_numpy(density)
        grid.sh_data.data[:] = torch.from_numpy(sh)
        origins = (rs.randn(c["n_rays"], 3) * 3).astype(np.float32)
        dirs = rs.randn(c["n_rays"], 3).astype(np.float32)
        # half of the rays aimed at the grid so that most of them cross it
        aim = (rs.uniform(-0.5, 0.5, size=(c["n_rays"], 3)) * np.asarray(c["radius"])).astype(np.float32)
        half = c["n_rays"] // 2
        dirs[:half] = aim[:half] + np.asarray(c["center"], np.float32) - origins[:half]
        dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
        rgb_gt = rs.uniform(0, 1, size=(c["n_rays"], 3)).astype(np.float32)
        grid.requires_grad_(True)
        rays = svox2.Rays(torch.from_numpy(origins.copy()), torch.from_numpy(dirs.copy()))
        rgb = grid.volume_render(rays, use_kernel=False)["rgb"]
        loss = torch.nn.functional.mse_loss(rgb, torch.from_numpy(rgb_gt))
        loss.backward()
        gsz = np.asarray(grid.links.shape, np.float32)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), links=grid.links.numpy(), density=density, sh=sh,
            origins=origins, dirs=dirs, rgb_gt=rgb_gt, rgb=rgb.detach().numpy(),
            grad_density=grid.density_data.grad.numpy(), grad_sh=grid.sh_data.grad.numpy(),
            offset=(grid._offset.numpy() * gsz).astype(np.float32),
            scaling=(grid._scaling.numpy() * gsz).astype(np.float32),
            basis_dim=np.int32(B), step_size=np.float32(c["step_size"]), loss=np.float32(loss.item()))
        hit = np.abs(rgb.detach().numpy() - 1.0).max(axis=1) > 1e-6
        print(name, "rows", n, "links<0", int((grid.links.numpy() < 0).sum()), "rays hitting", int(hit.sum()),
              "loss", float(loss.item()))


if __name__ == "__main__":
    main()
