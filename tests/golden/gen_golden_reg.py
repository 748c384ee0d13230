#!/usr/bin/env python3
"""Golden vectors for the per-iteration regularisers, produced by
IMPORTING the reference (build container only):

    python tests/golden/gen_golden_reg.py

Follows the reference's own checks in test/test_cuda_utils.py:305-363 (surface normal loss) and
:366-415 (eikonal loss): a 16^3 SDF grid (sphere initialisation + Gaussian noise), a contiguous
block of cells, and the gradient obtained by autograd from the reference's PyTorch implementations
`SparseGrid._surface_normal_loss_grad_check` (svox2.py:5165-5345) and
`SparseGrid._surface_eikonal_loss_grad_check` (svox2.py:5433-5472).  Part of the links is cleared so
that missing neighbours are exercised.  The .npz holds data only.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import import_reference  # noqa: E402


def main():
    torch, svox2 = import_reference()
    rs = np.random.RandomState(21)
    grid = svox2.SparseGrid(
        reso=16, center=[0.0, 0.0, 0.0], radius=[1.0, 1.0, 1.0], basis_dim=1, use_z_order=True,
        device="cpu", background_nlayers=0, basis_type=svox2.BASIS_TYPE_SH,
        surface_type=svox2.SURFACE_TYPE_SDF, use_sphere_bound=False, surface_init="sphere",
        use_octree=False)
    n = grid.surface_data.shape[0]
    surface = (grid.surface_data.data.numpy() + rs.randn(n, 1).astype(np.float32) * 0.3).astype(np.float32)
    grid.surface_data.data[:] = torch.from_numpy(surface)
    links = grid.links.numpy().copy()
    # knock out a few vertices (links = -1) so that cells with missing corners / neighbours occur
    hole = rs.rand(*links.shape) < 0.03
    links[hole] = -1
    grid.links = torch.from_numpy(links.copy())
    grid.level_set_data = torch.tensor([0.0])
    gsz = links.size
    start, num = 700, int(0.6 * gsz)
    rand_cells = ((np.arange(start, start + num)) % gsz).astype(np.int32)
    rc = torch.from_numpy(rand_cells.copy())
    rec = dict(links=links, surface=surface, rand_cells=rand_cells, h=np.float64(float(grid._get_h())))
    meta = {}
    variants = {
        "l1": dict(connectivity_check=False, ignore_empty=False, use_l1=True),
        "l2": dict(connectivity_check=False, ignore_empty=False, use_l1=False),
        "l1_con": dict(connectivity_check=True, ignore_empty=False, use_l1=True),
        "l2_con_empty": dict(connectivity_check=True, ignore_empty=True, use_l1=False),
    }
    for name, kw in variants.items():
        grid.surface_data.grad = None
        loss = grid._surface_normal_loss_grad_check(rc, 1.0, device="cpu", **kw)
        rec["normal_" + name] = grid.surface_data.grad.numpy().astype(np.float32).copy()
        meta["normal_" + name] = dict(kw, scaling=1.0, loss=float(loss) if loss is not None else None)
        print(name, float(np.abs(rec["normal_" + name]).max()))
    grid.surface_data.grad = None
    loss = grid._surface_eikonal_loss_grad_check(rc, 0.7, device="cpu")
    rec["eikonal"] = grid.surface_data.grad.numpy().astype(np.float32).copy()
    meta["eikonal"] = dict(scaling=0.7, loss=float(loss))
    print("eikonal", float(loss), float(np.abs(rec["eikonal"]).max()))
    rec["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "reg_r16.npz"), **rec)


if __name__ == "__main__":
    main()
