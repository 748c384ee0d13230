"""HIP regulariser kernels through the C ABI, against the CPU oracle and the
goldens produced by the reference's PyTorch implementations.  `pytest -m gpu`."""
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN_DIR, grad_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    from alphasurf_amd import csrc_api as _C, lib
    lib.lib()
    return torch, _C


@pytest.fixture(scope="module")
def reg():
    z = dict(np.load(os.path.join(GOLDEN_DIR, "reg_r16.npz")))
    meta = json.loads(bytes(z.pop("meta")).decode())
    return z, meta


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("name", ["l1", "l2", "l1_con", "l2_con_empty"])
def test_normal_loss_golden(env, oracle, reg, name):
    torch, _C = env
    z, meta = reg
    m = meta["normal_" + name]
    links, surf, cells = dev(torch, z["links"]), dev(torch, z["surface"]), dev(torch, z["rand_cells"])
    grad = torch.zeros_like(surf)
    mask = torch.zeros(surf.shape[0], dtype=torch.bool, device="cuda")
    _C.surface_normal_grad_sparse(links, surf, cells, mask, 0.0, 0, 1, m["scaling"], 0.0, -1.0, -1.0,
                                  m["connectivity_check"], m["ignore_empty"], m["use_l1"], grad)
    torch.cuda.synchronize()
    g = grad.cpu().numpy()
    grad_close(g, z["normal_" + name], 2e-4, name + " vs reference")
    go, mo = oracle.surface_normal_grad_sparse(z["links"], z["surface"], z["rand_cells"], m["scaling"],
                                               con_check=m["connectivity_check"],
                                               ignore_empty=m["ignore_empty"], use_l1=m["use_l1"])
    grad_close(g, go, 1e-5, name + " vs oracle")
    assert (mask.cpu().numpy() == (mo > 0)).all()


def test_eikonal_golden(env, oracle, reg):
    torch, _C = env
    z, meta = reg
    links, surf, cells = dev(torch, z["links"]), dev(torch, z["surface"]), dev(torch, z["rand_cells"])
    grad = torch.zeros_like(surf)
    loss = _C.surface_eikonal_grad_sparse(links, surf, cells, None, meta["eikonal"]["scaling"], float(z["h"]), grad)
    torch.cuda.synchronize()
    grad_close(grad.cpu().numpy(), z["eikonal"], 2e-5, "eikonal vs reference")
    assert abs(float(loss.item()) - meta["eikonal"]["loss"]) <= 1e-4 * abs(meta["eikonal"]["loss"])


def sparse_scene(reso=96, seed=5):
    from alphasurf_amd import synth
    sc = synth.make_sparse_scene(reso, 4, seed=seed)
    rs = np.random.RandomState(seed)
    n_all = sc["links"].size
    non_empty = np.nonzero(sc["links"].reshape(-1) >= 0)[0].astype(np.int32)
    start = rs.randint(0, n_all)
    block = ((np.arange(start, start + n_all // 3)) % n_all).astype(np.int32)
    return sc, block, non_empty


@pytest.mark.parametrize("which", ["tv_density", "tv_sh", "tv_surface", "tv_surface_edge"])
def test_tv_kernels_vs_oracle(env, oracle, which):
    torch, _C = env
    sc, block, non_empty = sparse_scene()
    links = dev(torch, sc["links"])
    n = sc["density"].shape[0]
    mask = torch.zeros(n, dtype=torch.bool, device="cuda")
    if which == "tv_density":
        data, cells = sc["density"], block
        grad = torch.zeros(n, 1, device="cuda")
        _C.tv_grad_sparse(links, dev(torch, data), dev(torch, cells), mask, 0, 1, 1e-2, False, 2.0, False, False,
                          -1.0, -1.0, grad)
        go, mo = oracle.tv_grad_sparse(sc["links"], data, cells, 1e-2)
    elif which == "tv_sh":
        data, cells = sc["sh"], block
        grad = torch.zeros_like(dev(torch, data))
        _C.tv_grad_sparse(links, dev(torch, data), dev(torch, cells), mask, 3, 9, 1e-2, False, 2.0, True, False,
                          -1.0, -1.0, grad)
        go, mo = oracle.tv_grad_sparse(sc["links"], data, cells, 1e-2, start_dim=3, end_dim=9, ignore_edge=True)
    else:
        data, cells = sc["surface"], non_empty
        dens = (sc["density"] * 0.2).astype(np.float32)   # raw alphas around 0.1: both branches of the boost
        edge = which == "tv_surface_edge"
        grad = torch.zeros(n, 1, device="cuda")
        _C.surf_tv_grad_sparse(links, dev(torch, data), dev(torch, dens), dev(torch, cells), mask, 0, 1, 1e-3,
                               not edge, 5.0 if edge else -1.0, False, -1.0, -1.0, True, grad)
        go, mo = oracle.tv_grad_sparse(sc["links"], data, cells, 1e-3, surf=True, density=dens,
                                       edge_value=5.0 if edge else -1.0, ignore_edge=not edge,
                                       alpha_dependency=True)
    torch.cuda.synchronize()
    assert np.abs(go).max() > 0
    # rsqrtf vs 1/sqrtf, and the order of the float sums: row 0 stands for every neighbour outside the grid and takes
    # thousands of terms (loss_kernel.cu:765-770) -- the oracle adds them one after the other in float32, the kernel
    # per block first (RegAcc, asurf_reg.h): 2.7e-5 of the largest entry apart on this scene
    grad_close(grad.cpu().numpy(), go, 4e-5 if which in ("tv_density", "tv_sh") else 2e-5, which)
    assert (mask.cpu().numpy() == (mo > 0)).all()


def test_sparsify_and_normal_on_sparse_scene(env, oracle):
    torch, _C = env
    sc, block, non_empty = sparse_scene(seed=8)
    links = dev(torch, sc["links"])
    n = sc["density"].shape[0]
    alpha = np.abs(sc["density"]).astype(np.float32)
    ga, gs = torch.zeros(n, 1, device="cuda"), torch.zeros(n, 1, device="cuda")
    mask = torch.zeros(n, dtype=torch.bool, device="cuda")
    _C.alpha_surf_sparsify_grad_sparse(links, dev(torch, alpha), dev(torch, sc["surface"]), dev(torch, non_empty),
                                       mask, 0.3, 0.1, True, 0.6, 0.45, -0.5, ga, gs)
    oa, os_, om = oracle.alpha_surf_sparsify_grad_sparse(sc["links"], alpha, sc["surface"], non_empty, 0.3, 0.1,
                                                         True, 0.6, 0.45, -0.5)
    torch.cuda.synchronize()
    np.testing.assert_allclose(ga.cpu().numpy(), oa, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(gs.cpu().numpy(), os_, rtol=1e-6, atol=1e-9)
    assert (mask.cpu().numpy() == (om > 0)).all()
    # normal loss on the sparse shell, every flag combination
    for con, emp, l1 in ((True, False, True), (True, True, False), (False, True, True)):
        grad = torch.zeros(n, 1, device="cuda")
        _C.surface_normal_grad_sparse(links, dev(torch, sc["surface"]), dev(torch, non_empty), None, 0.0, 0, 1,
                                      1.0, 0.0, -1.0, -1.0, con, emp, l1, grad)
        go, _ = oracle.surface_normal_grad_sparse(sc["links"], sc["surface"], non_empty, 1.0, con_check=con,
                                                  ignore_empty=emp, use_l1=l1)
        torch.cuda.synchronize()
        grad_close(grad.cpu().numpy(), go, 2e-5, f"normal {con} {emp} {l1}")


def test_grid_methods(env, oracle):
    """SparseGrid.inplace_* as opt.py calls them (opt.py:938-1062) with the live config's weights."""
    torch, _C = env
    import svox2
    from test_grid_api_gpu import make_grid
    grid = make_grid(torch, svox2, 32, 4)
    n = grid.capacity
    for p in (grid.density_data, grid.surface_data, grid.sh_data):
        p.grad = torch.zeros_like(p.data)
    grid.sparse_grad_indexer = torch.zeros(n, dtype=torch.bool, device="cuda")
    grid.sparse_sh_grad_indexer = torch.zeros(n, dtype=torch.bool, device="cuda")
    np.random.seed(0)
    grid.inplace_tv_grad(grid.density_data.grad, scaling=1e-5, sparse_frac=0.01, contiguous=True)
    assert grid.density_data.grad.abs().sum().item() > 0
    grid.inplace_tv_surface_grad(grid.surface_data.grad, scaling=1e-3, sparse_frac=1, ignore_edge=True,
                                 edge_value=-1.0, alpha_dependency=True)
    g_tv = grid.surface_data.grad.clone()
    assert g_tv.abs().sum().item() > 0
    grid.inplace_surface_normal_grad(grid.surface_data.grad, scaling=1e-6, sparse_frac=1, connectivity_check=True,
                                     ignore_empty=False, use_l1=True)
    assert (grid.surface_data.grad - g_tv).abs().sum().item() > 0
    before = grid.surface_data.grad.clone()
    loss = grid.inplace_surface_eikonal_grad(grid.surface_data.grad, scaling=1e-4, sparse_frac=1, use_kernel=False)
    cells = torch.where(grid.links.view(-1) >= 0)[0].int().cpu().numpy()
    go, lo, _ = oracle.surface_eikonal_grad_sparse(grid.links.cpu().numpy(), grid.surface_data.data.cpu().numpy(),
                                                   cells, 1e-4, grid._get_h())
    grad_close((grid.surface_data.grad - before).cpu().numpy(), go, 1e-4, "eikonal through the grid")
    assert abs(float(loss.item()) - lo) <= 1e-4 * abs(lo)
    grid.inplace_alpha_surf_sparsify_grad(grid.density_data.grad, grid.surface_data.grad, scaling_alpha=1e-3,
                                          scaling_surf=1e-3, sparse_frac=0.1)
    grid.inplace_tv_color_grad(grid.sh_data.grad, start_dim=1, scaling=1e-3, sparse_frac=0.01)
    assert grid.sh_data.grad[:, 1:].abs().sum().item() > 0 and grid.sh_data.grad[:, 0].abs().sum().item() == 0
    assert grid.sparse_grad_indexer.any()


def test_dense_tv_and_l2_colour(env, oracle):
    """sparse_frac >= 1 takes the dense tv_grad (loss_kernel.cu:120-190): every interior cell, no mask; the L2
    colour penalty is tensor arithmetic over the rows the last render touched (svox2.py:5897-5928)."""
    torch, _C = env
    import svox2
    from test_grid_api_gpu import make_grid
    grid = make_grid(torch, svox2, 24, 4)
    n = grid.capacity
    for p in (grid.density_data, grid.sh_data):
        p.grad = torch.zeros_like(p.data)
    grid.sparse_grad_indexer = torch.zeros(n, dtype=torch.bool, device="cuda")
    grid.inplace_tv_grad(grid.density_data.grad, scaling=1e-3, sparse_frac=1.0)
    assert grid.sparse_grad_indexer is None
    links = grid.links.cpu().numpy()
    sx, sy, sz = links.shape
    cells = np.arange(sx * sy * sz, dtype=np.int32).reshape(sx, sy, sz)[:-1, :-1, :-1].reshape(-1)
    go, _ = oracle.tv_grad_sparse(links, grid.density_data.data.cpu().numpy(), cells, 1e-3)
    grad_close(grid.density_data.grad.cpu().numpy(), go, 1e-4, "dense tv_grad")
    grid.inplace_tv_color_grad(grid.sh_data.grad, start_dim=3, end_dim=9, scaling=1e-2, sparse_frac=1.0)
    go, _ = oracle.tv_grad_sparse(links, grid.sh_data.data.cpu().numpy(), cells, 1e-2, start_dim=3, end_dim=9,
                                  ignore_edge=True)
    grad_close(grid.sh_data.grad.cpu().numpy(), go, 1e-4, "dense colour tv_grad")
    # L2 colour: all rows, then the touched rows only (mask and index-list forms)
    g = torch.zeros_like(grid.sh_data.data)
    grid.sparse_sh_grad_indexer = None
    grid.inplace_l2_color_grad(g, start_dim=1, scaling=2.0)
    assert torch.allclose(g[:, 1:], 2.0 / n * grid.sh_data.data[:, 1:]) and not g[:, 0].any()
    for frac in (0.5, 0.05):
        mask = torch.rand(n, device="cuda") < frac
        grid.sparse_sh_grad_indexer = mask
        g = torch.zeros_like(grid.sh_data.data)
        grid.inplace_l2_color_grad(g, scaling=3.0)
        nz = int(mask.sum())
        assert torch.allclose(g[mask], 3.0 / nz * grid.sh_data.data[mask]) and not g[~mask].any()


def test_tv_value_and_its_autograd(env, oracle):
    """_C.tv (loss_kernel.cu:73-117): the dense TV value against the oracle, both edge rules, on grids with link 0
    and empty cells; SparseGrid.tv_color differentiates through it with the dense tv_grad (svox2.py:507-546)."""
    torch, _C = env
    import svox2
    from test_grid_api_gpu import make_grid
    grid = make_grid(torch, svox2, 24, 4)
    holes = torch.rand(grid.links.shape, device="cuda") < 0.3
    holes[0, 0, 0] = False
    grid.links = torch.where(holes, torch.full_like(grid.links, -1), grid.links).contiguous()
    links = grid.links.cpu().numpy()
    assert (links < 0).any() and (links == 0).any()
    for data, dims, edge in ((grid.density_data.data, (0, 1), False), (grid.sh_data.data, (0, 12), True),
                             (grid.sh_data.data, (3, 9), True), (grid.sh_data.data, (5, 6), False)):
        got = float(_C.tv(grid.links, data, dims[0], dims[1], False, 2.0, edge, -1.0, -1.0).item())
        want = oracle.tv_value(links, data.cpu().numpy(), dims[0], dims[1], edge)
        assert abs(got - want) <= 2e-5 * abs(want), (dims, edge, got, want)
    with pytest.raises(RuntimeError):
        _C.tv(grid.links, grid.sh_data.data, 4, 4, False, 2.0, True, -1.0, -1.0)
    # a grid of one cell per axis has one term; a flat grid has none
    one = torch.zeros(2, 2, 2, dtype=torch.int32, device="cuda")
    one[1, 0, 0] = 1
    val = torch.tensor([[1.0], [3.0]], device="cuda")
    got = float(_C.tv(one, val, 0, 1, False, 2.0, False, -1.0, -1.0).item())
    assert abs(got - oracle.tv_value(one.cpu().numpy(), val.cpu().numpy(), 0, 1, False)) < 1e-6
    flat = torch.zeros(1, 4, 4, dtype=torch.int32, device="cuda")
    assert float(_C.tv(flat, val, 0, 1, False, 2.0, False, -1.0, -1.0).item()) == 0.0
    # autograd: value + the dense gradient, as the reference's _TotalVariationFunction
    with pytest.raises(NotImplementedError):
        grid.tv()  # surface grids: the reference raises too (svox2.py:4904-4905)
    grid.sh_data.grad = None
    tvc = grid.tv_color(start_dim=3, end_dim=-3)
    want = oracle.tv_value(links, grid.sh_data.data.cpu().numpy(), 3, 9, True)
    assert abs(float(tvc.item()) - want) <= 2e-5 * abs(want)
    tvc.backward()
    sx, sy, sz = links.shape
    cells = np.arange(sx * sy * sz, dtype=np.int32).reshape(sx, sy, sz)[:-1, :-1, :-1].reshape(-1)
    go, _ = oracle.tv_grad_sparse(links, grid.sh_data.data.cpu().numpy(), cells, 1.0, start_dim=3, end_dim=9,
                                  ignore_edge=True)
    grad_close(grid.sh_data.grad.cpu().numpy(), go, 1e-4, "tv_color backward")
