"""Point sampling and resampling through the C ABI against the golden the
reference's PyTorch branch of SparseGrid.sample produced (tests/golden/gen_golden_sample.py; the
comparison its own test/test_sample.py makes).  `pytest -m gpu`."""
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
    return torch, _C, dict(np.load(os.path.join(GOLDEN_DIR, "sample_r16.npz")))


def spec_of(torch, _C, z, grid_coords=False):
    g = _C.SparseGridSpec()
    g.links = torch.from_numpy(z["links"]).cuda()
    g.density_data = torch.from_numpy(z["density"]).cuda()
    g.sh_data = torch.from_numpy(z["sh"]).cuda()
    g.surface_data = torch.from_numpy(z["density"] * 0.5).cuda()
    g._offset = torch.zeros(3) if grid_coords else torch.from_numpy(z["offset"])
    g._scaling = torch.ones(3) if grid_coords else torch.from_numpy(z["scaling"])
    g.basis_dim = z["sh"].shape[1] // 3
    return g


def test_sample_matches_reference(env):
    torch, _C, z = env
    g = spec_of(torch, _C, z)
    pts = torch.from_numpy(z["points"]).cuda()
    d, s = _C.sample_grid(g, pts, True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(d.cpu().numpy(), z["out_density"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(s.cpu().numpy(), z["out_sh"], rtol=1e-5, atol=2e-6)
    d2, s2 = _C.sample_grid(g, pts, False)
    assert s2.shape == (0, z["sh"].shape[1]) and torch.equal(d2, d)
    gg = spec_of(torch, _C, z, grid_coords=True)
    d3, s3 = _C.sample_grid(gg, torch.from_numpy(z["grid_points"]).cuda(), True)
    np.testing.assert_allclose(d3.cpu().numpy(), z["grid_out_density"], rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(s3.cpu().numpy(), z["grid_out_sh"], rtol=1e-5, atol=2e-6)


def test_sample_backward_matches_reference(env):
    torch, _C, z = env
    g = spec_of(torch, _C, z)
    pts = torch.from_numpy(z["points"]).cuda()
    gd, gs = torch.zeros_like(g.density_data), torch.zeros_like(g.sh_data)
    _C.sample_grid_backward(g, pts, torch.from_numpy(z["wd"]).cuda(), torch.from_numpy(z["ws"]).cuda(), gd, gs, True)
    torch.cuda.synchronize()
    grad_close(gd.cpu().numpy(), z["grad_density"], 2e-5, "grad_density")
    grad_close(gs.cpu().numpy(), z["grad_sh"], 2e-5, "grad_sh")


def test_empty_value_variants(env):
    """raw-alpha and surface sampling differ from density sampling only in what an empty corner reads as."""
    torch, _C, z = env
    g = spec_of(torch, _C, z)
    pts = torch.from_numpy(z["points"]).cuda()
    a0 = _C.sample_grid_raw_alpha(g, pts, 0.0)
    d, _ = _C.sample_grid(g, pts, False)
    assert torch.equal(a0, d)
    a1 = _C.sample_grid_raw_alpha(g, pts, -7.0)
    # linear in the empty value: a(e) = a(0) + e * (weight of the empty corners)
    a2 = _C.sample_grid_raw_alpha(g, pts, -14.0)
    np.testing.assert_allclose((a2 - a0).cpu().numpy(), 2 * (a1 - a0).cpu().numpy(), rtol=1e-4, atol=1e-5)
    assert ((a1 - a0).abs() > 0).any()
    sh, sf = _C.sample_grid_sh_surf(g, pts, False, True, -7.0)
    np.testing.assert_allclose(sf.cpu().numpy(), (0.5 * a0 + 1.0 * (a1 - a0)).cpu().numpy(), rtol=1e-4, atol=1e-5)
    assert sh.shape[0] == 0


def test_grid_sample_autograd_and_resample(env):
    torch, _C, z = env
    import svox2
    grid = svox2.SparseGrid(reso=16, center=[0.0, 0.0, 0.0], radius=[1.0, 1.0, 1.0], basis_dim=4, use_z_order=False,
                            device="cuda:0", background_nlayers=0, basis_type=svox2.BASIS_TYPE_SH,
                            surface_type=svox2.SURFACE_TYPE_NONE, use_sphere_bound=False)
    L = grid.links.cpu().numpy()
    X, Y, Z = np.meshgrid(*[np.arange(r) for r in L.shape], indexing="ij")
    rad = np.sqrt((X - 7.5) ** 2 + (Y - 7.5) ** 2 + (Z - 7.5) ** 2)
    dens = np.zeros((grid.capacity, 1), np.float32)
    dens[L.reshape(-1), 0] = np.clip(12.0 - 2.0 * rad, 0.0, None).reshape(-1)
    grid.density_data.data[:] = torch.from_numpy(dens).cuda()
    grid.sh_data.data[:] = torch.randn_like(grid.sh_data.data)
    grid.requires_grad_(True)
    pts = (torch.rand(500, 3, device="cuda") * 2 - 1) * 0.9
    d, s = grid.sample(pts)
    (d.sum() + 2 * s.sum()).backward()
    assert abs(float(grid.density_data.grad.sum()) - 500.0) <= 1e-2       # interior points: weights sum to 1
    assert abs(float(grid.sh_data.grad.sum()) - 2 * 500.0 * 12) <= 1.0
    # same-resolution resample keeps the values of the kept vertices; upsampling interpolates
    before_pts = torch.tensor([[0.0, 0.0, 0.0], [0.2, -0.1, 0.3]], device="cuda")
    ref_d, ref_s = [t.detach().clone() for t in grid.sample(before_pts)]
    grid.resample(16, sigma_thresh=0.5, dilate=1)
    assert grid.density_data.shape[0] < 16 ** 3 and int(grid.links.min()) < -1
    d1, s1 = grid.sample(before_pts)
    np.testing.assert_allclose(d1.detach().cpu().numpy(), ref_d.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(s1.detach().cpu().numpy(), ref_s.cpu().numpy(), rtol=1e-4, atol=1e-4)
    grid.resample([32, 32, 32], sigma_thresh=0.5, dilate=2)
    assert tuple(grid.links.shape) == (32, 32, 32)
    d2, _ = grid.sample(before_pts, want_colors=False)
    # (a trilinear resampling of a kinked radial field: close to the coarse value, not equal)
    np.testing.assert_allclose(d2.detach().cpu().numpy(), ref_d.cpu().numpy(), rtol=0.05, atol=1.0)


def test_resample_surface():
    """resample_surface (:4172-4420): a sphere SDF grid with opacity concentrated in a shell, 24^3 -> 40^3.  The
    new vertices carry the trilinearly sampled fields; the zero level set stays where it was."""
    import torch
    import svox2
    dev = torch.device("cuda:0")
    grid = svox2.SparseGrid(reso=24, basis_dim=4, device=dev, surface_type=svox2.SURFACE_TYPE_SDF,
                            surface_init="sphere")
    lk = grid.links
    ax = (torch.arange(24, dtype=torch.float32) + 0.5) / 12 - 1
    r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2).to(dev)
    grid.surface_data.data[:] = (r - 0.55)[lk >= 0].view(-1, 1)
    grid.density_data.data[:] = torch.where((r - 0.55).abs() < 0.25, 2.0, 0.0)[lk >= 0].view(-1, 1)
    gen = torch.Generator().manual_seed(1)
    grid.sh_data.data[:] = torch.randn(grid.sh_data.shape, generator=gen).to(dev)
    old = grid.sample_surface(torch.tensor([[0.55, 0.0, 0.0], [0.0, -0.3, 0.4]], device=dev))
    ratio = grid.resample_surface(40, alpha_thresh=0.5, dilate=1, accelerate=False)
    kept = int((grid.links >= 0).sum())
    assert tuple(grid.links.shape) == (40, 40, 40) and 0.05 < ratio < 0.7 and abs(ratio - kept / 40 ** 3) < 1e-6
    assert grid.density_data.shape == (kept, 1) and grid.surface_data.shape == (kept, 1) and \
        grid.sh_data.shape == (kept, 12)
    # rows are numbered in raster order
    flat = grid.links.view(-1)
    assert torch.equal(flat[flat >= 0], torch.arange(kept, dtype=torch.int32, device=dev))
    # kept vertices lie around the shell, and the surface field there still measures r - 0.55
    ax2 = torch.linspace(0.0, 23.5, 40)                      # new vertex positions in old grid coordinates
    w = (ax2 + 0.5) / 12 - 1
    r2 = torch.sqrt(w[:, None, None] ** 2 + w[None, :, None] ** 2 + w[None, None, :] ** 2).to(dev)
    rk = r2[grid.links >= 0]
    assert rk.min() > 0.15 and rk.max() < 0.95
    surf = grid.surface_data.data[:, 0]
    near = (rk - 0.55).abs() < 0.15
    assert near.sum() > 1000 and (surf[near] - (rk[near] - 0.55)).abs().max() < 0.02
    new = grid.sample_surface(torch.tensor([[0.55, 0.0, 0.0], [0.0, -0.3, 0.4]], device=dev))
    assert (new[1] - old[1]).abs().max() < 0.02              # same field at the same world points
    # the sigmoid activation thresholds the logit and pads missing opacity with alpha_empty_val
    g2 = svox2.SparseGrid(reso=16, basis_dim=1, device=dev, surface_type=svox2.SURFACE_TYPE_SDF, surface_init="sphere",
                          use_sphere_bound=True)
    g2.opt.alpha_activation_type = svox2.SIGMOID_FN
    g2.density_data.data[:] = 3.0
    r0 = g2.resample_surface(20, alpha_thresh=0.5, dilate=0, accelerate=False, non_expanding=True)
    assert 0.2 < r0 < 0.9 and not torch.isnan(g2.surface_data.data).any()
    with pytest.raises(RuntimeError):
        svox2.SparseGrid(reso=8, basis_dim=1, device=dev).resample_surface(8)
