"""The reference's two-stage schedule at toy scale, end to end on the HIP path ("so the
whole train_eval.sh pipeline runs"): a Plenoxels stage with the cuvol renderer (opt.py with surface_type none),
up-sampling by camera weights, surface initialisation from the density, an alpha-surface stage with surf_trav and
the live config's regularisers, evaluation renders.  Loose, behaviour-level assertions: parity lives in the
per-kernel tests.  `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from alphasurf_amd import synth

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
EYES = [[2.6, 0.0, 0.4], [0.0, 2.6, -0.3], [-2.6, 0.2, 0.2], [0.1, -2.6, 0.5], [0.3, 0.2, 2.6], [-0.2, 0.4, -2.6],
        [1.8, 1.8, 0.6], [-1.7, 1.9, -0.5]]


def psnr(mse):
    return -10.0 * np.log10(max(mse, 1e-12))


def test_two_stage_schedule():
    import svox2
    torch.manual_seed(0)
    np.random.seed(0)
    cams = [svox2.Camera(torch.from_numpy(synth.look_at_c2w(e).astype(np.float32)).to(DEV), fx=70.0, width=56,
                         height=56) for e in EYES]

    # ---- the scene to recover: an opaque ball whose colour varies with position ----
    truth = svox2.SparseGrid(reso=32, basis_dim=1, device=DEV, surface_type=svox2.SURFACE_TYPE_NONE)
    truth.opt.backend, truth.opt.step_size, truth.opt.sigma_thresh, truth.opt.stop_thresh = "cuvol", 0.5, 1e-8, 1e-7
    truth.opt.background_brightness = 1.0
    ax = (torch.arange(32, dtype=torch.float32) + 0.5) / 16 - 1
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    r = torch.sqrt(X * X + Y * Y + Z * Z)
    truth.density_data.data[:] = torch.where(r < 0.55, 80.0, 0.0).view(-1, 1).to(DEV)
    col = torch.stack([0.5 * X, 0.5 * Y, 0.5 * Z], dim=-1).view(-1, 3) / 0.28209479177387814 * 0.6
    truth.sh_data.data[:] = col.to(DEV)
    with torch.no_grad():
        images = [truth.volume_render_image(c).reshape(-1, 3) for c in cams]
    rays_o = torch.cat([c.gen_rays().origins for c in cams])
    rays_d = torch.cat([c.gen_rays().dirs for c in cams])
    target = torch.cat(images)
    assert 0.05 < float((target < 0.99).any(dim=1).float().mean()) < 0.6      # the ball covers part of every view
    n_rays = rays_o.shape[0]

    def batch(k, size=4096):
        g = torch.Generator(device="cpu").manual_seed(k)
        idx = torch.randint(0, n_rays, (size,), generator=g).to(DEV)
        return svox2.Rays(rays_o[idx].contiguous(), rays_d[idx].contiguous()), target[idx].contiguous()

    def test_mse(grid, **kw):
        with torch.no_grad():
            out = torch.cat([grid.volume_render_image(c, **kw).reshape(-1, 3) for c in cams[:3]])
        return float(torch.mean((out - target[:out.shape[0]]) ** 2))

    # ---- stage 1: Plenoxels (opt.py:808-830 with the cuvol backend, TV on density and colour) ----
    grid = svox2.SparseGrid(reso=32, basis_dim=1, device=DEV, surface_type=svox2.SURFACE_TYPE_NONE)
    grid.opt.backend, grid.opt.step_size, grid.opt.sigma_thresh, grid.opt.stop_thresh = "cuvol", 0.5, 1e-8, 1e-7
    grid.opt.background_brightness = 1.0
    grid.density_data.data[:] = 0.1
    grid.requires_grad_(True)
    mse0 = test_mse(grid)
    for it in range(150):
        rays, gt = batch(it)
        grid.volume_render_fused(rays, gt, beta_loss=0.0, sparsity_loss=1e-11)
        grid.inplace_tv_grad(grid.density_data.grad, scaling=1e-5, sparse_frac=0.05)
        grid.inplace_tv_color_grad(grid.sh_data.grad, scaling=1e-3, sparse_frac=0.05)
        grid.optim_density_step(30.0, beta=0.95, optim="rmsprop")
        grid.optim_sh_step(1e-2, beta=0.95, optim="rmsprop")
    mse1 = test_mse(grid)
    assert psnr(mse1) > psnr(mse0) + 6 and psnr(mse1) > 20, (psnr(mse0), psnr(mse1))

    # ---- up-sampling by camera weights (opt.py resample schedule) ----
    before = int((grid.links >= 0).sum())
    grid.resample(48, weight_thresh=0.02, dilate=1, cameras=cams, accelerate=True)
    kept = int((grid.links >= 0).sum())
    assert tuple(grid.links.shape) == (48, 48, 48) and 0 < kept < 0.6 * 48 ** 3, (before, kept)
    grid.requires_grad_(True)
    mse2 = test_mse(grid)
    assert psnr(mse2) > psnr(mse1) - 3, (psnr(mse1), psnr(mse2))               # the picture survives the move
    for it in range(40):
        rays, gt = batch(1000 + it)
        grid.volume_render_fused(rays, gt, sparsity_loss=1e-11)
        grid.optim_density_step(30.0, beta=0.95, optim="rmsprop")
        grid.optim_sh_step(1e-2, beta=0.95, optim="rmsprop")
    mse3 = test_mse(grid)
    assert psnr(mse3) > psnr(mse2) - 0.5
    depth = grid.volume_render_depth_image(cams[0])
    assert abs(float(depth[28, 28]) - (np.linalg.norm(EYES[0]) - 0.55)) < 0.2, float(depth[28, 28])

    # ---- the alpha-surface stage: surface from the density, surf_trav, the live regularisers ----
    grid.opt.alpha_activation_type = svox2.EXP_FN
    grid.init_surface_from_density(density_lvs=[20.0], prune_threshold=1e-8, dilate=2, alpha_rescale=0.1)
    assert grid.surface_type == svox2.SURFACE_TYPE_SDF and grid.surface_data is not None
    grid.opt.backend = "surf_trav"
    grid.opt.surf_fake_sample = False
    grid.opt.only_outward_intersect = True
    grid.opt.truncated_vol_render = True
    grid.truncated_vol_render_a = 3.0
    grid.opt.sigma_thresh, grid.opt.stop_thresh = -1e4, -1e4
    grid.requires_grad_(True)
    mse4 = test_mse(grid)
    assert np.isfinite(mse4)
    first = last = None
    for it in range(60):
        rays, gt = batch(2000 + it)
        out = grid.volume_render_fused(rays, gt, lambda_l2=1.0, lambda_l_entropy=1e-4, lambda_conv_mode_samp=1e-6)
        loss = float(torch.mean((out["rgb"] - gt) ** 2))
        assert np.isfinite(loss)
        first = loss if first is None else first
        last = loss
        grid.inplace_tv_surface_grad(grid.surface_data.grad, scaling=1e-4, sparse_frac=0.1)
        grid.inplace_surface_normal_grad(grid.surface_data.grad, scaling=1e-5, sparse_frac=0.1)
        grid.inplace_surface_eikonal_grad(grid.surface_data.grad, scaling=1e-5, sparse_frac=0.1, use_kernel=False)
        grid.inplace_tv_color_grad(grid.sh_data.grad, scaling=1e-3, sparse_frac=0.05)
        grid.optim_density_step(1e-1, beta=0.95, optim="rmsprop")
        grid.optim_surface_step(1e-4, beta=0.95, optim="rmsprop")
        grid.optim_sh_step(1e-2, beta=0.95, optim="rmsprop")
    mse5 = test_mse(grid)
    assert psnr(mse5) > psnr(mse4) - 0.5 and psnr(mse5) > 15, (psnr(mse4), psnr(mse5), first, last)

    # ---- evaluation renders of the surface stage (eval_step logging of opt.py:430-600) ----
    with torch.no_grad():
        img = grid.volume_render_image(cams[1])
        dep = grid.volume_render_depth_image(cams[1])
        nrm = grid.volume_render_normal_image(cams[1])
    assert img.shape == (56, 56, 3) and dep.shape == (56, 56) and nrm.shape[:2] == (56, 56)
    assert torch.isfinite(img).all() and torch.isfinite(dep).all()
    centre = float(dep[28, 28])
    assert abs(centre - (np.linalg.norm(EYES[1]) - 0.55)) < 0.25, centre
    # a field learnt in 60 iterations is rough: the normals of the central patch, averaged, still line up with the
    # viewing axis (the field derives from the density, which grows inwards)
    patch = nrm[22:35, 22:35].reshape(-1, nrm.shape[-1])[:, :3]
    assert torch.isfinite(patch).all() and float(patch.norm(dim=1).max()) > 0
    n_c = patch.mean(dim=0)
    to_eye = torch.tensor(EYES[1], device=DEV) / np.linalg.norm(EYES[1])
    assert float(torch.dot(n_c / (n_c.norm() + 1e-9), to_eye).abs()) > 0.5, n_c
