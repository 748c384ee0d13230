"""HIP `cuvol` renderer (the Plenoxels volume renderer of render_lerp_kernel_cuvol.cu) through the
C ABI, against the reference's goldens (its PyTorch renderer, tests/golden/gen_golden_cuvol.py) and
the CPU oracle.  `pytest -m gpu`."""
import numpy as np
import pytest

from alphasurf_amd import lib as _asurf_lib

from golden_util import grad_close
from test_hip_parity import hip  # noqa: F401  (fixture)
from test_oracle_cuvol import CUVOL_CASES, load_cuvol

pytestmark = pytest.mark.gpu


def cuvol_scene(basis_dim, reso=48, seed=5, surface_type=100):
    """Pruned-grid look-alike with signed densities (the gate sigma > sigma_thresh acts)."""
    from alphasurf_amd import synth
    sc = synth.make_sparse_scene(reso, basis_dim, seed=seed)
    rs = np.random.RandomState(seed + 7)
    n = sc["density"].shape[0]
    sc["density"] = (rs.randn(n, 1) * 3.0 + 1.0).astype(np.float32)
    sc["sh"][:, ::basis_dim] = (0.5 + 0.3 * rs.randn(n, 3)).astype(np.float32)
    sc["surface_type"] = surface_type
    o, d = synth.random_rays(2000, seed + 100)
    return sc, o, d


def compare(hip, oracle, sc, o, d, opts, beta_loss=0.0, sparsity_loss=0.0, rgb_tol=2e-5, grad_tol=2e-4):
    gt = np.random.RandomState(3).rand(o.shape[0], 3).astype(np.float32)
    rgb, g = hip.cuvol_fused(sc, o, d, opts, gt, beta_loss=beta_loss, sparsity_loss=sparsity_loss)
    rgb_o, go = oracle.cuvol_fused(sc, o, d, opts, gt, beta_loss=beta_loss, sparsity_loss=sparsity_loss)
    assert np.abs(rgb - rgb_o).max() <= rgb_tol, np.abs(rgb - rgb_o).max()
    for k in ("sh", "density"):
        grad_close(g[k], go[k], grad_tol, f"grad_{k}")
    assert (g["mask"].astype(bool) == go["mask"].astype(bool)).all()
    return rgb, g


@pytest.mark.parametrize("name", CUVOL_CASES)
def test_golden(hip, oracle, name):
    z, sc, opt = load_cuvol(oracle, name)
    sc["surface_type"] = 100
    rgb, g = hip.cuvol_fused(sc, z["origins"], z["dirs"], opt, z["rgb_gt"])
    assert np.abs(rgb - z["rgb"]).max() <= 1e-4
    grad_close(g["sh"], z["grad_sh"], 5e-4, f"{name} grad_sh")
    grad_close(g["density"], z["grad_density"], 5e-4, f"{name} grad_density")
    compare(hip, oracle, sc, z["origins"], z["dirs"], opt)
    fwd = hip.cuvol_forward(sc, z["origins"], z["dirs"], opt)
    np.testing.assert_array_equal(fwd, rgb)


OPTION_SETS = {
    "plenoxels": dict(sigma_thresh=1e-8, stop_thresh=1e-7, step_size=0.5),
    "gate": dict(sigma_thresh=0.7, stop_thresh=1e-3, step_size=0.31),
    "no_gate": dict(sigma_thresh=-1e9, stop_thresh=0.0, step_size=0.8),
    "opaque_end": dict(sigma_thresh=0.0, stop_thresh=1e-7, step_size=0.5, last_sample_opaque=True),
    "near_clip": dict(sigma_thresh=0.0, stop_thresh=1e-7, step_size=0.45, near_clip=0.35),
    "spheric": dict(sigma_thresh=0.0, stop_thresh=0.0, step_size=0.5, use_spheric_clip=True, near_clip=0.1),
    "dark": dict(sigma_thresh=0.0, stop_thresh=1e-7, step_size=0.5, background_brightness=0.0),
}


@pytest.mark.parametrize("basis_dim", [1, 4, 9, 16])
@pytest.mark.parametrize("optname", list(OPTION_SETS))
def test_sparse_scene_vs_oracle(hip, oracle, basis_dim, optname):
    sc, o, d = cuvol_scene(basis_dim)
    rgb, g = compare(hip, oracle, sc, o, d, OPTION_SETS[optname])
    assert np.abs(g["density"]).max() > 0 and (np.abs(rgb - OPTION_SETS[optname].get("background_brightness", 1.0))
                                                .max(axis=1) > 1e-3).sum() > 300


@pytest.mark.parametrize("beta,sparsity", [(0.1, 0.0), (0.0, 0.02), (0.05, 0.01)])
def test_fused_losses(hip, oracle, beta, sparsity):
    sc, o, d = cuvol_scene(9)
    compare(hip, oracle, sc, o, d, OPTION_SETS["plenoxels"], beta_loss=beta, sparsity_loss=sparsity)


def test_backward_equals_fused(hip, oracle):
    sc, o, d = cuvol_scene(4)
    opt = OPTION_SETS["plenoxels"]
    gt = np.random.RandomState(3).rand(o.shape[0], 3).astype(np.float32)
    rgb, gf = hip.cuvol_fused(sc, o, d, opt, gt)
    gout = (rgb - gt) * np.float32(2.0 / (3 * rgb.shape[0]))
    gb = hip.cuvol_backward(sc, o, d, opt, gout, rgb)
    for k in ("sh", "density"):
        grad_close(gb[k], gf[k], 1e-5, k)
    go = oracle.cuvol_backward(sc, o, d, opt, gout, rgb)
    for k in ("sh", "density"):
        grad_close(gb[k], go[k], 2e-4, k)


def test_surface_plane_is_ignored(hip, oracle):
    """An SDF grid rendered with cuvol (the `no_surface` warm-up of opt.py:690-830) == the same
    grid without its surface plane."""
    sc, o, d = cuvol_scene(4, surface_type=102)
    a = hip.cuvol_forward(sc, o, d, OPTION_SETS["plenoxels"])
    b = hip.cuvol_forward(dict(sc, surface_type=100), o, d, OPTION_SETS["plenoxels"])
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("shape", [(40, 24, 56), (17, 33, 9), (8, 8, 8), (3, 5, 2)])
def test_odd_shapes_and_sample_set(hip, oracle, shape):
    """Grids that are not whole bricks, down to smaller than one; dense random links (some missing
    corners inside the cells that are sampled)."""
    rs = np.random.RandomState(sum(shape))
    present = rs.rand(*shape) < 0.6
    n = int(present.sum())
    links = np.full(shape, -1, np.int32)
    links[present] = np.arange(n, dtype=np.int32)
    B = 4
    sh = (rs.randn(n, 3 * B) * 0.2).astype(np.float32)
    sc = oracle.make_scene(links, (rs.randn(n, 1) * 3 + 1).astype(np.float32), np.zeros((n, 1), np.float32), sh,
                           np.zeros((0,), np.float32), 0.5 * np.asarray(shape, np.float32),
                           0.5 * np.asarray(shape, np.float32), B, surface_type=100)
    from alphasurf_amd import synth
    o, d = synth.random_rays(1500, 9)
    compare(hip, oracle, sc, o, d, dict(sigma_thresh=0.0, stop_thresh=1e-7, step_size=0.5))


def test_empty_brick_skipping_is_exact(hip, oracle):
    """A few small blobs in a 96^3 grid: almost every brick is empty, rays start far outside, inside,
    graze faces.  The HIP march leaves empty bricks in one move and must still evaluate exactly the
    reference's sample parameters (repeated t += step): tight tolerance, every ray."""
    from alphasurf_amd import synth
    sc = synth.make_sparse_scene(96, 4, seed=11, n_spheres=3, radius_range=(0.04, 0.08), band=2.0, dilate=1)
    rs = np.random.RandomState(2)
    n = sc["density"].shape[0]
    sc["density"] = (np.abs(rs.randn(n, 1)) * 2.0).astype(np.float32)
    sc["surface_type"] = 100
    o, d = synth.random_rays(3000, 77)
    # axis-parallel and face-grazing rays
    o2 = np.tile(np.array([[-3.0, 0.0, 0.0]], np.float32), (64, 1))
    o2[:, 1:] = rs.uniform(-0.6, 0.6, size=(64, 2))
    d2 = np.tile(np.array([[1.0, 0.0, 0.0]], np.float32), (64, 1))
    o3 = rs.uniform(-0.5, 0.5, size=(200, 3)).astype(np.float32)          # origins inside the grid
    d3 = rs.randn(200, 3).astype(np.float32)
    d3 /= np.linalg.norm(d3, axis=1, keepdims=True)
    o, d = np.concatenate([o, o2, o3]), np.concatenate([d, d2, d3])
    for opts in (dict(sigma_thresh=1e-8, stop_thresh=1e-7, step_size=0.5),
                 dict(sigma_thresh=1e-8, stop_thresh=0.0, step_size=0.07),
                 dict(sigma_thresh=1e-8, stop_thresh=1e-7, step_size=2.9, last_sample_opaque=True)):
        compare(hip, oracle, sc, o, d, opts, rgb_tol=1e-5, grad_tol=1e-4)


def test_errors(hip):
    sc, o, d = cuvol_scene(4)
    with pytest.raises(RuntimeError):
        hip.cuvol_forward(sc, o, d, dict(step_size=0.0))
    rgb = hip.cuvol_forward(sc, o[:0], d[:0], OPTION_SETS["plenoxels"])
    assert rgb.shape == (0, 3)


def test_sparse_grid_api(hip):
    """SparseGrid with backend 'cuvol' (a surface-less grid) and `no_surface=True` on an SDF grid."""
    import torch
    import svox2
    from alphasurf_amd import synth
    dev = torch.device("cuda:0")
    grid = svox2.SparseGrid(reso=32, basis_dim=4, use_sphere_bound=True, device=dev,
                            surface_type=svox2.SURFACE_TYPE_NONE)
    grid.opt.backend = "cuvol"
    grid.opt.sigma_thresh, grid.opt.stop_thresh, grid.opt.step_size = 1e-8, 1e-7, 0.5
    gen = torch.Generator(device="cpu").manual_seed(0)
    grid.density_data.data[:] = (torch.randn(grid.density_data.shape, generator=gen) * 2 + 1).to(dev)
    grid.sh_data.data[:] = (torch.randn(grid.sh_data.shape, generator=gen) * 0.2).to(dev)
    o, d = synth.random_rays(1024, 3)
    rays = svox2.Rays(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev))
    gt = torch.rand(1024, 3, generator=gen).to(dev)
    rgb = grid.volume_render(rays)["rgb"]
    torch.nn.functional.mse_loss(rgb, gt).backward()
    ga_d, ga_s = grid.density_data.grad.clone(), grid.sh_data.grad.clone()
    grid.density_data.grad = grid.sh_data.grad = None
    out = grid.volume_render_fused(rays, gt)["rgb"]
    assert torch.equal(out, rgb.detach())
    assert torch.allclose(grid.density_data.grad, ga_d, rtol=1e-4, atol=1e-7 * float(ga_d.abs().max()) + 1e-12)
    assert torch.allclose(grid.sh_data.grad, ga_s, rtol=1e-4, atol=1e-5 * float(ga_s.abs().max()))
    assert grid.sparse_grad_indexer.any()

    sdf = svox2.SparseGrid(reso=32, basis_dim=4, use_sphere_bound=True, device=dev,
                           surface_type=svox2.SURFACE_TYPE_SDF, surface_init="sphere")
    sdf.opt.backend = "surf_trav"
    sdf.opt.sigma_thresh, sdf.opt.stop_thresh, sdf.opt.step_size = 1e-8, 1e-7, 0.5
    sdf.density_data.data[:] = grid.density_data.data
    sdf.sh_data.data[:] = grid.sh_data.data
    out2 = sdf.volume_render_fused(rays, gt, no_surface=True)["rgb"]
    assert torch.equal(out2, out)
    assert sdf.surface_data.grad is None or not sdf.surface_data.grad.any()
    assert torch.equal(sdf.density_data.grad, grid.density_data.grad) or \
        torch.allclose(sdf.density_data.grad, grid.density_data.grad, rtol=1e-4, atol=1e-9)


def test_image_backward_is_the_ray_backward(hip):
    """volume_render_cuvol_image_backward (svox2.cpp:98-100): the gradient of an image render equals the ray
    backward over the camera's per-pixel rays, bit for bit when the atomics land in the same order or within
    float summation noise otherwise."""
    import torch
    import svox2
    import svox2.csrc as _C
    dev = torch.device("cuda:0")
    grid = svox2.SparseGrid(reso=32, basis_dim=4, use_sphere_bound=True, device=dev,
                            surface_type=svox2.SURFACE_TYPE_NONE)
    grid.opt.backend = "cuvol"
    grid.opt.sigma_thresh, grid.opt.stop_thresh, grid.opt.step_size = 1e-8, 1e-7, 0.5
    gen = torch.Generator(device="cpu").manual_seed(1)
    grid.density_data.data[:] = (torch.randn(grid.density_data.shape, generator=gen) * 2 + 1).to(dev)
    grid.sh_data.data[:] = (torch.randn(grid.sh_data.shape, generator=gen) * 0.2).to(dev)
    c2w = torch.eye(4, device=dev)
    c2w[2, 3] = -2.5
    cam = svox2.Camera(c2w, fx=30.0, width=20, height=14)
    g, o = grid._to_cpp(), grid.opt._to_cpp()
    img = _C.volume_render_cuvol_image(g, cam._to_cpp(), o)
    assert img.shape == (14, 20, 3)
    grad_out = torch.randn(14, 20, 3, generator=gen).to(dev)

    def holder():
        h = _C.GridOutputGrads()
        h.grad_density_out = torch.zeros_like(grid.density_data.data)
        h.grad_sh_out = torch.zeros_like(grid.sh_data.data)
        return h
    a, b = holder(), holder()
    _C.volume_render_cuvol_image_backward(g, cam._to_cpp(), o, grad_out, img, a)
    from alphasurf_amd import csrc_api
    rays = csrc_api._camera_rays(cam._to_cpp())  # the pixel rays of the image entry points
    assert torch.equal(_C.volume_render_cuvol(g, rays, o).view(14, 20, 3), img)
    _C.volume_render_cuvol_backward(g, rays, o, grad_out.view(-1, 3), img.view(-1, 3), b)
    assert torch.allclose(grid.volume_render(cam.gen_rays())["rgb"].detach().view(14, 20, 3), img, atol=1e-5)
    assert a.grad_sh_out.abs().sum() > 0
    for x, y in ((a.grad_density_out, b.grad_density_out), (a.grad_sh_out, b.grad_sh_out)):
        assert torch.allclose(x, y, rtol=1e-4, atol=1e-6 * float(y.abs().max()))
    with pytest.raises(RuntimeError):
        _C.volume_render_cuvol_image_backward(g, cam._to_cpp(), o, grad_out[:4], img, a)


@pytest.mark.parametrize("optname", ["plenoxels", "gate", "opaque_end", "spheric"])
def test_depth_renders_vs_oracle(hip, oracle, optname):
    sc, o, d = cuvol_scene(4)
    opt = OPTION_SETS[optname]
    for what, param in (("expected_term", 0.0), ("expected_term", 0.6), ("mode_term", 0.1), ("sigma_thresh", 0.5),
                        ("sigma_thresh", 4.0)):
        got = hip.cuvol_eval(sc, o, d, opt, what, param)
        ref = oracle.cuvol_eval(sc, o, d, opt, what, param)
        assert (ref > 0).sum() > 200
        if what == "mode_term":      # two samples of nearly equal weight may swap
            assert (np.abs(got - ref) > 1e-5 * (1 + np.abs(ref))).mean() < 0.01
        else:
            np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-6, err_msg=f"{what} {param}")
    gd, gs = hip.cuvol_eval(sc, o, d, opt, "med_term", max_sample=48)
    rd, rs_ = oracle.cuvol_eval(sc, o, d, opt, "med_term", max_sample=48)
    np.testing.assert_allclose(gd, rd, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(gs, rs_, rtol=1e-5, atol=1e-6)


def test_depth_through_the_grid_api(hip):
    import torch
    import svox2
    from alphasurf_amd import synth
    dev = torch.device("cuda:0")
    grid = svox2.SparseGrid(reso=32, basis_dim=1, use_sphere_bound=True, device=dev,
                            surface_type=svox2.SURFACE_TYPE_SDF, surface_init="sphere")
    grid.opt.backend = "surf_trav"
    grid.opt.sigma_thresh, grid.opt.stop_thresh, grid.opt.step_size = 1e-8, 1e-7, 0.5
    ax = (torch.arange(32, dtype=torch.float32) + 0.5) / 16 - 1
    r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2).to(dev)
    grid.density_data.data[:] = torch.where(r < 0.5, 80.0, 0.0)[grid.links >= 0].view(-1, 1)
    cam = svox2.Camera(torch.from_numpy(synth.look_at_c2w([2.5, 0.0, 0.0]).astype(np.float32)).to(dev), fx=60.0,
                       width=48, height=48)
    for kw in (dict(), dict(depth_type="mode"), dict(depth_type="med"), dict(sigma_thresh=10.0)):
        depth = grid.volume_render_depth_image(cam, no_surface=True, **kw)
        centre = float(depth[24, 24])
        tol = 0.5 if kw.get("depth_type") == "med" else 0.12         # 'med' picks a sample inside the ball
        assert depth.shape == (48, 48) and -0.05 < centre - 2.0 < tol, (kw, centre)   # eye at 2.5, ball radius 0.5
        assert float(depth[0, 0]) == 0.0 or np.isnan(float(depth[0, 0]))


def test_rays_per_wave_is_transparent(hip, monkeypatch):
    """Small batches spread their rays over more waves (1 .. 64 marching lanes per wave): the picture does not
    depend on it, the gradients only through the order of the atomic sums."""
    sc, o, d = cuvol_scene(9)
    gt = np.random.RandomState(3).rand(o.shape[0], 3).astype(np.float32)
    opts = OPTION_SETS["plenoxels"]
    ref = None
    for rpw in ("64", "16", "4", "1"):
        _asurf_lib.config_set("ASURF_CUVOL_RPW", rpw)
        rgb, g = hip.cuvol_fused(sc, o, d, opts, gt, beta_loss=0.05, sparsity_loss=0.01)
        _asurf_lib.config_set("ASURF_CUVOL_RPW", None)
        if ref is None:
            ref = (rgb, g)
            continue
        assert np.array_equal(rgb, ref[0]), rpw
        assert (g["mask"] == ref[1]["mask"]).all()
        for k in ("sh", "density"):
            grad_close(g[k], ref[1][k], 1e-5, f"rpw {rpw} grad_{k}")


def test_image_batch_by_pixel_tiles(hip):
    """asurf_rays::image_width: a wave takes an 8 x 8 pixel tile of the image instead of 64 pixels of a row and serves a
    round's records in the order of their cells -- the picture bit for bit, the gradients up to the order of the atomic
    sums; an image the tiles do not partition (width 300) is taken by rows."""
    from alphasurf_amd import synth
    sc = synth.make_sparse_scene(64, 9, seed=3, trunc_a=3.0)
    sc = dict(sc, density=(sc["density"] * 30.0).astype(np.float32))
    opts = OPTION_SETS["plenoxels"]
    for width, height in ((320, 264), (300, 250)):
        o, d = synth.pinhole_rays(synth.orbit_camera(1), width, height)
        assert o.shape[0] >= 65536     # (64 rays a wave: ASURF_CUVOL_RPW's automatic choice below that is smaller)
        gt = np.random.RandomState(3).rand(o.shape[0], 3).astype(np.float32)
        _asurf_lib.config_set("ASURF_CUVOL_RPW", "64")
        ref = hip.cuvol_fused(sc, o, d, opts, gt, beta_loss=0.05, sparsity_loss=0.01)
        for tile in (None, "4", "16"):     # (8 pixels wide by default)
            _asurf_lib.config_set("ASURF_CUVOL_TILE", tile)
            got = hip.cuvol_fused(sc, o, d, opts, gt, beta_loss=0.05, sparsity_loss=0.01, image_width=width)
            assert float(np.abs(ref[0]).max()) > 0 and np.array_equal(got[0], ref[0]), (width, tile)
            assert (got[1]["mask"] == ref[1]["mask"]).all()
            for k in ("sh", "density"):
                grad_close(got[1][k], ref[1][k], 1e-5, f"width {width} tile {tile} grad_{k}")
        _asurf_lib.config_set("ASURF_CUVOL_TILE", None)
        _asurf_lib.config_set("ASURF_CUVOL_RPW", None)
