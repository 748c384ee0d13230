"""Grid maintenance kernels against numpy restatements of
misc_kernel.cu:24-58 (dilate) and :113-180 (accel_dist_prop), and SparseGrid.prune_grid.
`pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch
    assert torch.cuda.is_available()
    from alphasurf_amd import csrc_api as _C, lib
    lib.lib()
    return torch, _C


def np_dilate(g):
    """OR over the 3x3x3 neighbourhood with clamped edges (misc_kernel.cu:34-57)."""
    p = np.pad(g, 1, mode="edge")
    out = np.zeros_like(g)
    sx, sy, sz = g.shape
    for i in range(3):
        for j in range(3):
            for k in range(3):
                out |= p[i:i + sx, j:j + sy, k:k + sz]
    return out


def np_accel(links):
    """-(1 + number of pyramid levels whose 2^k-cell around the voxel holds no vertex), the pyramid
    shrinking by ceil(n/2) per level while every side is > 1 (misc_kernel.cu:113-180)."""
    out = links.copy()
    occ = links >= 0
    res = np.full(links.shape, -1, np.int32)
    done = np.zeros(links.shape, bool)
    sx, sy, sz = links.shape
    X, Y, Z = np.meshgrid(np.arange(sx), np.arange(sy), np.arange(sz), indexing="ij")
    cur, shift = occ, 0
    while min(sx, sy, sz) > 1:
        shift += 1
        sx, sy, sz = ((sx - 1) >> 1) + 1, ((sy - 1) >> 1) + 1, ((sz - 1) >> 1) + 1
        lvl = np.zeros((sx, sy, sz), bool)
        ox, oy, oz = np.nonzero(occ)
        lvl[ox >> shift, oy >> shift, oz >> shift] = True
        hit = lvl[X >> shift, Y >> shift, Z >> shift]
        done |= hit
        res[~done] -= 1
    out[links < 0] = res[links < 0]
    return out


@pytest.mark.parametrize("shape", [(16, 16, 16), (33, 20, 47), (1, 9, 5)])
def test_dilate(env, shape):
    torch, _C = env
    rs = np.random.RandomState(1)
    g = rs.rand(*shape) < 0.03
    out = _C.dilate(torch.from_numpy(g).cuda())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), np_dilate(g))


@pytest.mark.parametrize("shape", [(32, 32, 32), (40, 24, 17), (3, 64, 64)])
def test_accel_dist_prop(env, shape):
    torch, _C = env
    rs = np.random.RandomState(2)
    links = np.full(shape, -1, np.int32)
    blob = rs.rand(*shape) < 0.01
    links[blob] = np.arange(int(blob.sum()), dtype=np.int32)
    t = torch.from_numpy(links.copy()).cuda()
    _C.accel_dist_prop(t)
    torch.cuda.synchronize()
    got = t.cpu().numpy()
    np.testing.assert_array_equal(got, np_accel(links))
    assert ((got >= 0) == (links >= 0)).all() and (got[links >= 0] == links[links >= 0]).all()


def test_prune_grid_keeps_the_picture(env):
    """Pruning (threshold below every opacity, surface-crossing test, 2 dilations) drops most of a dense
    grid and leaves the render unchanged; accelerate() does not change it either."""
    torch, _C = env
    import svox2
    from test_grid_api_gpu import make_grid
    grid = make_grid(torch, svox2, 32, 4)
    grid.opt.surf_fake_sample = False
    # one sphere of radius 10 voxels (the constructor's 'sphere' initialisation is a stack of
    # concentric shells 2 voxels apart: the surface crosses zero next to every vertex)
    L = grid.links.cpu().numpy()
    X, Y, Z = np.meshgrid(*[np.arange(r) for r in L.shape], indexing="ij")
    sdf = np.sqrt((X - 16.0) ** 2 + (Y - 16.0) ** 2 + (Z - 16.0) ** 2) - 10.0
    surf = np.zeros((grid.capacity, 1), np.float32)
    surf[L.reshape(-1), 0] = sdf.reshape(-1)
    grid.surface_data.data[:] = torch.from_numpy(surf).cuda()
    rs = np.random.RandomState(3)
    o = (rs.randn(600, 3) * 3).astype(np.float32)
    d = (-o + rs.randn(600, 3).astype(np.float32) * 0.3)
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    rays = svox2.Rays(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda())
    before = grid.volume_render(rays)["rgb"].detach().clone()
    n0 = grid.capacity
    kept = grid.prune_grid(density_raw_thres=-10.0, dilate=2, prune_surf=True)
    # (vertices on the grid border always count as crossing: their outside neighbours read as 0, as in the reference)
    assert 0.02 < kept < 0.9 and grid.density_data.shape[0] < n0
    assert (grid.links[grid.links >= 0].unique().numel() == grid.density_data.shape[0])
    assert int(grid.links.min()) < -1            # accelerate() ran
    after = grid.volume_render(rays)["rgb"].detach()
    assert (before - after).abs().max().item() <= 1e-5


def test_init_surface_from_density(env):
    """A Plenoxels-like density grid (a soft ball) becomes a surface grid whose level set is the chosen
    density iso-surface: pruned to a shell, gradient norm ~1, renderable; depth of the level set
    matches the ball's radius."""
    torch, _C = env
    import svox2
    reso = 48
    grid = svox2.SparseGrid(reso=reso, center=[0.0, 0.0, 0.0], radius=[1.0, 1.0, 1.0], basis_dim=4,
                            use_z_order=False, device="cuda:0", background_nlayers=0,
                            basis_type=svox2.BASIS_TYPE_SH, surface_type=svox2.SURFACE_TYPE_NONE,
                            use_sphere_bound=False)
    L = grid.links.cpu().numpy()
    X, Y, Z = np.meshgrid(*[np.arange(r) for r in L.shape], indexing="ij")
    rad = np.sqrt((X - 23.5) ** 2 + (Y - 23.5) ** 2 + (Z - 23.5) ** 2)
    dens = np.clip(30.0 - 2.0 * rad, 0.0, None).astype(np.float32)        # density 10 at radius 10 voxels
    d = np.zeros((grid.capacity, 1), np.float32)
    d[L.reshape(-1), 0] = dens.reshape(-1)
    grid.density_data.data[:] = torch.from_numpy(d).cuda()
    grid.sh_data.data[:, 0] = 0.5
    grid.opt.backend = "surf_trav"
    grid.opt.alpha_activation_type = 1
    grid.opt.sigma_thresh = 1e-8
    grid.opt.stop_thresh = 0.0
    grid.opt.step_size = 0.5
    grid.opt.surf_fake_sample = False
    grid.opt.truncated_vol_render = False
    n0 = grid.capacity
    grid.init_surface_from_density(density_lvs=[10.0], prune_threshold=5.0, dilate=2)
    assert grid.surface_data is not None and grid.surface_data.shape == grid.density_data.shape
    assert grid.density_data.shape[0] < n0 // 2                      # pruned to the ball (+ border shells)
    assert abs(float(grid.level_set_data[0])) < 1e-6                 # the chosen level becomes 0
    # |grad surface| = 1 on average: the raw field falls by 2 per voxel = 2/h per world unit
    h = grid._get_h()
    cells = grid.surface_data.data
    assert abs(float(cells.max()) - (30.0 - 10.0) / (2.0 / h)) <= 0.15 * float(cells.max())
    # the level set is the sphere of radius 10 voxels = 10 * h in world units
    c2w = torch.eye(4)
    c2w[2, 3] = -3.0
    cam = svox2.Camera(c2w.cuda(), fx=100.0, width=32, height=32)
    depth = grid.volume_render_depth_image(cam, sigma_thresh=0.0)
    centre = depth[14:18, 14:18]
    # the grid centre (23.5 voxels) sits at world (23.5 - 24) * h; the camera looks along +z from z = -3
    want = 3.0 + (23.5 - 24.0) * h - 10.0 * h
    assert (centre > 0).all() and (centre - want).abs().max().item() <= 0.03
