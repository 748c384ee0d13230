"""HIP weight renders (grid_weight_render, sparse_grid_weight_render) against the CPU oracle, and
SparseGrid.resample with the camera-weight criterion.  `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from alphasurf_amd import csrc_api as _C, synth
from test_hip_parity import hip  # noqa: F401  (fixture: builds / loads the library)

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def camera(eye, w, h, f):
    cam = _C.CameraSpec()
    cam.c2w = torch.from_numpy(synth.look_at_c2w(eye).astype(np.float32)).to(DEV)
    cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height = f, f * 1.05, w * 0.5, h * 0.5, w, h
    return cam


def scene(shape, seed):
    rs = np.random.RandomState(seed)
    data = np.maximum(rs.randn(*shape) * 3 + 0.5, 0).astype(np.float32)
    data[rs.rand(*shape) < 0.5] = 0
    gsz = np.asarray(shape, np.float32)
    return data, torch.from_numpy(0.5 * gsz - 0.5), torch.from_numpy(0.5 * gsz)


def close(got, ref, what):
    # a sample within float rounding of a cell face may fall into the neighbouring cell on one side only
    bad = np.abs(got - ref) > 1e-5
    assert bad.mean() < 2e-3, f"{what}: {bad.sum()} of {bad.size} differ, max {np.abs(got - ref).max():.3e}"


@pytest.mark.parametrize("shape", [(40, 36, 44), (9, 17, 5), (2, 2, 2)])
@pytest.mark.parametrize("opaque,stop", [(False, 0.2), (True, 0.0)])
def test_dense_weight_render(hip, oracle, shape, opaque, stop):
    data, off, scl = scene(shape, sum(shape))
    out = torch.zeros(shape, device=DEV)
    ref = np.zeros(shape, np.float32)
    for eye in ([1.9, -1.2, 1.0], [-1.4, 1.6, -0.7]):
        cam = camera(eye, 96, 64, 70.0)
        _C.grid_weight_render(torch.from_numpy(data).to(DEV), cam, 0.5, stop, opaque, off, scl, out)
        oracle.weight_render(data, None, cam.c2w.cpu().numpy(), cam.fx, cam.fy, cam.cx, cam.cy, cam.width, cam.height,
                             0.5, stop, opaque, off.numpy(), scl.numpy(), out=ref)
    torch.cuda.synchronize()
    assert ref.max() > 0.05
    close(out.cpu().numpy(), ref, "dense")


def test_sparse_weight_render(hip, oracle):
    sc = synth.make_sparse_scene(48, 1, seed=9)
    sc["density"] = (np.abs(np.random.RandomState(1).randn(*sc["density"].shape)) * 20).astype(np.float32)
    g = hip.grid_spec(sc)
    off, scl = torch.from_numpy(sc["offset"]), torch.from_numpy(sc["scaling"])
    out = torch.zeros(sc["links"].shape, device=DEV)
    cam = camera([2.0, 1.1, 0.8], 128, 96, 110.0)
    _C.sparse_grid_weight_render(g, cam, 0.5, 0.0, off, scl, out)
    torch.cuda.synchronize()
    ref = oracle.weight_render(sc["density"], sc["links"], cam.c2w.cpu().numpy(), cam.fx, cam.fy, cam.cx, cam.cy,
                               cam.width, cam.height, 0.5, 0.0, False, sc["offset"], sc["scaling"])
    assert ref.max() == 1.0 and (ref > 0).sum() > 1000
    close(out.cpu().numpy(), ref, "sparse")


def test_errors(hip):
    data, off, scl = scene((8, 8, 8), 0)
    out = torch.zeros((8, 8, 8), device=DEV)
    cam = camera([2, 0, 0], 8, 8, 8.0)
    with pytest.raises(RuntimeError):
        _C.grid_weight_render(torch.from_numpy(data).to(DEV), cam, 0.0, 0.2, False, off, scl, out)
    with pytest.raises(RuntimeError):
        _C.grid_weight_render(torch.from_numpy(data).to(DEV), cam, 0.5, 0.2, False, off, scl, out[:4])
    cam.ndc_coeffx = 1.0
    with pytest.raises(NotImplementedError):
        _C.grid_weight_render(torch.from_numpy(data).to(DEV), cam, 0.5, 0.2, False, off, scl, out)


def test_resample_with_camera_weights(hip):
    """A ball of matter inside a fog of small density: the density criterion keeps the fog, the camera-weight
    criterion (Plenoxels' default upsampling rule) keeps only what the views see with weight."""
    import svox2
    def make():
        grid = svox2.SparseGrid(reso=32, basis_dim=1, device=DEV, surface_type=svox2.SURFACE_TYPE_NONE)
        ax = (torch.arange(32, dtype=torch.float32) + 0.5) / 16 - 1
        r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
        dens = torch.where(r < 0.5, torch.tensor(60.0), torch.tensor(0.5))
        grid.density_data.data[:] = dens.view(-1, 1).to(DEV)
        grid.sh_data.data[:] = 0.3
        return grid

    grid, by_sigma = make(), make()
    cams = [svox2.Camera(torch.from_numpy(synth.look_at_c2w(e).astype(np.float32)).to(DEV), fx=90.0, width=96,
                         height=96) for e in ([2.5, 0, 0], [0, 2.5, 0.3], [-1.8, -1.8, 0.5], [0.2, 0.1, 2.5])]
    by_sigma.resample(48, sigma_thresh=0.1, dilate=0, accelerate=False)
    grid.resample(48, weight_thresh=0.05, dilate=1, cameras=cams, accelerate=False)
    kept_w, kept_s = int((grid.links >= 0).sum()), int((by_sigma.links >= 0).sum())
    assert kept_s == 48 ** 3 and 0 < kept_w < 0.35 * kept_s
    # what was kept is the ball's shell as seen from outside: it lies near r = 0.5, the opaque core and the far fog are gone
    ax2 = (torch.arange(48, dtype=torch.float32) + 0.5) / 24 - 1
    r2 = torch.sqrt(ax2[:, None, None] ** 2 + ax2[None, :, None] ** 2 + ax2[None, None, :] ** 2).to(DEV)
    rk = r2[grid.links >= 0]
    assert rk.max() < 0.75 and rk.min() > 0.2 and ((rk > 0.35) & (rk < 0.65)).float().mean() > 0.6
    assert grid.density_data.shape[0] == kept_w and grid.sh_data.shape == (kept_w, 3)


def test_weight_masked_surface_initialisation(hip):
    """init_type='weight' (:3118-3147): vertices no camera sees through empty space in front of them get
    surface = -level (inside the level set); seen ones keep density - level."""
    import svox2
    grid = svox2.SparseGrid(reso=32, basis_dim=1, device=DEV, surface_type=svox2.SURFACE_TYPE_NONE)
    grid.opt.alpha_activation_type = svox2.EXP_FN if hasattr(svox2, "EXP_FN") else 1
    ax = (torch.arange(32, dtype=torch.float32) + 0.5) / 16 - 1
    r = torch.sqrt(ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2)
    grid.density_data.data[:] = torch.where(r < 0.6, torch.tensor(40.0), torch.tensor(0.0)).view(-1, 1).to(DEV)
    cams = [svox2.Camera(torch.from_numpy(synth.look_at_c2w(e).astype(np.float32)).to(DEV), fx=90.0, width=96,
                         height=96) for e in ([2.5, 0, 0], [0, 2.5, 0.3], [-1.8, -1.8, 0.5], [0.2, 0.1, 2.5],
                                              [0.1, -0.2, -2.5], [-2.5, 0.3, 0.1], [0.2, -2.5, 0.2])]
    grid.init_surface_from_density(density_lvs=[10.0], init_type="weight", weight_init_cams=cams, prune_threshold=1e-8,
                                   dilate=1)
    assert grid.surface_data is not None and grid.surface_type == svox2.SURFACE_TYPE_SDF
    lk = grid.links
    rr = r.to(DEV)[lk >= 0]
    surf = grid.surface_data.data[lk[lk >= 0].long(), 0]
    core, rim = surf[rr < 0.3], surf[(rr > 0.5) & (rr < 0.58)]
    assert core.numel() > 100 and rim.numel() > 100
    assert (core < 0).all()                         # hidden behind the rim: density * 0 - level
    assert (rim > 0).float().mean() > 0.8           # the first thing every camera meets
    with pytest.raises(RuntimeError):
        svox2.SparseGrid(reso=8, basis_dim=1, device=DEV).init_surface_from_density(init_type="weight")
