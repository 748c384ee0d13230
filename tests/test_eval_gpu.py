"""HIP evaluation renders through the C ABI against the oracle and the
reference goldens.  `pytest -m gpu`."""
import numpy as np
import pytest

from alphasurf_amd import lib as _asurf_lib

from golden_util import load_case
from test_hip_parity import OPTION_SETS, hip, scene_of, sparse_case  # noqa: F401
from test_oracle_eval import delta_scale

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["r16_sh9_legacy", "r16_sh4_legacy"])
def test_goldens(hip, oracle, name):
    z, meta = load_case(name)
    sc = scene_of(oracle, z, meta)
    o, d, opt = z["origins"], z["dirs"], meta["opt"]
    depth = hip.render_eval(sc, o, d, opt, "expected")
    ref = z["depth"].reshape(-1) * delta_scale(z)
    assert np.abs(depth - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    depths, alphas = hip.render_eval(sc, o, d, opt, "extract_pts", param=0.1, max_sample=24)
    m = depths != 0
    pts = (o[:, None, :] + d[:, None, :] * depths[..., None])[m]
    assert pts.shape[0] == z["intersections"].shape[0]
    assert np.abs(pts - z["intersections"]).max() <= 2e-5
    assert np.abs(alphas[m] - z["intersect_alphas"]).max() <= 1e-5


@pytest.mark.parametrize("optname", ["syn_yaml", "sigmoid_stop", "spheric", "big_step"])
def test_every_mode_vs_oracle(hip, oracle, optname):
    lv = (-0.3, 0.0, 0.4) if optname == "sigmoid_stop" else (0.0,)
    sc, o, d = sparse_case(4, level_set=lv)
    opts = OPTION_SETS[optname]
    for what, param, ms in (("expected", 0.0, 1), ("mode", 0.2, 1), ("mode", 5.0, 1), ("sigma_thresh", 0.3, 1),
                            ("alpha", 0.3, 1), ("alpha", -1.0, 1), ("normal", 0.0, 1),
                            ("extract_pts", 0.05, 6), ("extract_pts", -1.0, 2)):
        got = hip.render_eval(sc, o, d, opts, what, param, ms)
        want = oracle.render_eval(sc, o, d, opts, what, param, ms)
        if what == "extract_pts":
            assert (got[0] != 0).sum() > 50
            np.testing.assert_allclose(got[0], want[0], rtol=2e-6, atol=2e-6)
            np.testing.assert_allclose(got[1], want[1], rtol=2e-6, atol=2e-6)
        else:
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5, err_msg=f"{what} {param}")


def test_edge_cases(hip, oracle):
    sc, o, d = sparse_case(1)
    # rays that miss the grid, a ray along an axis, an empty batch
    o2 = np.concatenate([o[:50], np.array([[5.0, 5.0, 5.0], [0.0, 0.0, -3.0]], np.float32)])
    d2 = np.concatenate([d[:50], np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 1.0]], np.float32)])
    for what in ("expected", "normal", "alpha"):
        np.testing.assert_allclose(hip.render_eval(sc, o2, d2, {}, what, 0.1),
                                   oracle.render_eval(sc, o2, d2, {}, what, 0.1), rtol=2e-5, atol=2e-5)
    assert hip.render_eval(sc, o[:0], d[:0], {}, "expected").shape == (0,)
    a, b = hip.render_eval(sc, o2, d2, {}, "extract_pts", 0.0, 3)
    assert a.shape == (52, 3) and (a[50] == 0).all()
    with pytest.raises(RuntimeError):
        hip.render_eval(sc, o, d, {}, "extract_pts", 0.0, 0)


def test_grid_methods(hip):
    import torch
    import svox2
    from test_grid_api_gpu import make_grid
    grid = make_grid(torch, svox2, 32, 4)
    c2w = torch.eye(4)
    c2w[2, 3] = -2.5
    cam = svox2.Camera(c2w.cuda(), fx=60.0, width=48, height=40)
    depth = grid.volume_render_depth_image(cam, batch_size=700)
    assert depth.shape == (40, 48) and (depth > 0).any()
    mode = grid.volume_render_depth_image(cam, depth_type="mode", weight_thresh=0.1)
    first = grid.volume_render_depth_image(cam, sigma_thresh=0.2)
    centre = (slice(15, 25), slice(19, 29))
    assert (first[centre] > 0).all() and (first[centre] <= mode[centre] + 1e-4).all()
    nrm = grid.volume_render_normal_image(cam)
    assert nrm.shape == (40, 48, 3) and nrm[centre].abs().sum() > 0
    amap = grid.volume_render_alpha_image(cam, thresh=0.2)
    assert amap.shape == (40, 48, 1) and ((amap == 0) | (amap > 0.2)).all()
    pts = grid.volume_render_extract_pts(cam)
    assert pts.shape[1] == 3 and pts.shape[0] == int((depth != 0).sum())
    pts2, al = grid.volume_render_extract_pts_with_alpha(cam, sigma_thresh=0.2, max_sample=8)
    assert pts2.shape[0] == al.shape[0] and (al > 0.2).all()
    # the points lie on the initial sphere-like surface, inside the unit cube
    assert pts2.abs().max() <= 1.0 + 1e-3


def test_both_forms_agree(hip, monkeypatch):
    """The render behind k_sweep + k_geom and the stepping form (ASURF_EVAL=walk) on the same rays; three level sets
    make some voxels hold more than three crossings, which hands those rays to the stepping form inside the first."""
    sc, o, d = sparse_case(4, level_set=(-0.3, 0.0, 0.4))
    for what, param, ms in (("expected", 0.0, 1), ("mode", 0.2, 1), ("sigma_thresh", 0.3, 1), ("alpha", 0.3, 1),
                            ("normal", 0.0, 1), ("extract_pts", 0.05, 6)):
        _asurf_lib.config_set("ASURF_EVAL", None)
        a = hip.render_eval(sc, o, d, OPTION_SETS["syn_yaml"], what, param, ms)
        _asurf_lib.config_set("ASURF_EVAL", "walk")
        b = hip.render_eval(sc, o, d, OPTION_SETS["syn_yaml"], what, param, ms)
        for x, y in zip(a if isinstance(a, tuple) else (a,), b if isinstance(b, tuple) else (b,)):
            np.testing.assert_allclose(x, y, rtol=2e-6, atol=2e-6, err_msg=what)
