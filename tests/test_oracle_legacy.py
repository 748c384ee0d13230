"""CPU checks of the oracle's restatement of the legacy `surface` backend
(render_lerp_kernel_surface.cu).  The reference cannot run that backend from Python (its
voxel lists need kaolin), so it is pinned indirectly: on an option set
where surf_trav computes the same quantity, the reference's surf_trav golden is the answer."""
import numpy as np
import pytest

from golden_util import grad_close, load_case
from test_oracle_golden import scene_of

LEGACY_CASES = ["r16_sh9_legacy", "r16_sh4_legacy"]


@pytest.mark.parametrize("name", LEGACY_CASES)
def test_legacy_forward_matches_reference_surf_trav(oracle, name):
    z, meta = load_case(name)
    sc = scene_of(oracle, z, meta)
    rv = oracle.trace_voxel_lists(sc, z["origins"], z["dirs"], meta["opt"])
    assert rv[2].sum() == rv[0].shape[0] and rv[0].shape[0] > 0
    rgb = oracle.surface_forward(sc, z["origins"], z["dirs"], rv, meta["opt"])
    err = np.abs(rgb - z["rgb"]).max()
    assert err <= 1e-4, f"{name}: RGB L-inf {err:.3e}"


@pytest.mark.parametrize("name", LEGACY_CASES)
def test_legacy_fused_gradients_match_reference_surf_trav(oracle, name):
    z, meta = load_case(name)
    sc = scene_of(oracle, z, meta)
    rv = oracle.trace_voxel_lists(sc, z["origins"], z["dirs"], meta["opt"])
    rgb, grads = oracle.surface_fused(sc, z["origins"], z["dirs"], rv, meta["opt"], z["rgb_gt"])
    assert np.abs(rgb - z["rgb"]).max() <= 1e-4
    grad_close(grads["sh"], z["grad_sh"], 5e-4, f"{name} grad_sh")
    grad_close(grads["density"], z["grad_density"], 5e-4, f"{name} grad_density")
    # float32 evaluation of d f_k / d s_corner about the far-away ray origin (the reference's
    # arithmetic, render_util.cuh:1705-1775) cancels to ~1e-4 of its terms ...
    grad_close(grads["surface"], z["grad_surface"], 3e-2, f"{name} grad_surface (f32)")
    # ... the same expression in float64 lands on the reference's surf_trav gradient
    oracle.set_legacy_f64_surface_grad(True)
    try:
        _, g64 = oracle.surface_fused(sc, z["origins"], z["dirs"], rv, meta["opt"], z["rgb_gt"])
    finally:
        oracle.set_legacy_f64_surface_grad(False)
    grad_close(g64["surface"], z["grad_surface"], 5e-4, f"{name} grad_surface (f64)")


def test_legacy_backward_equals_fused(oracle):
    z, meta = load_case("r16_sh4_legacy")
    sc = scene_of(oracle, z, meta)
    rv = oracle.trace_voxel_lists(sc, z["origins"], z["dirs"], meta["opt"])
    rgb, gf = oracle.surface_fused(sc, z["origins"], z["dirs"], rv, meta["opt"], z["rgb_gt"])
    q = rgb.shape[0]
    gout = (rgb - z["rgb_gt"]) * np.float32(2.0 / (3 * q))
    gb = oracle.surface_backward(sc, z["origins"], z["dirs"], rv, meta["opt"], gout, rgb)
    for k in ("sh", "density", "surface"):
        np.testing.assert_allclose(gb[k], gf[k], rtol=1e-5, atol=1e-9)


def test_legacy_root_gradient_is_the_implicit_derivative(oracle):
    """d t / d f_k = -t^k / P'(t) for every simple root; the f/g/h chain of
    render_util.cuh:1418-1702 must reproduce it (three-root and one-root cubics, quadratics,
    linear)."""
    rs = np.random.RandomState(5)
    n = 4000
    fs = rs.uniform(-2, 2, size=(n, 4))
    fs[: n // 8, 3] = 0.0                 # quadratics
    fs[: n // 16, 2] = 0.0                # linear
    for st_id in range(3):
        ids = np.full(n, st_id, np.int32)
        g, st = oracle.test_cubic_root_grad(fs, ids, np.ones(n, np.float32), use_vieta=False)
        gv, stv = oracle.test_cubic_root_grad(fs, ids, np.ones(n, np.float32), use_vieta=True)
        np.testing.assert_array_equal(st, stv)      # same solver feeds both (test_cuda.cu:51-56)
        t = st[:, st_id].astype(np.float64)
        dp = fs[:, 1] + 2 * fs[:, 2] * t + 3 * fs[:, 3] * t * t
        # keep well-conditioned, real roots that exist for this st_id
        resid = fs[:, 0] + fs[:, 1] * t + fs[:, 2] * t ** 2 + fs[:, 3] * t ** 3
        ok = (np.abs(g).sum(1) > 0) & (np.abs(dp) > 0.2) & (np.abs(resid) < 1e-4) & (np.abs(t) < 20)
        # quadratics with f2 < 0: both sign branches of render_util.cuh:1447-1462 are the same
        # code, so the reference differentiates the OTHER root there; the restatement keeps
        # that behaviour and the identity below does not apply
        ok &= ~((fs[:, 3] == 0) & (fs[:, 2] < 0))
        assert ok.sum() > 100
        want = np.stack([-(t ** k) / dp for k in range(4)], 1)
        live = np.abs(fs) > 0       # coefficients that are identically zero get no gradient
        err = np.abs(g[ok] - want[ok]) * live[ok]
        assert err.max() <= 2e-3 * max(1.0, np.abs(want[ok]).max()), err.max()
        np.testing.assert_allclose(g[ok], gv[ok], rtol=2e-3, atol=2e-4)


def test_legacy_finite_difference(oracle):
    """Directional finite difference of the MSE loss through the legacy oracle itself."""
    z, meta = load_case("r16_sh4_legacy")
    sc = scene_of(oracle, z, meta)
    o, d = z["origins"][:96], z["dirs"][:96]
    rv = oracle.trace_voxel_lists(sc, o, d, meta["opt"])
    gt = z["rgb_gt"][:96]
    _, grads = oracle.surface_fused(sc, o, d, rv, meta["opt"], gt)

    def loss(scn):
        rgb = oracle.surface_forward(scn, o, d, rv, meta["opt"]).astype(np.float64)
        return ((rgb - gt) ** 2).mean()

    for key, gkey, eps in (("density", "density", 1e-2), ("sh", "sh", 1e-2), ("surface", "surface", 1e-4)):
        # along sign(grad): every term of the directional derivative has the same sign, so the
        # rounding of the float32 forward pass stays small against it
        direction = np.sign(grads[gkey].reshape(sc[key].shape)).astype(np.float32)
        plus = dict(sc)
        minus = dict(sc)
        plus[key] = (sc[key] + eps * direction).astype(np.float32)
        minus[key] = (sc[key] - eps * direction).astype(np.float32)
        fd = (loss(plus) - loss(minus)) / (2 * eps)
        an = float((grads[gkey].reshape(direction.shape).astype(np.float64) * direction).sum())
        assert abs(fd - an) <= 0.05 * max(abs(an), abs(fd)) + 2e-6, (key, fd, an)
