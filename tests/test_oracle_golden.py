"""Pins the CPU oracle against the reference: golden vectors produced by importing the
reference's pure-PyTorch surface renderer (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from golden_util import ALL_CASES, SURFACE_GRAD_RTOL, fused_args, grad_close, load_case

RGB_TOL = 1e-4       # north_star parity bound (RGB L-inf)
GRAD_RTOL = 2e-4     # relative to max |g_ref|


def scene_of(oracle, z, meta):
    return oracle.make_scene(z["links"], z["density"], z["surface"], z["sh"], z["level_set"],
                             z["offset"], z["scaling"], meta["basis_dim"],
                             meta["fake_sample_std"], meta["trunc_a"])


@pytest.mark.parametrize("name", ALL_CASES)
def test_forward_matches_reference(oracle, name):
    z, meta = load_case(name)
    sc = scene_of(oracle, z, meta)
    rgb, st = oracle.render_forward(sc, z["origins"], z["dirs"], meta["opt"])
    err = np.abs(rgb - z["rgb"]).max()
    assert err <= RGB_TOL, f"{name}: RGB L-inf {err:.3e}"
    assert st["n_samples"].sum() > 0


@pytest.mark.parametrize("name", ALL_CASES)
def test_fused_gradients_match_reference(oracle, name):
    z, meta = load_case(name)
    sc = scene_of(oracle, z, meta)
    gt, fused = fused_args(z, meta)
    rgb, grads, _ = oracle.render_fused(sc, z["origins"], z["dirs"], meta["opt"], gt, fused)
    assert np.abs(rgb - z["rgb"]).max() <= RGB_TOL
    grad_close(grads["sh"], z["grad_sh"], GRAD_RTOL, f"{name} grad_sh")
    grad_close(grads["density"], z["grad_density"], GRAD_RTOL, f"{name} grad_density")
    if meta["conv_mode"] == 0.0:
        grad_close(grads["surface"], z["grad_surface"], SURFACE_GRAD_RTOL.get(name, GRAD_RTOL), f"{name} grad_surface")
    else:
        # The reference's CUDA kernel drops d/dt of the mode sample in l_conv_mode_samp
        # (surf_trav.cu:2394-2400 uses sign(t_i - t_mode) only) while its PyTorch path
        # differentiates through t_mode as well (svox2.py:2495-2502), so the two disagree
        # by design -- the reference's own test only prints the error
        # (test/test_render_gradcheck_surface.py:137-138, 231-233).
        ref = z["grad_surface"]
        assert np.isfinite(grads["surface"]).all()
        assert np.abs(grads["surface"]).max() <= 4 * np.abs(ref).max() + 1e-6
    # fake_sample_std is a single scalar summed over all rays: compare on the scale of
    # the larger of |ref| and the density-gradient scale
    scale = max(abs(float(z["grad_fake_sample_std"].reshape(-1)[0])), float(np.abs(z["grad_density"]).max()))
    assert abs(float(grads["fake_sample_std"][0]) - float(z["grad_fake_sample_std"].reshape(-1)[0])) <= 1e-2 * scale
    # mask_out marks exactly the rows that received a density gradient
    touched = np.abs(z["grad_density"].reshape(-1)) > 0
    assert (grads["mask"].astype(bool) | ~touched).all()


def test_backward_equals_fused_with_explicit_grad(oracle):
    """volume_render_surf_trav + _backward (autograd path, svox2.py:349-430) must agree
    with the fused call when given the same dL/drgb."""
    z, meta = load_case("r16_sh9_l2")
    sc = scene_of(oracle, z, meta)
    gt, fused = fused_args(z, meta)
    rgb, g_fused, _ = oracle.render_fused(sc, z["origins"], z["dirs"], meta["opt"], gt, fused)
    q = rgb.shape[0]
    grad_out = (rgb - gt) * (2.0 / (3 * q))
    g_bwd = oracle.render_backward(sc, z["origins"], z["dirs"], meta["opt"], grad_out.astype(np.float32), rgb)
    for k in ("sh", "density", "surface"):
        grad_close(g_bwd[k], g_fused[k], 1e-5, k)


def test_cubic_solver_roots_and_gradients(oracle):
    """Known-answer properties of cubic_equation_solver_vieta / calc_cubic_root_grad_vieta
    (render_util.cuh:1126-1415): returned roots are roots; the gradient of a simple root
    is the implicit-function derivative -t^k / F'(t)."""
    rs = np.random.RandomState(0)
    n = 4000
    # three-root cubics built from their roots, single-root cubics, quadratics, linears
    r = np.sort(rs.uniform(-2, 2, size=(n, 3)), axis=-1)
    a = rs.uniform(0.2, 2, size=n) * rs.choice([-1, 1], size=n)
    fs3 = np.stack([-a * r[:, 0] * r[:, 1] * r[:, 2],
                    a * (r[:, 0] * r[:, 1] + r[:, 0] * r[:, 2] + r[:, 1] * r[:, 2]),
                    -a * r.sum(-1), a], -1)
    gap = np.diff(r, axis=-1).min(-1)
    fs3 = fs3[gap > 0.05]
    r3 = r[gap > 0.05]
    for st_id in range(3):
        g, st = oracle.test_cubic_root_grad(fs3, np.full(len(fs3), st_id), np.ones(len(fs3)))
        np.testing.assert_allclose(np.sort(st, -1), r3, atol=2e-5)
        t = st[:, st_id].astype(np.float64)
        dF = fs3[:, 1] + 2 * fs3[:, 2] * t + 3 * fs3[:, 3] * t * t
        want = np.stack([-(t ** k) / dF for k in range(4)], -1)
        np.testing.assert_allclose(g, want, rtol=2e-3, atol=1e-4)
    # one real root: x^3 + p x + q with p > 0
    p, q = rs.uniform(0.1, 3, n), rs.uniform(-3, 3, n)
    fs1 = np.stack([q, p, np.zeros(n), np.ones(n)], -1)
    g, st = oracle.test_cubic_root_grad(fs1, np.zeros(n), np.ones(n))
    t = st[:, 0].astype(np.float64)
    np.testing.assert_allclose(t ** 3 + p * t + q, 0, atol=1e-5)
    dF = p + 3 * t * t
    np.testing.assert_allclose(g, np.stack([-(t ** k) / dF for k in range(4)], -1), rtol=2e-3, atol=1e-4)
    # quadratic (f3 = 0): ordered roots, zero gradient on f3
    fs2 = np.stack([rs.uniform(-1, -0.1, n), rs.uniform(-1, 1, n), rs.uniform(0.2, 2, n) , np.zeros(n)], -1)
    for st_id in range(2):
        g, st = oracle.test_cubic_root_grad(fs2, np.full(n, st_id), np.ones(n))
        assert (st[:, 0] < st[:, 1]).all()
        t = st[:, st_id].astype(np.float64)
        np.testing.assert_allclose(fs2[:, 0] + fs2[:, 1] * t + fs2[:, 2] * t * t, 0, atol=1e-5)
        dF = fs2[:, 1] + 2 * fs2[:, 2] * t
        np.testing.assert_allclose(g[:, :3], np.stack([-(t ** k) / dF for k in range(3)], -1), rtol=2e-3, atol=1e-4)
        assert (g[:, 3] == 0).all()
    # linear; and st_id beyond the root count gives zero gradient (test_cuda.cu:84-95)
    fsl = np.stack([rs.uniform(-1, 1, n), rs.uniform(0.5, 2, n), np.zeros(n), np.zeros(n)], -1)
    g, st = oracle.test_cubic_root_grad(fsl, np.zeros(n), np.ones(n))
    np.testing.assert_allclose(st[:, 0], -fsl[:, 0] / fsl[:, 1], rtol=1e-6)
    g, _ = oracle.test_cubic_root_grad(fsl, np.ones(n), np.ones(n))
    assert (g == 0).all()
    # no root
    g, st = oracle.test_cubic_root_grad(np.array([[1.0, 0, 0, 0]]), np.zeros(1), np.ones(1))
    assert (g == 0).all() and (st == -1).all()
