"""CPU checks of the oracle's restatement of the Plenoxels `cuvol` renderer
(render_lerp_kernel_cuvol.cu) against goldens from the reference's own PyTorch renderer
(svox2.py:1215-1351; tests/golden/gen_golden_cuvol.py)."""
import os

import numpy as np
import pytest

from golden_util import GOLDEN_DIR, grad_close

CUVOL_CASES = ["cuvol_r16_sh9", "cuvol_r20_sh4"]
CUVOL_OPT = dict(sigma_thresh=0.0, stop_thresh=0.0, background_brightness=1.0, near_clip=0.0)


def load_cuvol(oracle, name):
    z = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))
    sc = oracle.make_scene(z["links"], z["density"], np.zeros_like(z["density"]), z["sh"], np.zeros((0,), np.float32),
                           z["offset"], z["scaling"], int(z["basis_dim"]), surface_type=0)
    opt = dict(CUVOL_OPT, step_size=float(z["step_size"]))
    return z, sc, opt


@pytest.mark.parametrize("name", CUVOL_CASES)
def test_cuvol_forward_matches_reference(oracle, name):
    z, sc, opt = load_cuvol(oracle, name)
    rgb = oracle.cuvol_forward(sc, z["origins"], z["dirs"], opt)
    err = np.abs(rgb - z["rgb"]).max()
    assert err <= 1e-4, f"{name}: RGB L-inf {err:.3e}"
    assert (np.abs(z["rgb"] - 1.0).max(axis=1) > 1e-3).sum() > 100      # the rays do cross matter


@pytest.mark.parametrize("name", CUVOL_CASES)
def test_cuvol_fused_gradients_match_reference(oracle, name):
    z, sc, opt = load_cuvol(oracle, name)
    rgb, grads = oracle.cuvol_fused(sc, z["origins"], z["dirs"], opt, z["rgb_gt"])
    assert np.abs(rgb - z["rgb"]).max() <= 1e-4
    grad_close(grads["sh"], z["grad_sh"], 5e-4, f"{name} grad_sh")
    grad_close(grads["density"], z["grad_density"], 5e-4, f"{name} grad_density")
    touched = grads["mask"].astype(bool)
    assert touched.any() and not np.any(grads["density"][~touched])


def test_cuvol_backward_equals_fused(oracle):
    z, sc, opt = load_cuvol(oracle, "cuvol_r20_sh4")
    rgb, gf = oracle.cuvol_fused(sc, z["origins"], z["dirs"], opt, z["rgb_gt"])
    gout = (rgb - z["rgb_gt"]) * np.float32(2.0 / (3 * rgb.shape[0]))
    gb = oracle.cuvol_backward(sc, z["origins"], z["dirs"], opt, gout, rgb)
    for k in ("sh", "density"):
        np.testing.assert_allclose(gb[k], gf[k], rtol=1e-5, atol=1e-9)


def test_cuvol_density_gradient_is_the_derivative(oracle):
    """Central differences of the MSE loss through the oracle's own forward (sparsity and beta terms
    added analytically: render_lerp_kernel_cuvol.cu:396-400, :488-494)."""
    z, sc, opt = load_cuvol(oracle, "cuvol_r20_sh4")
    opt = dict(opt, sigma_thresh=-1e9)          # no gate: the loss is smooth in the densities
    o, d, gt = z["origins"][:96], z["dirs"][:96], z["rgb_gt"][:96]
    _, g = oracle.cuvol_fused(sc, o, d, opt, gt)
    rs = np.random.RandomState(0)
    rows = rs.choice(np.flatnonzero(np.abs(g["density"][:, 0]) > 1e-5), 12, replace=False)

    def loss(den):
        s2 = dict(sc, density=den)
        return float(np.mean((oracle.cuvol_forward(s2, o, d, opt).astype(np.float64) - gt) ** 2))

    for r in rows:
        h = 2e-2
        dp, dm = sc["density"].copy(), sc["density"].copy()
        dp[r] += h
        dm[r] -= h
        fd = (loss(dp) - loss(dm)) / (2 * h)
        assert abs(fd - g["density"][r, 0]) <= 2e-2 * abs(fd) + 2e-6, (r, fd, g["density"][r, 0])


def test_cuvol_skips_by_the_distance_transform(oracle):
    """Links below -1 (accelerate(): render_util.cuh:286-348) shorten the march but leave the picture
    within float rounding of the un-accelerated one."""
    z, sc, opt = load_cuvol(oracle, "cuvol_r16_sh9")
    links = z["links"].copy()
    occ = links >= 0
    n = 0
    # an aligned block of s^3 cells whose (s+1)^3 vertices are all missing may carry -(log2(s) + 1)
    for s, code in ((4, -3), (2, -2)):
        for x in range(0, 16, s):
            for y in range(0, 16, s):
                for zz in range(0, 16, s):
                    blk = links[x:x + s, y:y + s, zz:zz + s]
                    if (blk == -1).all() and not occ[x:x + s + 1, y:y + s + 1, zz:zz + s + 1].any():
                        blk[...] = code
                        n += 1
    assert n > 0
    sc2 = dict(sc, links=np.ascontiguousarray(links))
    a = oracle.cuvol_forward(sc, z["origins"], z["dirs"], opt)
    b = oracle.cuvol_forward(sc2, z["origins"], z["dirs"], opt)
    assert np.abs(a - b).max() <= 2e-4


def test_cuvol_options(oracle):
    z, sc, opt = load_cuvol(oracle, "cuvol_r16_sh9")
    base = oracle.cuvol_forward(sc, z["origins"], z["dirs"], opt)
    opaque = oracle.cuvol_forward(sc, z["origins"], z["dirs"], dict(opt, last_sample_opaque=1))
    assert np.abs(base - opaque).max() > 1e-3
    stop = oracle.cuvol_forward(sc, z["origins"], z["dirs"], dict(opt, stop_thresh=0.2))
    assert 1e-4 < np.abs(base - stop).max() < 0.25
    with pytest.raises(RuntimeError):
        oracle.cuvol_forward(sc, z["origins"], z["dirs"], dict(opt, step_size=0.0))


def test_cuvol_depth_renders(oracle):
    """The depth renders share the pinned march; their accumulators are checked against each other."""
    z, sc, opt = load_cuvol(oracle, "cuvol_r16_sh9")
    o, d = z["origins"], z["dirs"]
    mean = oracle.cuvol_eval(sc, o, d, opt, "expected_term", 0.0)
    mode = oracle.cuvol_eval(sc, o, d, opt, "mode_term", 0.0)
    depths, sigmas = oracle.cuvol_eval(sc, o, d, opt, "med_term", max_sample=64)
    first = oracle.cuvol_eval(sc, o, d, opt, "sigma_thresh", 0.0)
    hit = sigmas[:, 0] > 0            # (a ray that starts inside the grid may have its first sample at t = 0)
    assert hit.sum() > 100
    np.testing.assert_array_equal(first[hit], depths[hit, 0])        # sigma_thresh == the gate: the first sample
    assert not first[~hit].any() and not mean[~hit].any() and not mode[~hit].any()
    n = (sigmas > 0).sum(1)
    last = depths[np.arange(len(n)), np.maximum(n - 1, 0)]
    full = hit & (n < 64)
    assert (mean[full] <= last[full] * (1 + 1e-5)).all()             # a weighted sum with total weight <= 1
    assert (np.diff(depths, axis=1)[:, :1][n > 1] > 0).all()         # increasing along the ray
    assert ((mode[full] >= depths[full, 0] * (1 - 1e-6)) & (mode[full] <= last[full] * (1 + 1e-6))).all()
    # weight gate: an impossible threshold zeroes everything; a higher sigma threshold is met later or never
    assert not oracle.cuvol_eval(sc, o, d, opt, "expected_term", 1.0).any()
    later = oracle.cuvol_eval(sc, o, d, opt, "sigma_thresh", 3.0)
    assert ((later == 0) | (later >= first)).all() and (later > first).any()
    # depth is the world distance: t / step * world_step with |dir| = 1
    gsz = np.asarray(z["links"].shape, np.float32)
    assert abs(float(first[hit].min())) < 4 * float(np.linalg.norm(o, axis=1).max())
