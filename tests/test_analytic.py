"""Known answers for the CUDA-only gates of the live option set (tests/analytic.py): closed forms in float64 against the
oracle (CPU, here) and against the HIP path (`-m gpu`).  Neither side is consulted to produce the expectation."""
import numpy as np
import pytest

import analytic

GATES = {
    "both_crossings": dict(),
    "outward_only": dict(only_outward=True),
    "density_gate": dict(sigma_thresh=1.2),                    # sigma runs from 0.4 to 1.95 across the grid
    "density_gate_outward": dict(sigma_thresh=1.2, only_outward=True),
    "gate_above_everything": dict(sigma_thresh=5.0),           # nothing passes: background everywhere
    "stop_after_first": dict(stop_thresh=0.95),                # 1 - alpha < 0.95 for every sample: one sample, no background
    "stop_between": dict(stop_thresh=0.35),                    # some rays stop after one sample, some after two, some never
    "black_background": dict(bg=0.0),
}


def run_case(render_fused, axis, gate, sigma1=0.05):
    sc, model = analytic.slab_scene(axis=axis, sigma_axis=(axis + 1) % 3, sigma1=sigma1,
                                    sigma0=(0.4 if sigma1 else 0.9))
    o, d = analytic.slab_rays(model, n=300, seed=axis)
    gt = np.random.RandomState(9).rand(o.shape[0], 3).astype(np.float32)
    kw = GATES[gate]
    clear = analytic.expected(model, o, d, gt, **kw)["margin"] > 1e-3     # no test ray sits on a gate
    assert clear.sum() > 250
    o, d, gt = o[clear], d[clear], gt[clear]
    exp = analytic.expected(model, o, d, gt, **kw)
    opts = dict(sigma_thresh=kw.get("sigma_thresh", -1e4), stop_thresh=kw.get("stop_thresh", -1e4),
                only_outward_intersect=kw.get("only_outward", False), background_brightness=kw.get("bg", 1.0),
                step_size=0.5, near_clip=0.0, truncated_vol_render=False, surf_fake_sample=False,
                alpha_activation_type=1)
    rgb, g, n_samples = render_fused(sc, o, d, opts, gt, dict(lambda_l2=1.0, lambda_l1=0.0))
    if n_samples is not None:
        assert (n_samples == exp["n_samples"]).all()
    assert np.abs(rgb - exp["rgb"]).max() <= 2e-5, gate
    for k in ("density", "sh", "surface"):
        ref = exp["grad_" + k]
        if ref is None:
            continue
        scale = max(float(np.abs(ref).max()), 1e-12)
        err = float(np.abs(g[k].reshape(ref.shape) - ref).max())
        if np.abs(ref).max() == 0:
            assert np.abs(g[k]).max() <= 1e-12, f"{gate} grad_{k} must vanish"
        else:
            assert err <= 3e-4 * scale, f"{gate} grad_{k}: {err:.3e} vs scale {scale:.3e}"
    touched = np.abs(exp["grad_density"].reshape(-1)) > 0
    assert (g["mask"].astype(bool) | ~touched).all()
    return exp


@pytest.mark.parametrize("gate", list(GATES))
@pytest.mark.parametrize("axis", [0, 1, 2])
def test_oracle_against_closed_form(oracle, axis, gate):
    def render(sc, o, d, opts, gt, fused):
        rgb, g, st = oracle.render_fused(sc, o, d, opts, gt, fused)
        return rgb, g, st["n_samples"]
    exp = run_case(render, axis, gate)
    if gate == "both_crossings":
        assert (exp["n_samples"] == 2).sum() > 200 and (exp["n_samples"] == 0).sum() >= 10
    if gate == "outward_only":
        assert exp["n_samples"].max() == 1
    if gate == "stop_between":
        assert len(set(exp["n_samples"].tolist())) == 3


SPHERE_GATES = {
    "both_crossings": dict(),
    "outward_only": dict(only_outward=True),
    "gate_below_sigma": dict(sigma_thresh=0.5),                # sigma = 0.9 passes
    "gate_above_sigma": dict(sigma_thresh=1.0),                # ... and here it does not: background everywhere
    "stop_after_first": dict(stop_thresh=0.5),                 # 1 - alpha = 0.41 < 0.5
    "stop_after_second": dict(stop_thresh=0.2),                # 0.41^2 = 0.165 < 0.2: two samples, no background
    "outward_never_stops": dict(only_outward=True, stop_thresh=0.2),
}


def run_sphere(render_fused, gate):
    sc, model = analytic.sphere_scene()
    o, d, hits = analytic.sphere_rays(model, n=300, seed=3)
    gt = np.random.RandomState(10).rand(o.shape[0], 3).astype(np.float32)
    kw = SPHERE_GATES[gate]
    exp = analytic.sphere_expected(model, hits, gt, **kw)
    opts = dict(sigma_thresh=kw.get("sigma_thresh", -1e4), stop_thresh=kw.get("stop_thresh", -1e4),
                only_outward_intersect=kw.get("only_outward", False), background_brightness=kw.get("bg", 1.0),
                step_size=0.5, near_clip=0.0, truncated_vol_render=False, surf_fake_sample=False,
                alpha_activation_type=1)
    rgb, g, n_samples = render_fused(sc, o, d, opts, gt, dict(lambda_l2=1.0, lambda_l1=0.0))
    if n_samples is not None:
        assert (n_samples == exp["n_samples"]).all()
    assert np.abs(rgb - exp["rgb"]).max() <= 2e-5, gate
    scale = max(abs(exp["sum_grad_density"]), float(np.abs(g["density"]).sum()) * 1e-3, 1e-12)
    assert abs(float(g["density"].astype(np.float64).sum()) - exp["sum_grad_density"]) <= 3e-4 * scale
    got_sh = g["sh"].astype(np.float64).sum(axis=0)
    assert np.abs(got_sh - exp["sum_grad_sh"]).max() <= 3e-4 * max(float(np.abs(exp["sum_grad_sh"]).max()), 1e-12)
    assert float(np.abs(g["surface"]).max()) <= 1e-10      # constant sigma and colour: the crossing's place is immaterial
    return exp


@pytest.mark.parametrize("gate", list(SPHERE_GATES))
def test_oracle_sphere_known_answers(oracle, gate):
    def render(sc, o, d, opts, gt, fused):
        rgb, g, st = oracle.render_fused(sc, o, d, opts, gt, fused)
        return rgb, g, st["n_samples"]
    exp = run_sphere(render, gate)
    want = dict(both_crossings={0, 2}, outward_only={0, 1}, gate_below_sigma={0, 2}, gate_above_sigma={0},
                stop_after_first={0, 1}, stop_after_second={0, 2}, outward_never_stops={0, 1})[gate]
    assert set(exp["n_samples"].tolist()) == want


@pytest.mark.parametrize("axis", [0, 1, 2])
def test_oracle_slab_constant_sigma_has_no_surface_gradient(oracle, axis):
    def render(sc, o, d, opts, gt, fused):
        rgb, g, st = oracle.render_fused(sc, o, d, opts, gt, fused)
        return rgb, g, st["n_samples"]
    exp = run_case(render, axis, "both_crossings", sigma1=0.0)
    assert exp["grad_surface"] is not None


@pytest.mark.gpu
@pytest.mark.parametrize("gate", list(SPHERE_GATES))
def test_hip_sphere_known_answers(gate, planes):
    import hip_util

    def render(sc, o, d, opts, gt, fused):
        rgb, g = hip_util.render_fused(sc, o, d, opts, gt, fused)
        return rgb, g, None
    run_sphere(render, gate)


@pytest.mark.gpu
@pytest.mark.parametrize("axis", [0, 1, 2])
def test_hip_slab_constant_sigma_has_no_surface_gradient(axis):
    import hip_util

    def render(sc, o, d, opts, gt, fused):
        rgb, g = hip_util.render_fused(sc, o, d, opts, gt, fused)
        return rgb, g, None
    run_case(render, axis, "both_crossings", sigma1=0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("gate", list(GATES))
@pytest.mark.parametrize("axis", [0, 1, 2])
def test_hip_against_closed_form(axis, gate, planes):
    import hip_util

    def render(sc, o, d, opts, gt, fused):
        rgb, g = hip_util.render_fused(sc, o, d, opts, gt, fused)
        return rgb, g, None
    run_case(render, axis, gate)
    # (the kernel's own S counter)
    sc, model = analytic.slab_scene(axis=axis, sigma_axis=(axis + 1) % 3)
    o, d = analytic.slab_rays(model, n=300, seed=axis)
    gt = np.random.RandomState(9).rand(o.shape[0], 3).astype(np.float32)
    kw = GATES[gate]
    clear = analytic.expected(model, o, d, gt, **kw)["margin"] > 1e-3
    o, d, gt = o[clear], d[clear], gt[clear]
    exp = analytic.expected(model, o, d, gt, **kw)
    opts = dict(sigma_thresh=kw.get("sigma_thresh", -1e4), stop_thresh=kw.get("stop_thresh", -1e4),
                only_outward_intersect=kw.get("only_outward", False), background_brightness=kw.get("bg", 1.0),
                step_size=0.5, near_clip=0.0, truncated_vol_render=False, surf_fake_sample=False,
                alpha_activation_type=1)
    _, _, cnt = hip_util.render_fused(sc, o, d, opts, gt, dict(lambda_l2=1.0, lambda_l1=0.0), counters=True)
    assert int(cnt[2]) == int(exp["n_samples"].sum())
