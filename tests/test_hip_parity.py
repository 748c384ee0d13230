"""Parity of the HIP path (through the C ABI) against the CPU oracle and the reference's
golden vectors.  Needs a real MI355X: run with `pytest -m gpu`."""
import os

import numpy as np
import pytest

from alphasurf_amd import lib as _asurf_lib

from golden_util import ALL_CASES, SURFACE_GRAD_RTOL, fused_args, grad_close, load_case

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-4         # north_star: RGB L-inf <= 1e-4 vs the reference semantics
RGB_TOL_ORACLE = 2e-5  # HIP vs oracle: same algorithm, only libm/ocml + summation order differ
GRAD_RTOL = 2e-4       # relative to max |g|, float atomics reorder sums


@pytest.fixture(scope="module")
def hip():
    import torch
    assert torch.cuda.is_available(), "these tests need the GPU box"
    import hip_util
    from alphasurf_amd import lib
    lib.lib()  # fail loudly if libasurf_hip.so is missing
    return hip_util


def scene_of(oracle, z, meta):
    return oracle.make_scene(z["links"], z["density"], z["surface"], z["sh"], z["level_set"],
                             z["offset"], z["scaling"], meta["basis_dim"],
                             meta["fake_sample_std"], meta["trunc_a"])


@pytest.mark.parametrize("name", ALL_CASES)
def test_golden_forward(hip, oracle, planes, name):
    z, meta = load_case(name)
    sc = scene_of(oracle, z, meta)
    rgb = hip.render_forward(sc, z["origins"], z["dirs"], meta["opt"])
    assert np.abs(rgb - z["rgb"]).max() <= RGB_TOL
    rgb_o, _ = oracle.render_forward(sc, z["origins"], z["dirs"], meta["opt"])
    assert np.abs(rgb - rgb_o).max() <= RGB_TOL_ORACLE


@pytest.mark.parametrize("name", ALL_CASES)
def test_golden_fused(hip, oracle, planes, name):
    z, meta = load_case(name)
    sc = scene_of(oracle, z, meta)
    gt, fused = fused_args(z, meta)
    rgb, g = hip.render_fused(sc, z["origins"], z["dirs"], meta["opt"], gt, fused)
    assert np.abs(rgb - z["rgb"]).max() <= RGB_TOL
    # against the reference's torch gradients
    grad_close(g["sh"], z["grad_sh"], GRAD_RTOL, f"{name} grad_sh vs reference")
    grad_close(g["density"], z["grad_density"], GRAD_RTOL, f"{name} grad_density vs reference")
    if meta["conv_mode"] == 0.0:
        grad_close(g["surface"], z["grad_surface"], SURFACE_GRAD_RTOL.get(name, GRAD_RTOL),
                   f"{name} grad_surface vs reference")
    # against the oracle (covers the conv_mode term, which the reference's torch path
    # differentiates differently from its CUDA path)
    _, go, _ = oracle.render_fused(sc, z["origins"], z["dirs"], meta["opt"], gt, fused)
    for k in ("sh", "density", "surface"):
        grad_close(g[k], go[k], GRAD_RTOL, f"{name} grad_{k} vs oracle")
    assert abs(float(g["fake_sample_std"][0]) - float(go["fake_sample_std"][0])) <= \
        1e-3 * max(abs(float(go["fake_sample_std"][0])), float(np.abs(go["density"]).max()))
    assert (g["mask"].astype(bool) == go["mask"].astype(bool)).all()


@pytest.mark.parametrize("name", ["cfg1_r64_sh1", "cfg1_r64_sh1_offaxis"])
def test_config1_whole_view_vs_oracle(hip, oracle, name):
    """BASELINE.json configs[0] -- 64^3 dense SDF grid, 1 SH band, 32x32 rays -- with ALL 1024 rays of the view (the
    reference-pinned golden keeps only the rays on which the reference's PyTorch and CUDA paths are defined alike:
    28 from the on-axis pose, 862 from the off-axis one; the rest, rays through sliver voxels included, are checked
    here against the oracle), under the golden's option set and under the shipped config's."""
    z, meta = load_case(name)
    sc = scene_of(oracle, z, meta)
    o, d = z["rays_all_origins"], z["rays_all_dirs"]
    assert o.shape == (1024, 3) and int(z["rays_all_keep"].sum()) == z["origins"].shape[0]
    gt = np.random.RandomState(21).rand(1024, 3).astype(np.float32)
    for opts, fused in ((meta["opt"], dict(lambda_l1=1.0, lambda_l2=0.0, lambda_conv_mode_samp=1e-3)),
                        (OPTION_SETS["syn_yaml"], dict(lambda_l2=1.0, lambda_l_entropy=1e-4,
                                                        lambda_conv_mode_samp=1e-6))):
        rgb, g = hip.render_fused(sc, o, d, opts, gt, fused)
        rgb_o, go, st = oracle.render_fused(sc, o, d, opts, gt, fused)
        assert st["n_samples"].sum() > 1024
        assert np.abs(rgb - rgb_o).max() <= RGB_TOL_ORACLE
        for k in ("sh", "density", "surface"):
            grad_close(g[k], go[k], GRAD_RTOL, f"{name} grad_{k} vs oracle")
        assert (g["mask"].astype(bool) == go["mask"].astype(bool)).all()
        # the rays the golden keeps render to the golden's colours inside the whole view too
        if opts is meta["opt"]:
            assert np.abs(rgb[z["rays_all_keep"]] - z["rgb"]).max() <= RGB_TOL


def sparse_case(basis_dim, reso=48, seed=5, **kw):
    from alphasurf_amd import synth
    sc = synth.make_sparse_scene(reso, basis_dim, seed=seed, **kw)
    o, d = synth.random_rays(2000, seed + 100)
    return sc, o, d


OPTION_SETS = {
    "syn_yaml": dict(sigma_thresh=-1e4, stop_thresh=-1e4, only_outward_intersect=True,
                     truncated_vol_render=True, trunc_vol_weight_min=1e-10, limited_fake_sample=True,
                     fake_sample_l_dist=False),
    "sigmoid_stop": dict(alpha_activation_type=0, sigma_thresh=-3.0, stop_thresh=1e-3),
    "density_gate": dict(sigma_thresh=0.5, stop_thresh=1e-7),
    "fake_limited": dict(sigma_thresh=-20, stop_thresh=0.0, surf_fake_sample=True,
                         surf_fake_sample_min_vox_len=0.05, limited_fake_sample=True,
                         truncated_vol_render=True),
    "fake_all": dict(sigma_thresh=-20, stop_thresh=0.0, surf_fake_sample=True,
                     surf_fake_sample_min_vox_len=0.05, limited_fake_sample=False,
                     fake_sample_normalize_surf=False),
    "no_sh_dpos": dict(sigma_thresh=-20, stop_thresh=0.0, no_surf_grad_from_sh=True),
    "spheric": dict(sigma_thresh=-20, stop_thresh=0.0, use_spheric_clip=True, near_clip=0.1),
    "big_step": dict(sigma_thresh=-20, stop_thresh=0.0, step_size=3.0, near_clip=0.2),
}


@pytest.mark.parametrize("basis_dim", [1, 4, 9, 16])
@pytest.mark.parametrize("optname", list(OPTION_SETS))
def test_sparse_scene_vs_oracle(hip, oracle, basis_dim, optname):
    """Pruned-grid look-alike (links = -1 outside a shell), every option family."""
    lv = (0.0,) if optname != "fake_limited" else (-0.3, 0.0, 0.4)
    sc, o, d = sparse_case(basis_dim, level_set=lv, trunc_a=3.0, fake_sample_std=0.8)
    opts = OPTION_SETS[optname]
    rs = np.random.RandomState(3)
    gt = rs.rand(o.shape[0], 3).astype(np.float32)
    fused = dict(lambda_l2=0.7, lambda_l1=0.3)
    rgb, g, cnt = hip.render_fused(sc, o, d, opts, gt, fused, counters=True)
    rgb_o, go, st = oracle.render_fused(sc, o, d, opts, gt, fused)
    assert np.abs(rgb - rgb_o).max() <= RGB_TOL_ORACLE
    for k in ("sh", "density", "surface"):
        grad_close(g[k], go[k], GRAD_RTOL, f"grad_{k}")
    assert (g["mask"].astype(bool) == go["mask"].astype(bool)).all()
    # the kernel's V_t / V_a / S counters equal the oracle's
    assert int(cnt[0]) == int(st["n_traversed"].sum())
    assert int(cnt[1]) == int(st["n_active"].sum())
    assert int(cnt[2]) == int(st["n_samples"].sum())
    assert int(cnt[3]) == o.shape[0]
    # forward-only entry point agrees with the fused one
    rgb_f = hip.render_forward(sc, o, d, opts)
    assert np.abs(rgb_f - rgb).max() <= 1e-6


AUX_SETS = {
    "entropy_norm": dict(lambda_l_entropy=0.05, no_norm_weight_l_entropy=False),
    "entropy_raw": dict(lambda_l_entropy=0.05, no_norm_weight_l_entropy=True),
    "l_dist": dict(lambda_l_dist=0.05),
    "l_dist_a+entropy_a": dict(lambda_l_dist_a=0.05, lambda_l_entropy_a=0.02),
    "samp_dist+conv_mode": dict(lambda_l_samp_dist=0.05, lambda_conv_mode_samp=1e-3),
    "sparsity+inwards": dict(sparsity_loss=0.01, lambda_inwards_norm_loss=0.01),
    "l_di+inplace_sparse": dict(lambda_l_di=0.01, l_di_alpha_thresh=0.5, surf_sparse_alpha_thresh=0.5,
                                lambda_inplace_surf_sparse=1e-3),
    "small_cache": dict(lambda_l_dist=0.05, lambda_l_entropy=0.05, l_dist_max_sample=3),
}


@pytest.mark.parametrize("auxname", list(AUX_SETS))
def test_fused_aux_losses_vs_oracle(hip, oracle, planes, auxname):
    sc, o, d = sparse_case(9, level_set=(-0.2, 0.0, 0.3), trunc_a=4.0, fake_sample_std=0.7)
    opts = dict(sigma_thresh=-20, stop_thresh=0.0, surf_fake_sample=True, limited_fake_sample=True,
                surf_fake_sample_min_vox_len=0.05, truncated_vol_render=True)
    gt = np.random.RandomState(4).rand(o.shape[0], 3).astype(np.float32)
    fused = dict(lambda_l2=1.0)
    fused.update(AUX_SETS[auxname])
    rgb, g = hip.render_fused(sc, o, d, opts, gt, fused)
    rgb_o, go, _ = oracle.render_fused(sc, o, d, opts, gt, fused)
    assert np.abs(rgb - rgb_o).max() <= RGB_TOL_ORACLE
    for k in ("sh", "density", "surface"):
        grad_close(g[k], go[k], 5e-4, f"{auxname} grad_{k}")


def test_backward_entry_point(hip, oracle):
    sc, o, d = sparse_case(9)
    opts = OPTION_SETS["syn_yaml"]
    rgb = hip.render_forward(sc, o, d, opts)
    grad_out = np.random.RandomState(1).randn(*rgb.shape).astype(np.float32) * 1e-3
    g = hip.render_backward(sc, o, d, opts, grad_out, rgb)
    go = oracle.render_backward(sc, o, d, opts, grad_out, rgb)
    for k in ("sh", "density", "surface"):
        grad_close(g[k], go[k], GRAD_RTOL, f"grad_{k}")


def test_edge_cases(hip, oracle, planes):
    from alphasurf_amd import synth
    sc = synth.make_sparse_scene(32, 4, seed=9)
    opts = OPTION_SETS["syn_yaml"]
    # empty batch
    rgb = hip.render_forward(sc, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), opts)
    assert rgb.shape == (0, 3)
    # rays that miss the grid, axis-aligned rays (zero direction components), a ray count
    # that is not a multiple of the wave or block size
    o = np.array([[5, 5, 5], [0, 0, -3], [0.01, 0.02, -3], [-3, 0, 0], [0, 3, 0]], np.float32)
    d = np.array([[1, 0, 0], [0, 0, 1], [0, 0, 1], [1, 0, 0], [0, -1, 0]], np.float32)
    o2, d2 = synth.random_rays(322, 77)
    o, d = np.concatenate([o, o2]), np.concatenate([d, d2])
    rgb = hip.render_forward(sc, o, d, opts)
    rgb_o, _ = oracle.render_forward(sc, o, d, opts)
    assert np.abs(rgb - rgb_o).max() <= RGB_TOL_ORACLE
    assert np.allclose(rgb[0], 1.0)  # miss -> background_brightness
    # fully empty grid
    sc2 = dict(sc)
    sc2["links"] = np.full_like(sc["links"], -1)
    rgb = hip.render_forward(sc2, o, d, opts)
    assert np.allclose(rgb, 1.0)
    # non-cubic grid
    sc3 = synth.make_sparse_scene(40, 1, seed=2)
    sc3["links"] = np.ascontiguousarray(sc3["links"][:, :33, :27])
    sc3["offset"] = 0.5 * np.asarray(sc3["links"].shape, np.float32)
    sc3["scaling"] = 0.5 * np.asarray(sc3["links"].shape, np.float32)
    rgb = hip.render_forward(sc3, o, d, OPTION_SETS["fake_all"])
    rgb_o, _ = oracle.render_forward(sc3, o, d, OPTION_SETS["fake_all"])
    assert np.abs(rgb - rgb_o).max() <= RGB_TOL_ORACLE


def test_determinism_and_accumulation(hip):
    """Grad buffers are accumulated into (never zeroed); two runs agree to atomic-reorder noise."""
    import torch
    from alphasurf_amd import csrc_api as _C
    sc, o, d = sparse_case(9)
    opts = OPTION_SETS["syn_yaml"]
    gt = np.random.RandomState(4).rand(o.shape[0], 3).astype(np.float32)
    _, g1 = hip.render_fused(sc, o, d, opts, gt, dict(lambda_l2=1.0))
    _, g2 = hip.render_fused(sc, o, d, opts, gt, dict(lambda_l2=1.0))
    for k in ("sh", "density", "surface"):
        assert np.abs(g1[k] - g2[k]).max() <= 1e-6 * max(np.abs(g1[k]).max(), 1e-12) + 1e-12
    g, r, op = hip.grid_spec(sc), hip.rays_spec(o, d), hip.opt_spec(opts)
    gr = hip.zero_grads(g)
    gr.grad_sh_out += 1.0
    gtt = torch.from_numpy(gt).cuda()
    args = [dflt for _, dflt in hip.FUSED_ORDER]
    _C.volume_render_surf_trav_fused(g, r, op, gtt, *args, torch.zeros_like(gtt), gr)
    torch.cuda.synchronize()
    assert np.abs((gr.grad_sh_out.cpu().numpy() - 1.0) - g1["sh"]).max() <= 1e-6


def test_ray_sharding_equals_full_batch(hip):
    """Multi-GPU contract: shards rendered with the GLOBAL ray count in the
    loss normaliser sum to the single-batch gradient; RGB is just the concatenation."""
    sc, o, d = sparse_case(9, level_set=(-0.2, 0.0, 0.3), trunc_a=4.0)
    opts = dict(OPTION_SETS["syn_yaml"])
    gt = np.random.RandomState(4).rand(o.shape[0], 3).astype(np.float32)
    fused = dict(lambda_l2=0.8, lambda_l1=0.2, lambda_l_entropy=0.05, lambda_conv_mode_samp=1e-3)
    rgb, g = hip.render_fused(sc, o, d, opts, gt, fused)
    q = o.shape[0]
    parts = [(0, 700), (700, 1301), (1301, q)]
    acc = None
    rgbs = []
    for lo, hi in parts:
        r_i, g_i = hip.render_fused(sc, o[lo:hi], d[lo:hi], opts, gt[lo:hi], fused, global_n_rays=q)
        rgbs.append(r_i)
        if acc is None:
            acc = {k: v.astype(np.float64) for k, v in g_i.items() if k != "mask"}
            acc["mask"] = g_i["mask"].astype(bool)
        else:
            for k in ("sh", "density", "surface"):
                acc[k] += g_i[k]
            acc["mask"] |= g_i["mask"].astype(bool)
    assert np.abs(np.concatenate(rgbs) - rgb).max() <= 1e-7
    for k in ("sh", "density", "surface"):
        grad_close(acc[k].astype(np.float32), g[k], 1e-5, f"sharded grad_{k}")
    assert (acc["mask"] == g["mask"].astype(bool)).all()


def test_replay_equals_walking_backward(hip, monkeypatch):
    """The record-replay backward (default for fused calls) and the walking backward must agree,
    including rays whose record overflows (unrestricted fake samples: ~30 samples per ray)."""
    gt = np.random.RandomState(4).rand(2000, 3).astype(np.float32)
    for optname in ("syn_yaml", "fake_all", "sigmoid_stop"):
        sc, o, d = sparse_case(9, level_set=(-0.2, 0.0, 0.3), trunc_a=4.0)
        fused = dict(lambda_l2=1.0, lambda_l_entropy=0.05, lambda_l_dist=0.01)
        _asurf_lib.config_set("ASURF_NO_REPLAY", None)
        rgb1, g1 = hip.render_fused(sc, o, d, OPTION_SETS[optname], gt, fused)
        _asurf_lib.config_set("ASURF_NO_REPLAY", "1")
        rgb2, g2 = hip.render_fused(sc, o, d, OPTION_SETS[optname], gt, fused)
        _asurf_lib.config_set("ASURF_NO_REPLAY", None)
        # replay in one kernel instead of three, and the one-kernel forward instead of the pipeline
        _asurf_lib.config_set("ASURF_NO_SPLIT_BWD", "1")
        _asurf_lib.config_set("ASURF_NO_PIPELINE", "1")
        rgb3, g3 = hip.render_fused(sc, o, d, OPTION_SETS[optname], gt, fused)
        _asurf_lib.config_set("ASURF_NO_SPLIT_BWD", None)
        _asurf_lib.config_set("ASURF_NO_PIPELINE", None)
        assert np.abs(rgb1 - rgb2).max() <= 1e-6 and np.abs(rgb1 - rgb3).max() <= 1e-6
        for k in ("sh", "density", "surface"):
            grad_close(g1[k], g2[k], 1e-5, f"{optname} grad_{k} (replay vs walk)")
            grad_close(g1[k], g3[k], 1e-5, f"{optname} grad_{k} (split vs one-kernel)")
        assert (g1["mask"] == g2["mask"]).all() and (g1["mask"] == g3["mask"]).all()


def test_error_behaviour(hip):
    """Misuse raises RuntimeError (TORCH_CHECK class), unsupported features NotImplementedError."""
    import torch
    from alphasurf_amd import csrc_api as _C
    sc, o, d = sparse_case(4, reso=16)
    g, r, op = hip.grid_spec(sc), hip.rays_spec(o, d), hip.opt_spec({})
    g_bad = hip.grid_spec(sc)
    g_bad.density_data = g_bad.density_data.cpu()
    with pytest.raises(RuntimeError):
        _C.volume_render_surf_trav(g_bad, r, op)
    g_bad = hip.grid_spec(sc)
    g_bad._offset = g_bad._offset.cuda()
    with pytest.raises(RuntimeError):
        _C.volume_render_surf_trav(g_bad, r, op)
    r_bad = hip.rays_spec(o, d)
    r_bad.dirs = r_bad.dirs[:, :2]
    with pytest.raises(RuntimeError):
        _C.volume_render_surf_trav(g, r_bad, op)
    gt = torch.zeros_like(r.origins)
    args = [dflt for _, dflt in hip.FUSED_ORDER]
    args[2] = 1.0  # fused_surf_norm_reg_scale > 0: the reference asserts
    with pytest.raises(RuntimeError):
        _C.volume_render_surf_trav_fused(g, r, op, gt, *args, torch.zeros_like(gt), hip.zero_grads(g))
    with pytest.raises(NotImplementedError, match="N4"):
        _C.sparse_grid_mask_render()


def test_cubic_root_grad_kernel(hip, oracle):
    import torch
    from alphasurf_amd import csrc_api as _C
    rs = np.random.RandomState(0)
    n = 5000
    fs = rs.uniform(-2, 2, size=(n, 4))
    fs[: n // 5, 3] = 0          # quadratics
    fs[: n // 10, 2] = 0         # linears
    fs[: n // 50, 1] = 0         # no root
    st_ids = rs.randint(0, 3, size=n).astype(np.int32)
    g_in = rs.randn(n).astype(np.float32)
    go, sto = oracle.test_cubic_root_grad(fs, st_ids, g_in)
    T_fs = torch.from_numpy(fs).cuda()
    T_ids = torch.from_numpy(st_ids).cuda()
    T_g = torch.from_numpy(np.repeat(g_in[:, None], 4, 1).copy()).cuda()
    T_st = torch.zeros(n, 3, device="cuda")
    out = _C.test_cubic_root_grad(T_fs, T_ids, T_g, T_st, True)
    torch.cuda.synchronize()
    assert out is T_g
    np.testing.assert_allclose(T_st.cpu().numpy(), sto, rtol=1e-5, atol=1e-6)
    g = T_g.cpu().numpy()
    scale = np.maximum(np.abs(go), 1.0)
    assert (np.abs(g - go) / scale).max() < 1e-3


def test_sweep_dense_scene_and_long_rays(hip, monkeypatch):
    """k_sweep where its LDS budgets are exceeded: a scene in which nearly every brick holds a work voxel (the
    task list overflows: the block falls back to four passes) and rays that run inside the shell for most of
    their length (more entries than a ray's key buffer holds: handed to the one-kernel path).  Reference: the
    per-voxel walk."""
    from alphasurf_amd import synth
    rs = np.random.RandomState(3)
    reso = 64
    sc = synth.make_sparse_scene(reso, 4, seed=9, band=40.0, dilate=1)   # everything active
    n = sc["density"].shape[0]
    # a surface field that changes sign between most neighbouring vertices: a work voxel almost everywhere
    sc["surface"] = (rs.rand(n, 1).astype(np.float32) - 0.5)
    sc["density"] = (0.5 + 0.1 * rs.randn(n, 1)).astype(np.float32)
    o1, d1 = synth.random_rays(3000, 21)
    # rays along grid diagonals and almost along an axis: long runs through occupied voxels
    o2 = np.tile(np.array([[-1.4, -1.4, -1.4]], np.float32), (200, 1)) + 0.01 * rs.randn(200, 3).astype(np.float32)
    d2 = np.ones((200, 3), np.float32) + 0.02 * rs.randn(200, 3).astype(np.float32)
    o3 = np.stack([np.full(200, -1.5), rs.uniform(-0.6, 0.6, 200), rs.uniform(-0.6, 0.6, 200)], 1).astype(np.float32)
    d3 = np.stack([np.ones(200), 0.01 * rs.randn(200), 0.01 * rs.randn(200)], 1).astype(np.float32)
    o = np.concatenate([o1, o2, o3])
    d = np.concatenate([d1, d2, d3])
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    gt = rs.rand(o.shape[0], 3).astype(np.float32)
    opts = dict(OPTION_SETS["syn_yaml"], step_size=0.5)
    fused = dict(lambda_l2=1.0, lambda_l_entropy=1e-3)
    _asurf_lib.config_set("ASURF_DEBUG", "8")
    _asurf_lib.config_set("ASURF_WALK_SEG", "1")
    rgb_ref, g_ref = hip.render_fused(sc, o, d, opts, gt, fused)
    _asurf_lib.config_set("ASURF_DEBUG", None)
    _asurf_lib.config_set("ASURF_WALK_SEG", None)
    rgb, g = hip.render_fused(sc, o, d, opts, gt, fused)
    # (the rays handed to the one-kernel path sum their colours in another order than k_comp: an ulp)
    assert np.abs(rgb - rgb_ref).max() <= 5e-7, np.abs(rgb - rgb_ref).max()
    assert (g["mask"] == g_ref["mask"]).all()
    for k in ("sh", "density", "surface"):
        grad_close(g[k], g_ref[k], 1e-5, f"dense scene grad_{k}")


def test_sweep_small_and_odd_grids(hip, monkeypatch):
    """k_sweep and k_build_masks on grids whose edges are not multiples of the brick edge, down to 2 vertices per
    axis (one voxel), against the per-voxel walk."""
    from alphasurf_amd import synth
    rs = np.random.RandomState(17)
    base = synth.make_sparse_scene(24, 4, seed=2, band=30.0, dilate=0)   # every vertex active
    n = base["density"].shape[0]
    base["surface"] = (rs.rand(n, 1).astype(np.float32) - 0.5)           # a level set in most voxels
    o, d = synth.random_rays(3000, 8)
    o2 = (rs.rand(1500, 3).astype(np.float32) - 0.5) * 1.6               # origins inside the box too
    d2 = rs.randn(1500, 3).astype(np.float32)
    o = np.concatenate([o, o2])
    d = np.concatenate([d, d2 / np.linalg.norm(d2, axis=1, keepdims=True)]).astype(np.float32)
    gt = rs.rand(o.shape[0], 3).astype(np.float32)
    opts = dict(OPTION_SETS["syn_yaml"], step_size=0.5)
    fused = dict(lambda_l2=1.0)
    for shape in ((2, 2, 2), (2, 9, 3), (8, 8, 8), (9, 9, 9), (17, 5, 24), (16, 23, 10), (24, 24, 24)):
        links = base["links"][: shape[0], : shape[1], : shape[2]]
        # re-number the rows of the crop (raster order), keep the values of the vertices that remain
        keep = links.reshape(-1)
        sc = dict(base, links=np.arange(keep.size, dtype=np.int32).reshape(shape),
                  density=base["density"][keep], surface=base["surface"][keep], sh=base["sh"][keep],
                  offset=(0.5 * np.array(shape)).astype(np.float32), scaling=(0.5 * np.array(shape)).astype(np.float32))
        _asurf_lib.config_set("ASURF_DEBUG", "8")
        _asurf_lib.config_set("ASURF_WALK_SEG", "1")
        rgb_ref, g_ref = hip.render_fused(sc, o, d, opts, gt, fused)
        _asurf_lib.config_set("ASURF_DEBUG", None)
        _asurf_lib.config_set("ASURF_WALK_SEG", None)
        rgb, g = hip.render_fused(sc, o, d, opts, gt, fused)
        assert np.abs(rgb - rgb_ref).max() <= 5e-7, (shape, np.abs(rgb - rgb_ref).max())
        assert (g["mask"] == g_ref["mask"]).all(), shape
        for k in ("sh", "density", "surface"):
            grad_close(g[k], g_ref[k], 1e-5, f"shape {shape} grad_{k}")
        assert np.abs(rgb - 1.0).max() > 1e-3 or shape == (2, 2, 2)  # (something was hit)


def test_brick_skipping_is_exact(hip, monkeypatch):
    """k_sweep finds a ray's work voxels in closed form and k_walk jumps over empty 8^3 bricks; their
    work-voxel lists must be those of the per-voxel walk.  RGB is a deterministic function of that list,
    so it must match bit for bit -- on random rays and on rays built to graze voxel planes, brick corners
    and lattice points."""
    from alphasurf_amd import synth
    rs = np.random.RandomState(11)
    for reso, seed in ((48, 5), (96, 6), (130, 7)):
        sc = synth.make_sparse_scene(reso, 4, seed=seed, band=2.0, dilate=1)
        o1, d1 = synth.random_rays(6000, seed + 1)
        # world coordinates of lattice points / brick corners (grid = world * reso/2 + reso/2)
        def world(p):
            return (np.asarray(p, np.float64) - reso / 2) / (reso / 2)
        n = 3000
        tgt = rs.randint(0, reso, size=(n, 3)).astype(np.float64)
        tgt[: n // 3] = (tgt[: n // 3] // 8) * 8                     # brick corners
        src = rs.randint(-reso, 2 * reso, size=(n, 3)).astype(np.float64)
        src[n // 2:] += rs.rand(n - n // 2, 3)                       # half of them from integer origins
        keep = np.linalg.norm(tgt - src, axis=1) > 1
        o2 = world(src[keep]).astype(np.float32)
        d2 = (tgt[keep] - src[keep])
        d2 = (d2 / np.linalg.norm(d2, axis=1, keepdims=True)).astype(np.float32)
        # axis-parallel and plane-parallel rays
        o3 = world(rs.randint(0, reso, size=(600, 3)) + np.array([-2.0 * reso, 0.25, 0.5])).astype(np.float32)
        d3 = np.tile(np.array([[1.0, 0.0, 0.0]], np.float32), (600, 1))
        d3[300:] = np.array([0.6, 0.8, 0.0], np.float32)
        # rays that enter the [-0.5, R-0.5] box outside the voxel range on another axis (walk_init
        # clamps the index there: the state names a voxel the ray has not reached yet)
        m = 2000
        ent = np.stack([np.full(m, -0.5), rs.uniform(-0.5, 0.0, m), rs.uniform(0, reso, m)], 1)
        ent[m // 2:, 1] = rs.uniform(reso - 1.0, reso - 0.5, m - m // 2)
        dd = np.stack([np.ones(m), rs.uniform(0.02, 0.6, m), rs.uniform(-0.5, 0.5, m)], 1)
        dd[m // 2:, 1] *= -1.0
        perm = rs.permutation(3)                     # any axis as the entry axis
        ent, dd = ent[:, perm], dd[:, perm]
        dd /= np.linalg.norm(dd, axis=1, keepdims=True)
        o4 = world(ent - 5.0 * dd).astype(np.float32)
        d4 = dd.astype(np.float32)
        o = np.concatenate([o1, o2, o3, o4]).astype(np.float32)
        d = np.concatenate([d1, d2, d3, d4]).astype(np.float32)
        gt = rs.rand(o.shape[0], 3).astype(np.float32)
        opts = dict(OPTION_SETS["syn_yaml"], step_size=0.5)
        fused = dict(lambda_l2=1.0, lambda_l_entropy=1e-3)
        # the cube, and a non-cubic crop of it whose sizes are not multiples of the brick edge
        crop = dict(sc, links=np.ascontiguousarray(sc["links"][: reso - 5, : reso - 19, : reso // 2 + 3]))
        for scene, extra in ((sc, {}), (crop, dict(near_clip=0.3)), (sc, dict(use_spheric_clip=True))):
            op = dict(opts, **extra)
            _asurf_lib.config_set("ASURF_DEBUG", "8")        # per-voxel walk only,
            _asurf_lib.config_set("ASURF_WALK_SEG", "1")     # one lane per ray
            rgb_ref, g_ref = hip.render_fused(scene, o, d, op, gt, fused)
            _asurf_lib.config_set("ASURF_DEBUG", None)
            _asurf_lib.config_set("ASURF_WALK_SEG", None)
            # ... against the closed-form sweep (the default), and the hierarchical walk with the ray on 1, 2, 4 and
            # 8 lanes (k_walk_seg: every lane but the first enters its segment at a landing point in the middle of
            # the ray)
            for seg in ("sweep", "1", "2", "4", "8"):
                if seg != "sweep":
                    _asurf_lib.config_set("ASURF_WALK", "step")
                    _asurf_lib.config_set("ASURF_WALK_SEG", seg)
                rgb, g = hip.render_fused(scene, o, d, op, gt, fused)
                _asurf_lib.config_set("ASURF_WALK", None)
                _asurf_lib.config_set("ASURF_WALK_SEG", None)
                assert np.array_equal(rgb, rgb_ref), f"reso {reso} {extra} seg {seg}: {np.abs(rgb - rgb_ref).max()}"
                assert (g["mask"] == g_ref["mask"]).all()
                for k in ("sh", "density", "surface"):
                    grad_close(g[k], g_ref[k], 1e-5, f"reso {reso} {extra} seg {seg} grad_{k}")


def oracle_sensitivity(oracle, sc, o, d, opts, fused, gt, go):
    """How far each gradient element of the ORACLE moves under perturbations far below anything the data means: surface
    values and raw opacities by 1e-6 relative (a root at a voxel face, a normal's sign, an alpha threshold), one level at
    a time by 1e-5 (where the samples of a ray lie relative to one another: sign(mean t - t_i) of the sample-distance loss,
    surf_trav.cu:2385-2390, the mode of the conv-mode loss), the targets by 2e-7 (the sign of the L1 residual,
    surf_trav.cu:3310-3315: a colour that equals its target to the last bit), and the surface values vertex by vertex
    by 1e-6 (four draws of noise: a ray that grazes the surface)."""
    moved = {k: np.zeros_like(go[k]) for k in ("sh", "density", "surface")}

    def take(sc2, gt2):
        _, go2, _ = oracle.render_fused(sc2, o, d, opts, gt2, fused)
        for k in moved:
            moved[k] = np.maximum(moved[k], np.abs(go2[k] - go[k]))
    for plane in ("surface", "density"):
        for eps in (1e-6, -1e-6):
            sc2 = dict(sc)
            sc2[plane] = (sc[plane].astype(np.float64) * (1.0 + eps)).astype(np.float32)
            take(sc2, gt)
    for li in range(len(sc["level_set"])):
        for eps in (1e-5, -1e-5):
            lv = np.array(sc["level_set"], np.float32).copy()
            lv[li] += eps
            take(dict(sc, level_set=lv), gt)
    for eps in (2e-7, -2e-7):
        take(sc, (gt.astype(np.float64) + eps).astype(np.float32))
    # vertex by vertex (a ray that grazes the surface: two roots a hair apart, the normal across the ray -- the pair is
    # there or not, and passes the outward test or not, by the last bits of single corner values)
    rs = np.random.RandomState(12345)
    for _ in range(4):
        noise = 1.0 + 1e-6 * rs.randn(*sc["surface"].shape)
        take(dict(sc, surface=(sc["surface"].astype(np.float64) * noise).astype(np.float32)), gt)
    return moved


def explained_by_sensitivity(g, go, moved, max_rows=64, max_frac=0.10):
    """Wherever the HIP path differs from the oracle by more than the usual bound, the oracle itself moves by as much
    under the perturbations of oracle_sensitivity(), it does so in a handful of rows (a sample's voxel and the
    neighbours its loss terms read; one ray's samples), and the difference stays below `max_frac` of the plane's largest
    element: the rule must not widen itself -- an unstable oracle excuses a sample on a knife edge, not a large error
    (VERDICT r3, weak 1a).  Returns (ok, what)."""
    rows = set()
    for k in ("sh", "density", "surface"):
        scale = float(np.abs(go[k]).max())
        err = np.abs(g[k].reshape(go[k].shape) - go[k])
        over = err > 3e-4 * scale
        rows |= set(np.nonzero(over.any(axis=1))[0].tolist())
        if not (err[over] <= 1.1 * moved[k][over] + 3e-4 * scale).all():
            return False, f"grad_{k}: differs from the oracle by {float(err[over].max()):.3e} where the oracle is stable"
        if over.any() and float(err[over].max()) > max_frac * scale:
            return False, (f"grad_{k}: differs from the oracle by {float(err[over].max()) / scale:.3f} of the plane's maximum "
                           f"(an explained miss may be {max_frac} of it at most)")
    if len(rows) > max_rows:
        return False, f"{len(rows)} rows differ"
    return True, f"{len(rows)} rows on a knife edge"


def test_random_option_fuzz(hip, oracle, planes):
    """Randomised sweep over scenes, options and loss weights: every draw must agree with the
    oracle (RGB, gradients, masks).  Seeds are fixed; each draw is printed on failure.
    The fused losses hold sign and threshold tests (the L1 residual, sign(mean t - t_i), the inwards-normal term, alpha
    gates): about one draw in a thousand puts a sample within an ulp of such an edge, where the host's libm and the
    device's decide differently and a few gradient rows jump.  A draw that misses the bound must therefore be EXPLAINED:
    the oracle itself moves by as much under perturbations of 1e-6 (explained_by_sensitivity); test_fuzz_knife_edge_draw
    keeps one draw of each kind found so far and also asks that every internal form of the HIP path agrees there."""
    import fuzz_util
    seed = int(os.environ.get("ASURF_FUZZ_SEED", "2024"))
    n_draws = int(os.environ.get("ASURF_FUZZ_DRAWS", "24"))     # a longer sweep: set the two variables
    edges = []
    for it, sc, o, d, opts, fused, gt, tag in fuzz_util.draws(seed, n_draws):
        rgb, g = hip.render_fused(sc, o, d, opts, gt, fused)
        rgb_o, go, st = oracle.render_fused(sc, o, d, opts, gt, fused)
        assert np.abs(rgb - rgb_o).max() <= RGB_TOL_ORACLE, tag
        try:
            for k in ("sh", "density", "surface"):
                if np.abs(go[k]).max() > 0:
                    grad_close(g[k], go[k], 3e-4, tag + f" grad_{k}")
        except AssertionError as miss:
            ok, what = explained_by_sensitivity(g, go, oracle_sensitivity(oracle, sc, o, d, opts, fused, gt, go))
            assert ok, f"{miss}\n... and not a knife edge: {what}"
            edges.append((it, what))
        scale = max(abs(float(go["fake_sample_std"][0])), float(np.abs(go["density"]).max()), 1e-12)
        if not edges or edges[-1][0] != it:
            assert abs(float(g["fake_sample_std"][0]) - float(go["fake_sample_std"][0])) <= 1e-2 * scale, tag
        assert (g["mask"].astype(bool) == go["mask"].astype(bool)).all(), tag
    if edges:
        print(f"seed {seed}: {len(edges)} of {n_draws} draws on a knife edge: {edges}")
    assert len(edges) <= max(1, n_draws // 100), edges


@pytest.mark.parametrize("optname", ["syn_yaml", "density_gate", "fake_limited", "fake_all", "sigmoid_stop"])
def test_vertex_plane_occupancy_pass_equals_per_voxel_pass(hip, optname):
    """The occupancy pyramid built in bit arithmetic on vertex bit planes (asurf_masks2.h) against the
    per-voxel kernel (ASURF_MASKS=row): the same planes, so the same picture, gradients and masks bit for bit -- one and
    three level sets, a non-cubic crop whose sizes are no multiples of 8 or 64, the `live` plane (fake_all), the
    density gate; and the full pass in three steps (ASURF_MASKS=split: the planes cleared, the rows that hold a vertex
    listed, the listed rows built -- what grids of 512^3 and more take by default; it needs a z size that is a multiple
    of 4, the crop (45, 61, 68) has one, (45, 61, 70) makes it fall back)."""
    from alphasurf_amd import synth
    for lv, crop in (((0.0,), None), ((-0.3, 0.0, 0.4), None), ((0.0,), (45, 61, 70)), ((0.0, 0.3), (45, 61, 68))):
        sc = synth.make_sparse_scene(72, 4, seed=9, level_set=lv, trunc_a=3.0, fake_sample_std=0.8)
        if crop is not None:
            links = np.ascontiguousarray(sc["links"][:crop[0], :crop[1], :crop[2]])
            sc = dict(sc, links=links, offset=np.asarray(crop, np.float32) * 0.5, scaling=np.asarray(crop, np.float32) * 0.5)
        o, d = synth.random_rays(1500, 77)
        gt = np.random.RandomState(3).rand(o.shape[0], 3).astype(np.float32)
        fused = dict(lambda_l2=1.0, lambda_l_entropy=1e-4)
        res = []
        for form in (None, "row", "split"):
            _asurf_lib.config_set("ASURF_MASKS", form)
            res.append(hip.render_fused(sc, o, d, OPTION_SETS[optname], gt, fused))
        _asurf_lib.config_set("ASURF_MASKS", None)
        rgb_a, g_a = res[0]
        for form, (rgb_b, g_b) in zip(("row", "split"), res[1:]):
            if optname == "fake_all":
                # every voxel of the shell is a work voxel here: rays overflow their tables, and WHICH rays the spill list
                # turns away (to the one-kernel path, which sums a ray's colours in another order) depends on timing
                assert np.abs(rgb_a - rgb_b).max() <= 2.5e-7, (optname, lv, crop, form)
            else:
                assert np.array_equal(rgb_a, rgb_b), (optname, lv, crop, form)
            assert (g_a["mask"] == g_b["mask"]).all(), form
            for k in ("sh", "density", "surface"):
                grad_close(g_a[k], g_b[k], 1e-5, f"{optname} {lv} {crop} {form} grad_{k}")


INTERNAL_FORMS = (("default", {}), ("no filter", {"ASURF_DEBUG": "32"}), ("one-kernel replay", {"ASURF_NO_SPLIT_BWD": "1"}),
                  ("walking backward", {"ASURF_NO_REPLAY": "1"}), ("one-kernel forward", {"ASURF_NO_PIPELINE": "1"}),
                  ("k_walk", {"ASURF_WALK": "step"}), ("fallbacks on a side stream", {"ASURF_SIDE_STREAM": "1"}),
                  ("block-synchronous k_sweep", {"ASURF_SWEEP": "block"}), ("per-voxel occupancy pass", {"ASURF_MASKS": "row"}),
                  ("occupancy pass in three steps", {"ASURF_MASKS": "split"}))


@pytest.mark.parametrize("seed,draw", [(31337, 107), (424242, 109)])
def test_fuzz_knife_edge_draw(hip, oracle, seed, draw):
    """Draw 107 of sweep seed 31337 (round 2's long sweep): one sample sits on a knife edge -- the sign test of the
    inwards-normal loss term, decided one way by the host's libm and the other way by the device's -- and ONE element
    of grad_surface differs from the oracle by 4 % of the plane's maximum.  (Draw 109 of seed 424242, found by round 3's
    sweeps, is the second of its kind: three samples in one voxel whose weighted mean distance falls within an ulp
    of the middle sample's -- the levels' spacing happens to be in the ratio that does it, 8.2 -- so that the sign in the
    sample-distance loss, surf_trav.cu:2385-2390, is decided by the last bit of expf: five rays, 0.5 % of the plane's
    maximum, the same in every internal form.)  Kept as cases: (1) every internal form
    of the HIP path (pipeline or one kernel, k_sweep or the stepping walk, filter on or off, replay or walking
    backward) gives the same result; (2) wherever the HIP path differs from the oracle by more than the usual bound,
    the ORACLE itself moves by as much when its surface values or its raw opacities are perturbed by 1e-6 relative,
    one of its levels by 1e-5, or its targets by 2e-7 (oracle_sensitivity)."""
    import fuzz_util
    (it, sc, o, d, opts, fused, gt, tag), = list(fuzz_util.draws(seed, draw + 1, only=draw))
    res = {}
    for name, switches in INTERNAL_FORMS:
        for k, v in switches.items():
            _asurf_lib.config_set(k, v)
        res[name] = hip.render_fused(sc, o, d, opts, gt, fused)
        for k in switches:
            _asurf_lib.config_set(k, None)
    rgb, g = res["default"]
    for name, (rgb_n, g_n) in res.items():
        assert np.abs(rgb_n - rgb).max() <= 2e-6, name
        for k in ("sh", "density", "surface"):
            grad_close(g_n[k], g[k], 2e-5, f"{name} grad_{k} vs the default form")
        assert (g_n["mask"] == g["mask"]).all(), name
    rgb_o, go, _ = oracle.render_fused(sc, o, d, opts, gt, fused)
    assert np.abs(rgb - rgb_o).max() <= RGB_TOL_ORACLE
    assert (g["mask"].astype(bool) == go["mask"].astype(bool)).all()
    ok, what = explained_by_sensitivity(g, go, oracle_sensitivity(oracle, sc, o, d, opts, fused, gt, go))
    assert ok, what


def test_walk_quotient_is_the_ieee_quotient(hip):
    """walk_hier forms (plane - origin) / dir from the ray's correctly rounded reciprocal (mul, fma, fma:
    Markstein's correction step).  On the device, bit for bit against IEEE division: the walk's operand ranges,
    adversarial mantissas (all ones, just above / below a power of two), exact quotients and zero numerators."""
    import ctypes as C
    import torch
    from alphasurf_amd import lib as L
    dev = torch.device("cuda:0")
    gen = torch.Generator(device="cpu").manual_seed(7)
    total = bad = 0

    def check(x, d):
        nonlocal total, bad
        x, d = x.to(dev, torch.float32).contiguous(), d.to(dev, torch.float32).contiguous()
        fast, ieee = torch.empty_like(x), torch.empty_like(x)
        L.check(L.lib().asurf_test_quotient(C.c_void_p(x.data_ptr()), C.c_void_p(d.data_ptr()), x.numel(),
                                            C.c_void_p(fast.data_ptr()), C.c_void_p(ieee.data_ptr()), None))
        torch.cuda.synchronize()
        total += x.numel()
        usable = (1.0 / d).abs() < 1e30          # Walk::fast_div: zero / subnormal components keep the division
        ne = (fast.view(torch.int32) != ieee.view(torch.int32)) & usable
        if bool(ne.any()):
            i = torch.nonzero(ne).view(-1)[:6]
            print("MISMATCH", [(float(x[k]).hex(), float(d[k]).hex(), float(fast[k]).hex(), float(ieee[k]).hex()) for k in i])
        bad += int(ne.sum())

    for _ in range(12):
        n = 8_000_000
        x = (torch.rand(n, generator=gen) * 2 - 1) * 10 ** (torch.rand(n, generator=gen) * 6.3 - 3)
        d = torch.sign(torch.rand(n, generator=gen) - 0.5) * 10 ** (-5 * torch.rand(n, generator=gen))
        check(x, d)
    # plane - origin with integer planes, as the walk forms it
    n = 8_000_000
    planes = torch.randint(-300, 1300, (n,), generator=gen).float()
    check(planes - torch.rand(n, generator=gen) * 1024, torch.rand(n, generator=gen) * 2 - 1)
    m = torch.cat([1 + torch.arange(0, 3000) * 2.0 ** -23, 2 - torch.arange(1, 3000) * 2.0 ** -23,
                   1 + torch.rand(3000, generator=gen)])
    A, B = torch.meshgrid(m, m, indexing="ij")
    check(A.reshape(-1) * 37.0, B.reshape(-1) * 0.01)
    check(A.reshape(-1) * -0.003, B.reshape(-1) * 0.7)
    q = torch.randint(-2000, 2000, (1_000_000,), generator=gen).float()
    dd = torch.rand(1_000_000, generator=gen) + 0.01
    check(q * dd, dd)                       # near-exact quotients
    check(torch.zeros(1000), dd[:1000])     # zero numerators
    assert total > 1.8e8 and bad == 0, (total, bad)


def test_segmented_walk_with_long_voxel_lists(hip, oracle, monkeypatch):
    """Every live voxel is a work voxel here (unrestricted fake samples) and the shell is thick: rays carry
    40-100 work voxels, more than the 32-slot table (spill list) and more than the 8 a lane of k_walk_seg keeps
    in LDS (second pass).  1, 4 and 8 lanes per ray must agree with each other and with the oracle."""
    from alphasurf_amd import synth
    sc = synth.make_sparse_scene(64, 4, seed=21, band=6.0, dilate=2, n_spheres=5, fake_sample_std=0.7)
    o, d = synth.random_rays(3000, 77)
    opts = dict(OPTION_SETS["fake_all"], step_size=0.5)
    gt = np.random.RandomState(1).rand(o.shape[0], 3).astype(np.float32)
    fused = dict(lambda_l2=1.0)
    res = {}
    for seg in ("1", "4", "8"):
        _asurf_lib.config_set("ASURF_WALK_SEG", seg)
        res[seg] = hip.render_fused(sc, o, d, opts, gt, fused, counters=True)
        _asurf_lib.config_set("ASURF_WALK_SEG", None)
    rgb1, g1, c1 = res["1"]
    assert c1[1] / o.shape[0] > 20          # V_a: many active voxels per ray
    for seg in ("4", "8"):
        rgb, g, c = res[seg]
        assert np.array_equal(rgb, rgb1), f"seg {seg}: {np.abs(rgb - rgb1).max()}"
        assert (g["mask"] == g1["mask"]).all()
        for k in ("sh", "density", "surface"):
            grad_close(g[k], g1[k], 1e-5, f"seg {seg} grad_{k}")
    rgb_o, go = oracle.render_fused(sc, o, d, opts, gt, fused)[:2]
    assert np.abs(rgb1 - rgb_o).max() <= 2e-5
    for k in ("sh", "density", "surface"):
        grad_close(g1[k], go[k], 2e-4, f"grad_{k} vs oracle")


def test_ray_ordering_inside_the_fused_call_is_transparent(hip):
    """asurf_fused_params::order_rays: a shuffled batch of 70 000 rays rendered as given (1), ordered inside the call
    (2) and left to the library (0: the first call renders as given, the probe it leaves makes the later ones order):
    the colours come back where the caller's rays are, gradients and masks are those of the batch."""
    import torch
    from alphasurf_amd import csrc_api as _C, synth
    sc = synth.make_sparse_scene(96, 9, seed=4, trunc_a=4.0)
    o, d = synth.pinhole_rays(synth.orbit_camera(2), 265, 265)
    perm = np.random.RandomState(5).permutation(o.shape[0])
    o, d = o[perm], d[perm]
    assert o.shape[0] >= 65536
    gt = np.random.RandomState(6).rand(o.shape[0], 3).astype(np.float32)
    g, r, op = hip.grid_spec(sc), hip.rays_spec(o, d), hip.opt_spec(OPTION_SETS["syn_yaml"])
    gt_t = torch.from_numpy(gt).cuda()
    fused = dict(lambda_l2=1.0, lambda_l_entropy=1e-4, lambda_conv_mode_samp=1e-6)
    args = [fused.get(k, dflt) for k, dflt in hip.FUSED_ORDER]

    def run(order_rays):
        gr = hip.zero_grads(g)
        rgb = torch.zeros_like(gt_t)
        _C.volume_render_surf_trav_fused(g, r, op, gt_t, *args, rgb, gr, order_rays=order_rays)
        torch.cuda.synchronize()
        return rgb.cpu().numpy(), hip.grads_np(gr)

    rgb1, g1 = run(1)
    for mode in (2, 0, 0, 0):
        rgb2, g2 = run(mode)
        assert np.abs(rgb2 - rgb1).max() <= 2.5e-7, mode
        assert (g2["mask"] == g1["mask"]).all()
        for k in ("sh", "density", "surface"):
            grad_close(g2[k], g1[k], 2e-5, f"order_rays {mode} grad_{k}")


def test_a_declared_image_batch_is_taken_as_coherent(hip):
    """asurf_fused_params::image_width states that the rays come in image order: under order_rays = 0 the call then
    neither consults nor queues the coherence probe (the library's own spans show which kernels a call launched); the
    same batch without the declaration ends with the probe.  Same colours, gradients and masks either way."""
    import torch
    from alphasurf_amd import csrc_api as _C, synth
    sc = synth.make_sparse_scene(96, 9, seed=4, trunc_a=4.0)
    W = 265
    o, d = synth.pinhole_rays(synth.orbit_camera(2), W, W)
    assert o.shape[0] >= 65536
    gt = np.random.RandomState(6).rand(o.shape[0], 3).astype(np.float32)
    g, r, op = hip.grid_spec(sc), hip.rays_spec(o, d), hip.opt_spec(OPTION_SETS["syn_yaml"])
    gt_t = torch.from_numpy(gt).cuda()
    fused = dict(lambda_l2=1.0, lambda_l_entropy=1e-4, lambda_conv_mode_samp=1e-6)
    args = [fused.get(k, dflt) for k, dflt in hip.FUSED_ORDER]

    def run(image_width):
        gr = hip.zero_grads(g)
        rgb = torch.zeros_like(gt_t)
        _C.volume_render_surf_trav_fused(g, r, op, gt_t, *args, rgb, gr, order_rays=0, image_width=image_width)
        torch.cuda.synchronize()
        return rgb.cpu().numpy(), hip.grads_np(gr), [n for n, _ in _asurf_lib.profile_read_spans()]

    _asurf_lib.profile_enable(True)
    try:
        run(0)
        rgb0, g0, names0 = run(0)
        run(W)
        rgb1, g1, names1 = run(W)
    finally:
        _asurf_lib.profile_enable(False)
    assert "k_order_probe" in names0, names0
    assert "k_order_probe" not in names1 and "ray_order" not in names1, names1
    assert np.abs(rgb1 - rgb0).max() <= 2.5e-7
    assert (g1["mask"] == g0["mask"]).all()
    for k in ("sh", "density", "surface"):
        grad_close(g1[k], g0[k], 2e-5, f"image_width grad_{k}")


def test_fallback_kernels_beside_the_backward_pipeline(hip, oracle):
    """A scene that hands many rays to the one-kernel path (dense noisy shell, fake samples in every voxel): the fused
    call as it is (one stream) against the same call with its fallback kernels on the library's side stream
    (ASURF_SIDE_STREAM=1), and against the oracle."""
    from alphasurf_amd import synth
    sc = synth.make_sparse_scene(40, 4, seed=7, band=3.5, level_set=(-0.4, 0.0, 0.4), trunc_a=4.0, fake_sample_std=0.9)
    o, d = synth.random_rays(3000, 41)
    gt = np.random.RandomState(8).rand(o.shape[0], 3).astype(np.float32)
    opts = OPTION_SETS["fake_all"]
    fused = dict(lambda_l2=1.0, lambda_l_entropy=1e-3)
    rgb_a, g_a = hip.render_fused(sc, o, d, opts, gt, fused)
    _asurf_lib.config_set("ASURF_SIDE_STREAM", "1")
    rgb_b, g_b = hip.render_fused(sc, o, d, opts, gt, fused)
    _asurf_lib.config_set("ASURF_SIDE_STREAM", None)
    assert np.abs(rgb_a - rgb_b).max() <= 2.5e-7
    assert (g_a["mask"] == g_b["mask"]).all()
    rgb_o, go, _ = oracle.render_fused(sc, o, d, opts, gt, fused)
    assert np.abs(rgb_a - rgb_o).max() <= RGB_TOL_ORACLE
    for k in ("sh", "density", "surface"):
        grad_close(g_a[k], g_b[k], 2e-5, f"side stream grad_{k}")
        grad_close(g_a[k], go[k], GRAD_RTOL, f"grad_{k} vs oracle")
    assert (g_a["mask"].astype(bool) == go["mask"].astype(bool)).all()


def test_record_depth_follows_the_workspace(hip, oracle):
    """Fake samples in every voxel of a thick shell: most rays leave more than 32 samples.  With a workspace sized for
    32 records a ray they are differentiated by the walking kernel, with one sized for 128 (what the binding asks for
    with these options: asurf_workspace_bytes_rec) by the replaying kernels -- same gradients either way, and the
    oracle's."""
    from alphasurf_amd import synth
    sc = synth.make_sparse_scene(48, 4, seed=11, band=9.0, level_set=(-0.4, 0.0, 0.4), trunc_a=4.0, fake_sample_std=0.9)
    o, d = synth.random_rays(2500, 43)
    gt = np.random.RandomState(9).rand(o.shape[0], 3).astype(np.float32)
    opts = OPTION_SETS["fake_all"]
    fused = dict(lambda_l2=1.0, lambda_l_entropy=1e-3)
    size = sc["links"].shape
    assert (_asurf_lib.workspace_bytes(size, 2500, 64, 128) > _asurf_lib.workspace_bytes(size, 2500, 64, 64) >
            _asurf_lib.workspace_bytes(size, 2500, 64, 32) == _asurf_lib.workspace_bytes(size, 2500, 64))
    assert _asurf_lib.workspace_bytes(size, 2500, 64, 500) == _asurf_lib.workspace_bytes(size, 2500, 64, 128)
    rgb_s, g_s, cnt = hip.render_fused(sc, o, d, opts, gt, fused, rec_depth=32, counters=True)
    assert cnt[2] / cnt[3] > 20, "the scene is meant to leave rays with more than 32 samples"
    rgb_m, g_m = hip.render_fused(sc, o, d, opts, gt, fused, rec_depth=64)
    rgb_d, g_d = hip.render_fused(sc, o, d, opts, gt, fused)
    rgb_o, go, _ = oracle.render_fused(sc, o, d, opts, gt, fused)
    for rgb, g, what in ((rgb_s, g_s, "32 records"), (rgb_m, g_m, "64 records"), (rgb_d, g_d, "128 records")):
        assert np.abs(rgb - rgb_o).max() <= RGB_TOL_ORACLE, what
        for k in ("sh", "density", "surface"):
            grad_close(g[k], go[k], GRAD_RTOL, f"grad_{k} vs oracle, {what}")
        assert (g["mask"].astype(bool) == go["mask"].astype(bool)).all(), what


def test_ray_ordering_handles_many_origins_and_crowded_strips(hip):
    """The ordering's two sort paths on batches that are not one camera's pixels: rays with an origin each (every ray its
    own hash, strips of a ray or two), eight cameras mixed and shuffled, and 6 000 copies of one ray (one strip far beyond
    the 2 048 rays a block sorts at once: sorted in pieces).  Strip sort and rocPRIM's radix sort (ASURF_ORDER=radix) against
    the batch as given."""
    import torch
    from alphasurf_amd import csrc_api as _C, synth
    sc = synth.make_sparse_scene(64, 4, seed=11, trunc_a=4.0)
    rs = np.random.RandomState(12)
    o_r, d_r = synth.random_rays(30000, 13)                         # an origin each
    cams = [synth.pinhole_rays(synth.orbit_camera(k), 70, 70) for k in range(8)]
    o_c = np.concatenate([c[0] for c in cams])
    d_c = np.concatenate([c[1] for c in cams])
    o_d = np.repeat(cams[3][0][2477:2478], 6000, axis=0)            # one ray, 6 000 times
    d_d = np.repeat(cams[3][1][2477:2478], 6000, axis=0)
    o = np.concatenate([o_r, o_c, o_d]).astype(np.float32)
    d = np.concatenate([d_r, d_c, d_d]).astype(np.float32)
    perm = rs.permutation(o.shape[0])
    o, d = np.ascontiguousarray(o[perm]), np.ascontiguousarray(d[perm])
    assert o.shape[0] >= 65536
    gt = rs.rand(o.shape[0], 3).astype(np.float32)
    g, r, op = hip.grid_spec(sc), hip.rays_spec(o, d), hip.opt_spec(OPTION_SETS["syn_yaml"])
    gt_t = torch.from_numpy(gt).cuda()
    args = [dict(lambda_l2=1.0, lambda_l_entropy=1e-4).get(k, dflt) for k, dflt in hip.FUSED_ORDER]

    def run(order_rays):
        gr = hip.zero_grads(g)
        rgb = torch.zeros_like(gt_t)
        _C.volume_render_surf_trav_fused(g, r, op, gt_t, *args, rgb, gr, order_rays=order_rays)
        torch.cuda.synchronize()
        return rgb.cpu().numpy(), hip.grads_np(gr)

    rgb1, g1 = run(1)
    for form in (None, "radix"):
        _asurf_lib.config_set("ASURF_ORDER", form)
        rgb2, g2 = run(2)
        _asurf_lib.config_set("ASURF_ORDER", None)
        assert np.abs(rgb2 - rgb1).max() <= 2.5e-7, form
        assert (g2["mask"] == g1["mask"]).all()
        for k in ("sh", "density", "surface"):
            grad_close(g2[k], g1[k], 5e-5, f"{form} grad_{k}")
