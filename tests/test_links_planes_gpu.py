"""Small batches over planes of `links` alone (asurf_masks2.h: k_links_checksum, k_links_planes; the value gates in
k_sweep_wave_lazy's exact test, asurf_sweep.h).  The tables the sweep leaves must be those of the per-call planes, so the
picture is the same bit for bit and the gradients differ by the order of float atomics only; the planes must follow
`links` -- through the call's own checksum when the caller promises nothing, through links_epoch when it does -- and
must not follow the values.  The form is OFF by default (measured slower than the per-call planes it was built to
replace: asurf_api.hip, kLazyMaxRays); ASURF_LAZY turns it on here.  `pytest -m gpu`."""
import numpy as np
import pytest
from test_hip_parity import OPTION_SETS, hip, sparse_case  # noqa: F401

pytestmark = pytest.mark.gpu


def _both(hip, sc, o, d, opts, gt, fused, **kw):
    from alphasurf_amd import lib
    lib.config_set("ASURF_LAZY", None)
    ref = hip.render_fused(sc, o, d, opts, gt, fused, **kw)
    lib.config_set("ASURF_LAZY", "32768")
    got = hip.render_fused(sc, o, d, opts, gt, fused, **kw)
    return ref, got


def _same(ref, got, what=""):
    assert np.array_equal(got[0], ref[0]), what
    assert (got[1]["mask"] == ref[1]["mask"]).all(), what
    for k in ("sh", "density", "surface"):
        scale = max(1.0, float(np.abs(ref[1][k]).max()))
        np.testing.assert_allclose(got[1][k], ref[1][k], rtol=1e-4, atol=2e-6 * scale, err_msg=f"{what} {k}")


@pytest.mark.parametrize("optname", ["syn_yaml", "sigmoid_stop", "density_gate", "fake_limited", "no_sh_dpos", "spheric",
                                     "big_step"])
@pytest.mark.parametrize("basis_dim", [4, 9])
def test_links_planes_equal_value_planes(hip, basis_dim, optname):
    """(fake_all tests the `live` plane: the stepping walk, never the sweep)"""
    lv = (0.0,) if optname != "fake_limited" else (-0.3, 0.0, 0.4)
    sc, o, d = sparse_case(basis_dim, level_set=lv, trunc_a=3.0, fake_sample_std=0.8)
    gt = np.random.RandomState(3).rand(o.shape[0], 3).astype(np.float32)
    fused = dict(lambda_l2=0.7, lambda_l1=0.3, lambda_l_entropy=1e-4, lambda_conv_mode_samp=1e-6)
    ref, got = _both(hip, sc, o, d, OPTION_SETS[optname], gt, fused)
    _same(ref, got, optname)


def test_voxel_gate_on_densities_and_many_level_sets(hip):
    """The voxel gates themselves: half of the vertices far below sigma_thresh (voxels with all 8 corners below are
    skipped BEFORE their crossings are counted, surf_trav.cu:230-239: the truncation window of later samples depends on
    it), and five level sets (beyond the three the vertex-plane pass takes)."""
    sc, o, d = sparse_case(9, level_set=(-0.4, -0.2, 0.0, 0.2, 0.4), trunc_a=2.0)
    rs = np.random.RandomState(8)
    dens = sc["density"].copy()
    # whole z-slabs of vertices below the gate, so that voxels with all 8 corners below exist
    links = sc["links"]
    low = np.zeros(dens.shape[0], bool)
    zz = np.arange(links.shape[2])
    sel = links[:, :, (zz // 3) % 2 == 0]
    low[sel[sel >= 0]] = True
    dens[low] = -5.0 + rs.rand(int(low.sum()), 1).astype(np.float32)
    sc = dict(sc, density=dens)
    gt = rs.rand(o.shape[0], 3).astype(np.float32)
    opts = dict(sigma_thresh=0.1, stop_thresh=0.0, truncated_vol_render=True, trunc_vol_weight_min=1e-10)
    ref, got = _both(hip, sc, o, d, opts, gt, dict(lambda_l2=1.0))
    assert (ref[1]["mask"] != 0).sum() > 100
    _same(ref, got)


def test_dense_scene_hands_rays_back(hip):
    """A scene in which nearly every voxel exists and the value planes are dense too: the sweep's lists overflow and rays
    go to the one-kernel paths, which then walk planes of links alone and apply the density gate themselves."""
    from alphasurf_amd import synth
    sc = synth.make_sparse_scene(40, 4, seed=11, band=40.0)    # every vertex inside the band: a dense grid
    rs = np.random.RandomState(2)
    sc = dict(sc, surface=(rs.randn(*sc["surface"].shape).astype(np.float32) * 0.5),
              density=(rs.randn(*sc["density"].shape).astype(np.float32)))
    o, d = synth.random_rays(1500, 77)
    gt = rs.rand(o.shape[0], 3).astype(np.float32)
    opts = dict(sigma_thresh=0.3, stop_thresh=0.0)
    from alphasurf_amd import lib
    lib.config_set("ASURF_LAZY", None)
    ref = hip.render_fused(sc, o, d, opts, gt, dict(lambda_l2=1.0))
    lib.config_set("ASURF_LAZY", "32768")
    got = hip.render_fused(sc, o, d, opts, gt, dict(lambda_l2=1.0))
    # (which rays are handed back depends on the capacity of the lists, and the one-kernel path sums a ray's colours in
    # another order than k_comp: an ulp, not bit equality)
    assert float(np.abs(got[0] - ref[0]).max()) <= 2.5e-7
    assert (got[1]["mask"] == ref[1]["mask"]).all()
    for k in ("sh", "density", "surface"):
        scale = max(1.0, float(np.abs(ref[1][k]).max()))
        np.testing.assert_allclose(got[1][k], ref[1][k], rtol=1e-4, atol=2e-6 * scale)


def _spec_call(H, g, r, op, gt_t, fused=None):
    import torch
    from alphasurf_amd import csrc_api as _C
    gr = H.zero_grads(g)
    f = dict(fused or {})
    args = [f.pop(k, dflt) for k, dflt in H.FUSED_ORDER]
    rgb = torch.zeros_like(gt_t)
    _C.volume_render_surf_trav_fused(g, r, op, gt_t, *args, rgb, gr)
    torch.cuda.synchronize()
    return rgb.cpu().numpy(), H.grads_np(gr)


@pytest.mark.parametrize("epoch", [0, 5])
def test_planes_follow_links_and_not_the_values(hip, epoch):
    """ONE links tensor that stays where it is.  The values change between calls (the planes must not care); then the
    links change IN PLACE -- the default call (links_epoch 0) must see it through its checksum, the promising caller
    passes a new value -- and change back."""
    import torch
    H = hip
    from alphasurf_amd import lib
    sc, o, d = sparse_case(9, level_set=(0.0,), trunc_a=4.0)
    gt = np.random.RandomState(4).rand(o.shape[0], 3).astype(np.float32)
    fused = dict(lambda_l2=1.0, lambda_l_entropy=1e-4)
    opts = OPTION_SETS["syn_yaml"]
    g, r, op = H.grid_spec(sc), H.rays_spec(o, d), H.opt_spec(opts)
    gt_t = torch.from_numpy(gt).to(H.dev())
    links_t = g.links
    g.links_epoch = epoch
    rs = np.random.RandomState(0)
    cur = dict(sc)
    for step in range(7):
        if step in (2, 3):     # the values move, as in training
            cur = dict(cur, density=cur["density"] + rs.randn(*cur["density"].shape).astype(np.float32) * 0.5,
                       surface=cur["surface"] + rs.randn(*cur["surface"].shape).astype(np.float32) * 0.05)
            g.density_data = torch.from_numpy(cur["density"]).to(H.dev())
            g.surface_data = torch.from_numpy(cur["surface"]).to(H.dev())
        if step == 4:          # a third of the scene goes away, in place
            links = cur["links"].copy()
            links[: links.shape[0] // 3] = -1
            cur = dict(cur, links=links)
            links_t.copy_(torch.from_numpy(links))
            if epoch:
                g.links_epoch = epoch + 1
        if step == 6:          # and comes back
            cur = dict(cur, links=sc["links"])
            links_t.copy_(torch.from_numpy(sc["links"]))
            if epoch:
                g.links_epoch = epoch + 2
        lib.config_set("ASURF_LAZY", None)
        ref = H.render_fused(cur, o, d, opts, gt, fused)
        lib.config_set("ASURF_LAZY", "32768")
        got = _spec_call(H, g, r, op, gt_t, fused)
        _same(ref, got, f"step {step}")


def test_graph_capture_and_big_batches_through_the_same_workspace(hip):
    """A captured small call replays over planes it did not build; a large batch (per-call planes, whose full pass voids
    the header) in between makes the next small call build its planes again."""
    import torch
    H = hip
    from alphasurf_amd import csrc_api as _C, synth, lib
    sc, o, d = sparse_case(4, level_set=(0.0,))
    opts = OPTION_SETS["syn_yaml"]
    ref = H.render_forward(sc, o, d, opts)
    lib.config_set("ASURF_LAZY", "32768")
    ob, db = synth.random_rays(40000, 5)           # beyond the small-batch bound
    ref_big = H.render_forward(sc, ob, db, opts)
    g, r, op = H.grid_spec(sc), H.rays_spec(o, d), H.opt_spec(opts)
    rb = H.rays_spec(ob, db)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        _C.volume_render_surf_trav(g, rb, op)      # sizes the stream's workspace for the large batch
        a0 = _C.volume_render_surf_trav(g, r, op)
        a1 = _C.volume_render_surf_trav(g, r, op)
    s.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        a2 = _C.volume_render_surf_trav(g, r, op)
    with torch.cuda.stream(s):
        graph.replay()
        a2a = a2.clone()
        b = _C.volume_render_surf_trav(g, rb, op)  # per-call planes over the same workspace
        graph.replay()                             # the captured small call must notice
        a2b = a2.clone()
        a3 = _C.volume_render_surf_trav(g, r, op)
    s.synchronize()
    assert np.array_equal(b.cpu().numpy(), ref_big)
    for t in (a0, a1, a2a, a2b, a3):
        assert np.array_equal(t.cpu().numpy(), ref)
