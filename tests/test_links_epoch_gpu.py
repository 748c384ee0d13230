"""asurf_grid::links_epoch (include/asurf.h): the occupancy pass over the remembered list of non-empty brick rows must
leave the same planes as the full pass -- same picture, same gradients -- while densities and surfaces change between
calls, and must fall back to the full pass when the value, the links or the workspace change.  `pytest -m gpu`."""
import numpy as np
import pytest
from test_hip_parity import OPTION_SETS, hip, sparse_case  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("full_pass", [None, "split"])
def test_epoch_pass_equals_full_pass(hip, full_pass):
    """(full_pass "split": the full pass in three steps -- memset, k_rows_alive, the listed rows -- leaves the list that
    the later calls use; grids of 512^3 and more take it by default)"""
    import hip_util as H
    from alphasurf_amd import lib as _asurf_lib
    _asurf_lib.config_set("ASURF_MASKS", full_pass)
    sc, o, d = sparse_case(9, level_set=(0.0,), trunc_a=4.0)
    assert sc["links"].shape[2] % 4 == 0
    gt = np.random.RandomState(4).rand(o.shape[0], 3).astype(np.float32)
    fused = dict(lambda_l2=1.0, lambda_l_entropy=1e-4)
    ref_rgb, ref_g = H.render_fused(sc, o, d, OPTION_SETS["syn_yaml"], gt, fused)
    rs = np.random.RandomState(0)
    for it in range(4):
        # it 0 builds the list, 1.. use it; the values change in between as in training
        if it >= 2:
            sc = dict(sc, density=(sc["density"] + rs.randn(*sc["density"].shape).astype(np.float32) * 0.5),
                      surface=(sc["surface"] + rs.randn(*sc["surface"].shape).astype(np.float32) * 0.05))
            ref_rgb, ref_g = H.render_fused(sc, o, d, OPTION_SETS["syn_yaml"], gt, fused)
        rgb, g = H.render_fused(sc, o, d, OPTION_SETS["syn_yaml"], gt, fused, links_epoch=7)
        assert np.array_equal(rgb, ref_rgb), it
        assert (g["mask"] == ref_g["mask"]).all()
        for k in ("sh", "density", "surface"):
            np.testing.assert_allclose(g[k], ref_g[k], rtol=1e-4, atol=1e-6 * max(1.0, float(np.abs(ref_g[k]).max())))


@pytest.mark.parametrize("full_pass", [None, "split"])
def test_links_change_is_seen(hip, full_pass):
    import hip_util as H
    from alphasurf_amd import lib as _asurf_lib
    _asurf_lib.config_set("ASURF_MASKS", full_pass)
    sc, o, d = sparse_case(4, level_set=(0.0,))
    gt = np.random.RandomState(4).rand(o.shape[0], 3).astype(np.float32)
    fused = dict(lambda_l2=1.0)
    H.render_fused(sc, o, d, OPTION_SETS["syn_yaml"], gt, fused, links_epoch=11)
    H.render_fused(sc, o, d, OPTION_SETS["syn_yaml"], gt, fused, links_epoch=11)
    # half of the scene goes away: a new links tensor under a new value
    links = sc["links"].copy()
    links[: links.shape[0] // 2] = -1
    sc2 = dict(sc, links=links)
    ref_rgb, _ = H.render_fused(sc2, o, d, OPTION_SETS["syn_yaml"], gt, fused)
    rgb, _ = H.render_fused(sc2, o, d, OPTION_SETS["syn_yaml"], gt, fused, links_epoch=12)
    assert np.array_equal(rgb, ref_rgb)
    rgb, _ = H.render_fused(sc2, o, d, OPTION_SETS["syn_yaml"], gt, fused, links_epoch=12)
    assert np.array_equal(rgb, ref_rgb)
    # and the other way round: rows that were empty space hold vertices again
    ref_rgb, _ = H.render_fused(sc, o, d, OPTION_SETS["syn_yaml"], gt, fused)
    rgb, _ = H.render_fused(sc, o, d, OPTION_SETS["syn_yaml"], gt, fused, links_epoch=13)
    assert np.array_equal(rgb, ref_rgb)
    rgb, _ = H.render_fused(sc, o, d, OPTION_SETS["syn_yaml"], gt, fused, links_epoch=13)
    assert np.array_equal(rgb, ref_rgb)


def test_back_to_back_calls_before_the_count_has_arrived(hip):
    """Calls issued faster than the list's length travels to the host: the pass runs with the long grid meanwhile
    (and the query that says "not ready" must not surface as an error of the call)."""
    import torch
    import hip_util as H
    from alphasurf_amd import csrc_api as _C
    sc, o, d = sparse_case(4, level_set=(0.0,))
    gt = np.random.RandomState(4).rand(o.shape[0], 3).astype(np.float32)
    ref_rgb, _ = H.render_fused(sc, o, d, OPTION_SETS["syn_yaml"], gt, dict(lambda_l2=1.0))
    g, r, op = H.grid_spec(sc, links_epoch=21), H.rays_spec(o, d), H.opt_spec(OPTION_SETS["syn_yaml"])
    outs = [_C.volume_render_surf_trav(g, r, op) for _ in range(6)]  # no synchronisation in between
    torch.cuda.synchronize()
    for rgb in outs:
        assert np.array_equal(rgb.cpu().numpy(), ref_rgb)


def test_same_shape_grids_back_to_back():
    """Two different grids of one shape, built and freed one after the other (a test suite, an evaluation loop over
    checkpoints): the second must not render with the first one's list of live rows.  Each SparseGrid render (which
    passes its links_epoch) is compared with the same render through a spec that carries links_epoch = 0."""
    import torch
    import svox2
    from alphasurf_amd import csrc_api as _C
    assert torch.cuda.is_available()
    o, d = None, None
    for seed in range(4):
        g = svox2.SparseGrid(reso=32, center=[0.0, 0.0, 0.0], radius=[1.0, 1.0, 1.0], basis_dim=4,
                             device="cuda:0", basis_type=svox2.BASIS_TYPE_SH, surface_type=svox2.SURFACE_TYPE_SDF,
                             surface_init="sphere", use_sphere_bound=False)
        g.opt.backend = "surf_trav"
        g.opt.step_size = 0.5
        g.opt.sigma_thresh = -20.0
        g.level_set_data = torch.tensor([0.0], device="cuda:0")
        rs = np.random.RandomState(seed)
        # a different half-space of the volume is empty in every grid
        links = g.links.clone()
        ax, side = seed % 3, seed // 3
        sl = [slice(None)] * 3
        sl[ax] = slice(0, 14) if side == 0 else slice(18, 32)
        links[tuple(sl)] = -1
        g.links = links
        with torch.no_grad():
            g.density_data.data[:] = torch.from_numpy(rs.normal(2.0, 0.3, size=(g.capacity, 1)).astype(np.float32)).cuda()
            g.sh_data.data[:] = torch.from_numpy(rs.normal(0, 0.3, size=(g.capacity, 12)).astype(np.float32)).cuda()
            g.surface_data.data += torch.from_numpy((rs.rand(g.capacity, 1).astype(np.float32) - 0.5) * 0.2).cuda()
        if o is None:
            rr = np.random.RandomState(100)
            o = rr.randn(600, 3).astype(np.float32)
            o = 2.5 * o / np.linalg.norm(o, axis=1, keepdims=True)
            t = (rr.rand(600, 3).astype(np.float32) - 0.5) * 0.8
            d = t - o
            d /= np.linalg.norm(d, axis=1, keepdims=True)
        rays = svox2.Rays(torch.from_numpy(o).cuda(), torch.from_numpy(d).cuda())
        outs = [g.volume_render(rays, use_kernel=True)["rgb"].detach() for _ in range(3)]   # full pass + list, then short passes
        spec = g._to_cpp()
        assert spec.links_epoch != 0
        spec.links_epoch = 0
        ref = _C.volume_render_surf_trav(spec, rays._to_cpp(), g.opt._to_cpp())
        for i, rgb in enumerate(outs):
            # (an ulp, not bit equality: this dense, noisy scene can exhaust the spill list, and WHICH rays are then
            # handed to the one-kernel path -- which sums a ray's colours in another order -- depends on timing; a
            # stale list of live rows loses geometry and shows up at 1e-2 and more)
            diff = (rgb - ref).abs().max(dim=1).values
            assert float(diff.max()) <= 2.5e-7, (seed, i, int((diff > 0).sum()), float(diff.max()))
        del g, spec


def test_replayed_full_pass_of_another_grid_invalidates_the_list(hip):
    """A full occupancy pass captured into a HIP graph for grid B, replayed over the workspace that holds grid A's list
    of live rows: the replay rewrites every plane, so it must also void A's header on the device -- the next call
    for A (the host table still says "listed") has to rebuild every row, not only A's live ones."""
    import torch
    import hip_util as H
    from alphasurf_amd import csrc_api as _C
    sc, o, d = sparse_case(4, level_set=(0.0,))
    links_b = sc["links"].copy()
    links_b[links_b.shape[0] // 2:] = -1            # B: the other half is empty space ...
    sc_b = dict(sc, links=links_b)
    links_a = sc["links"].copy()
    links_a[: links_a.shape[0] // 2] = -1           # ... than in A
    sc_a = dict(sc, links=links_a)
    ref_a = H.render_forward(sc_a, o, d, OPTION_SETS["syn_yaml"])
    ref_b = H.render_forward(sc_b, o, d, OPTION_SETS["syn_yaml"])
    assert not np.array_equal(ref_a, ref_b)
    r, op = H.rays_spec(o, d), H.opt_spec(OPTION_SETS["syn_yaml"])
    ga = H.grid_spec(sc_a, links_epoch=0)
    ga.links_epoch = 41
    gb = H.grid_spec(sc_b, links_epoch=0)
    gb.links_epoch = 42
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            rgb = _C.volume_render_surf_trav(ga, r, op)
        _C.volume_render_surf_trav(gb, r, op)       # (sizes the stream's workspace before the capture)
        rgb_a0 = _C.volume_render_surf_trav(ga, r, op)
    s.synchronize()
    assert np.array_equal(rgb.cpu().numpy(), ref_a) and np.array_equal(rgb_a0.cpu().numpy(), ref_a)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        rgb_b = _C.volume_render_surf_trav(gb, r, op)
    with torch.cuda.stream(s):
        rgb_a1 = _C.volume_render_surf_trav(ga, r, op)       # the host table learns A again (full pass + list)
        rgb_a2 = _C.volume_render_surf_trav(ga, r, op)       # short pass
        graph.replay()                                        # B's full pass over the same workspace
        rgb_b_np = rgb_b.clone()
        rgb_a3 = _C.volume_render_surf_trav(ga, r, op)       # the host says "listed", the device header says no
        rgb_a4 = _C.volume_render_surf_trav(ga, r, op)
    s.synchronize()
    assert np.array_equal(rgb_b_np.cpu().numpy(), ref_b)
    for t in (rgb_a1, rgb_a2, rgb_a3, rgb_a4):
        assert np.array_equal(t.cpu().numpy(), ref_a)
