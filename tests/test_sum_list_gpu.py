"""The default call's occupancy pass (links_epoch = 0): the list of live rows the call vouches for itself by a checksum of
`links` (asurf_api.hip, prepare(); asurf_masks2.h, use_list 3).  It must leave the planes of the full pass -- same
picture bit for bit, same gradients -- while the values change, must notice `links` rewritten IN PLACE (same tensor, same
pointer) in the very call that follows, and must survive what else happens to a workspace: calls with a promise in
between, a captured full pass of another grid replayed over it.  ASURF_SUM_LIST=1 forces the form on the small test
grids (by default it starts at 128^3); =0 is the reference.  `pytest -m gpu`."""
import numpy as np
import pytest
from test_hip_parity import OPTION_SETS, hip, sparse_case  # noqa: F401

pytestmark = pytest.mark.gpu


def _setup(basis_dim=9):
    import torch
    import hip_util as H
    sc, o, d = sparse_case(basis_dim, level_set=(0.0,), trunc_a=4.0)
    g, r, op = H.grid_spec(sc), H.rays_spec(o, d), H.opt_spec(OPTION_SETS["syn_yaml"])
    gt = torch.from_numpy(np.random.RandomState(4).rand(o.shape[0], 3).astype(np.float32)).to(H.dev())
    return H, sc, g, r, op, gt


_ref_stream = []


def _fused(H, g, r, op, gt, reference=False):
    """reference: the full pass (ASURF_SUM_LIST=0), on a stream of its own -- the library's workspaces are per stream, and
    a full pass through the workspace under test would void the very list the test is about."""
    import torch
    from alphasurf_amd import csrc_api as _C, lib
    if not _ref_stream:
        _ref_stream.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    lib.config_set("ASURF_SUM_LIST", "0" if reference else "1")
    with torch.cuda.stream(_ref_stream[0] if reference else torch.cuda.current_stream()):
        gr = H.zero_grads(g)
        args = [dflt for _, dflt in H.FUSED_ORDER]   # (lambda_l2 = 1)
        rgb = torch.full_like(gt, float("nan"))
        _C.volume_render_surf_trav_fused(g, r, op, gt, *args, rgb, gr)
    torch.cuda.synchronize()
    return rgb.cpu().numpy(), H.grads_np(gr)


def _same(a, b, what):
    assert np.array_equal(a[0], b[0]), what
    assert (a[1]["mask"] == b[1]["mask"]).all(), what
    for k in ("sh", "density", "surface"):
        np.testing.assert_allclose(a[1][k], b[1][k], rtol=1e-4, atol=1e-6 * max(1.0, float(np.abs(b[1][k]).max())), err_msg=what)


def test_list_under_a_checksum_equals_the_full_pass(hip):
    import torch
    from alphasurf_amd import lib
    H, sc, g, r, op, gt = _setup()
    rs = np.random.RandomState(0)
    for it in range(5):
        if it >= 2:   # the values change between calls, as in training (in place: same tensors)
            g.density_data.add_(torch.from_numpy(rs.randn(*sc["density"].shape).astype(np.float32) * 0.5).to(H.dev()))
            g.surface_data.add_(torch.from_numpy(rs.randn(*sc["surface"].shape).astype(np.float32) * 0.05).to(H.dev()))
        ref = _fused(H, g, r, op, gt, reference=True)
        _same(_fused(H, g, r, op, gt), ref, f"call {it}")   # call 0 builds list and sum, the others use them


def test_links_rewritten_in_place_are_seen(hip):
    from alphasurf_amd import lib
    H, sc, g, r, op, gt = _setup(4)
    _fused(H, g, r, op, gt)
    _fused(H, g, r, op, gt)
    keep = g.links.clone()
    half = g.links.shape[0] // 2
    for change in ("half away", "back again", "the other half away"):
        if change == "half away":
            g.links[:half] = -1            # same tensor, same pointer: only the checksum can tell
        elif change == "back again":
            g.links.copy_(keep)            # rows that were empty space hold vertices again
        else:
            g.links[half:] = -1
        ref = _fused(H, g, r, op, gt, reference=True)
        _same(_fused(H, g, r, op, gt), ref, change + ": the call that finds the sum changed")
        _same(_fused(H, g, r, op, gt), ref, change + ": the call that rebuilds the list")
        _same(_fused(H, g, r, op, gt), ref, change + ": the call that uses the new list")


def test_calls_faster_than_the_count_travels_and_calls_with_a_promise_in_between(hip):
    import torch
    from alphasurf_amd import csrc_api as _C, lib
    H, sc, g, r, op, gt = _setup(4)
    lib.config_set("ASURF_SUM_LIST", "0")
    ref = _C.volume_render_surf_trav(g, r, op).cpu().numpy()
    lib.config_set("ASURF_SUM_LIST", "1")
    outs = [_C.volume_render_surf_trav(g, r, op) for _ in range(6)]   # no synchronisation in between
    g.links_epoch = 77                                                # ... then the same grid under a promise
    outs += [_C.volume_render_surf_trav(g, r, op) for _ in range(2)]
    g.links_epoch = 0                                                 # ... and without again
    outs += [_C.volume_render_surf_trav(g, r, op) for _ in range(3)]
    torch.cuda.synchronize()
    for i, rgb in enumerate(outs):
        assert np.array_equal(rgb.cpu().numpy(), ref), i


def test_replayed_full_pass_of_another_grid_voids_the_list(hip):
    """A captured call (captured calls take the full pass) for grid B replayed over the workspace that holds A's list and
    sum: the replay clears the header, A's next call finds no sum to compare with and takes every row."""
    import torch
    import hip_util as H
    from alphasurf_amd import csrc_api as _C, lib
    sc, o, d = sparse_case(4, level_set=(0.0,))
    links_b = sc["links"].copy()
    links_b[links_b.shape[0] // 2:] = -1
    links_a = sc["links"].copy()
    links_a[: links_a.shape[0] // 2] = -1
    sc_a, sc_b = dict(sc, links=links_a), dict(sc, links=links_b)
    lib.config_set("ASURF_SUM_LIST", "0")
    ref_a = H.render_forward(sc_a, o, d, OPTION_SETS["syn_yaml"])
    ref_b = H.render_forward(sc_b, o, d, OPTION_SETS["syn_yaml"])
    assert not np.array_equal(ref_a, ref_b)
    lib.config_set("ASURF_SUM_LIST", "1")
    r, op = H.rays_spec(o, d), H.opt_spec(OPTION_SETS["syn_yaml"])
    ga, gb = H.grid_spec(sc_a), H.grid_spec(sc_b)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        _C.volume_render_surf_trav(gb, r, op)       # (sizes the stream's workspace before the capture)
        a0 = _C.volume_render_surf_trav(ga, r, op)
    s.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        rgb_b = _C.volume_render_surf_trav(gb, r, op)
    with torch.cuda.stream(s):
        a1 = _C.volume_render_surf_trav(ga, r, op)       # builds A's list and sum
        a2 = _C.volume_render_surf_trav(ga, r, op)       # uses them
        graph.replay()                                    # B's full pass over the same workspace
        b_np = rgb_b.clone()
        a3 = _C.volume_render_surf_trav(ga, r, op)       # the host says "listed", the header says nothing
        a4 = _C.volume_render_surf_trav(ga, r, op)
        a5 = _C.volume_render_surf_trav(ga, r, op)
    s.synchronize()
    assert np.array_equal(b_np.cpu().numpy(), ref_b)
    for i, t in enumerate((a0, a1, a2, a3, a4, a5)):
        assert np.array_equal(t.cpu().numpy(), ref_a), i
