"""asurf_grid::links_epoch (include/asurf.h): the occupancy pass over the remembered list of non-empty brick rows must
leave the same planes as the full pass -- same picture, same gradients -- while densities and surfaces change between
calls, and must fall back to the full pass when the value, the links or the workspace change.  `pytest -m gpu`."""
import numpy as np
import pytest
from test_hip_parity import OPTION_SETS, hip, sparse_case  # noqa: F401

pytestmark = pytest.mark.gpu


def test_epoch_pass_equals_full_pass(hip):
    import hip_util as H
    sc, o, d = sparse_case(9, level_set=(0.0,), trunc_a=4.0)
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


def test_links_change_is_seen(hip):
    import hip_util as H
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
