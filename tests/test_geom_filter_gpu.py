"""k_geom's light stage drops the work voxels whose cubic can have no root where a sample could be
(voxel_may_hold_root, asurf_kernels.h) before the reference's solver sees them.  The test is conservative by
construction; here it is compared with the unfiltered kernel (ASURF_DEBUG=32): a ray's RGB is a deterministic
function of its candidates, so it must come out bit for bit, and the masks must be identical.  (The same check over
millions of rays: variants/geom_filter_check.py.)  `pytest -m gpu`."""
import numpy as np
import pytest

from alphasurf_amd import lib as _asurf_lib

from alphasurf_amd import synth
from test_hip_parity import hip  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("level_set", [(0.0,), (-0.4, 0.0, 0.3)])
@pytest.mark.parametrize("extra", [{}, {"only_outward_intersect": False}, {"near_clip": 0.4}])
def test_filter_changes_nothing(hip, monkeypatch, level_set, extra):
    import hip_util as H
    rs = np.random.RandomState(3)
    sc = synth.make_sparse_scene(96, 4, trunc_a=5.0, level_set=level_set, **dict(synth.BENCH_SCENE, seed=17))
    o1, d1 = synth.random_rays(30000, 3)
    o2, d2 = synth.pinhole_rays(synth.orbit_camera(2), 160, 160)
    # rays through lattice points: axis-parallel components, faces and edges grazed
    n = 8000
    tgt = rs.randint(0, 96, size=(n, 3)).astype(np.float64)
    src = rs.randint(-96, 192, size=(n, 3)).astype(np.float64)
    keep = np.linalg.norm(tgt - src, axis=1) > 1
    o3 = ((src[keep] - 48) / 48).astype(np.float32)
    d3 = tgt[keep] - src[keep]
    d3 = (d3 / np.linalg.norm(d3, axis=1, keepdims=True)).astype(np.float32)
    o = np.concatenate([o1, o2, o3]).astype(np.float32)
    d = np.concatenate([d1, d2, d3]).astype(np.float32)
    gt = rs.rand(o.shape[0], 3).astype(np.float32)
    opts = dict(synth.SYN_OPTS, **extra)
    fused = dict(lambda_l2=1.0, lambda_l_entropy=1e-4)
    _asurf_lib.config_set("ASURF_DEBUG", "32")
    rgb_ref, g_ref = H.render_fused(sc, o, d, opts, gt, fused)
    _asurf_lib.config_set("ASURF_DEBUG", None)
    rgb, g = H.render_fused(sc, o, d, opts, gt, fused)
    assert (rgb != 0).any()
    assert np.array_equal(rgb, rgb_ref)
    assert (g["mask"] == g_ref["mask"]).all()
    for k in ("sh", "density", "surface"):
        scale = max(float(np.abs(g_ref[k]).max()), 1e-20)
        assert float(np.abs(g[k] - g_ref[k]).max()) <= 1e-5 * scale, k
