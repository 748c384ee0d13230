"""CPU checks of the oracle's evaluation renders (surf_trav.cu:564-1705).
Pinned through the reference's PyTorch renderer on the option set where both see the same samples
(sigmoid alpha, no window, fake-sample weight driven to 0): its expected depth and its list of
intersections (svox2.py:2446-2452, 2746-2761) are in the r16_sh{9,4}_legacy goldens."""
import numpy as np
import pytest

from golden_util import load_case
from test_oracle_golden import scene_of

CASES = ["r16_sh9_legacy", "r16_sh4_legacy"]


def delta_scale(z):
    d = z["dirs"].astype(np.float64) * z["scaling"][None, :].astype(np.float64)
    return 1.0 / np.linalg.norm(d, axis=1)


@pytest.mark.parametrize("name", CASES)
def test_expected_depth_matches_reference(oracle, name):
    z, meta = load_case(name)
    sc = scene_of(oracle, z, meta)
    depth = oracle.render_eval(sc, z["origins"], z["dirs"], meta["opt"], "expected")
    ref = z["depth"].reshape(-1) * delta_scale(z)          # the reference's PyTorch depth is in grid units
    assert np.abs(ref).max() > 0
    assert np.abs(depth - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("name", CASES)
def test_extract_pts_matches_reference_intersections(oracle, name):
    z, meta = load_case(name)
    sc = scene_of(oracle, z, meta)
    depths, alphas = oracle.render_eval(sc, z["origins"], z["dirs"], meta["opt"], "extract_pts", param=0.1,
                                        max_sample=24)
    m = depths != 0
    pts = z["origins"][:, None, :] + z["dirs"][:, None, :] * depths[..., None]
    pts, al = pts[m], alphas[m]
    assert pts.shape[0] == z["intersections"].shape[0]        # same samples, same (ray-major) order
    assert np.abs(pts - z["intersections"]).max() <= 2e-5
    assert np.abs(al - z["intersect_alphas"]).max() <= 1e-5


def test_modes_are_consistent(oracle):
    z, meta = load_case("r16_sh9_legacy")
    sc = scene_of(oracle, z, meta)
    o, d, opt = z["origins"], z["dirs"], meta["opt"]
    depths, alphas = oracle.render_eval(sc, o, d, opt, "extract_pts", param=0.3, max_sample=24)
    first = oracle.render_eval(sc, o, d, opt, "sigma_thresh", param=0.3)
    amap = oracle.render_eval(sc, o, d, opt, "alpha", param=0.3)
    np.testing.assert_array_equal(first, depths[:, 0])
    np.testing.assert_array_equal(amap, alphas[:, 0])
    # mode depth: one of the ray's sample depths, or 0 when the ray carries too little weight
    all_d, _ = oracle.render_eval(sc, o, d, opt, "extract_pts", param=-1.0, max_sample=32)
    mode = oracle.render_eval(sc, o, d, opt, "mode", param=0.0)
    has = all_d[:, 0] != 0
    assert ((np.abs(all_d - mode[:, None]) <= 1e-6).any(1) | ~has).all() and (mode[~has] == 0).all()
    assert (oracle.render_eval(sc, o, d, opt, "mode", param=2.0) == 0).all()
    # expected depth lies between the first and the last sample of a ray that is nearly opaque
    exp = oracle.render_eval(sc, o, d, opt, "expected")
    last = np.where(all_d > 0, all_d, 0).max(1)
    assert (exp <= last + 1e-5).all()
    # the normal is the gradient of the trilinear surface field at the first sample with alpha > 0:
    # recompute it in numpy from the 8 corner values of the sample's voxel
    nrm = oracle.render_eval(sc, o, d, opt, "normal")
    hit = np.abs(nrm).sum(1) > 0
    assert hit.sum() > 100
    gp = (o + d * all_d[:, :1]).astype(np.float64) * z["scaling"][None, :] + z["offset"][None, :]
    links, surf = z["links"], z["surface"][:, 0].astype(np.float64)
    checked = 0
    for q in np.nonzero(hit)[0]:
        l = np.floor(gp[q]).astype(int)
        p = gp[q] - l
        if (p < 0.02).any() or (p > 0.98).any():
            continue                      # the sample sits on a voxel face: either voxel may own it
        c = np.array([[[surf[links[l[0] + i, l[1] + j, l[2] + k]] for k in (0, 1)] for j in (0, 1)] for i in (0, 1)])
        wx, wy, wz = np.array([1 - p[0], p[0]]), np.array([1 - p[1], p[1]]), np.array([1 - p[2], p[2]])
        gx = np.einsum("jk,j,k->", c[1] - c[0], wy, wz)
        gy = np.einsum("ik,i,k->", c[:, 1] - c[:, 0], wx, wz)
        gz = np.einsum("ij,i,j->", c[:, :, 1] - c[:, :, 0], wx, wy)
        assert np.abs(nrm[q] - np.array([gx, gy, gz])).max() <= 1e-3, q
        checked += 1
    assert checked > 50
