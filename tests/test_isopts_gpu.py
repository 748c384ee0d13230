"""cubic_extract_iso_pts / SparseGrid.extract_pts through the C ABI against the oracle.
`pytest -m gpu`."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def scene(seed, shape, keep_frac=0.9):
    rs = np.random.RandomState(seed)
    links = np.full(shape, -1, np.int32)
    keep = rs.rand(*shape) < keep_frac
    n = int(keep.sum())
    links[keep] = rs.permutation(n).astype(np.int32)
    g = np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"), -1).astype(np.float64)
    c = (np.asarray(shape) - 1) * np.array([0.52, 0.45, 0.5])
    sdf = np.linalg.norm(g - c, axis=-1) - 0.3 * min(shape) + rs.randn(*shape) * 0.08
    level = np.zeros(n, np.float32)
    mask = np.zeros(n, np.float32)
    level[links[keep]] = sdf[keep]
    mask[links[keep]] = (rs.rand(*shape) * 2)[keep]
    return links, level, mask


@pytest.mark.parametrize("seed,shape,n_sample,thresh", [(0, (12, 10, 14), 5, 0.8), (1, (17, 17, 17), 2, -100.0),
                                                       (2, (8, 20, 9), 7, 0.0)])
def test_iso_pts_vs_oracle(oracle, seed, shape, n_sample, thresh):
    import torch
    from alphasurf_amd import csrc_api as _C
    links, level, mask = scene(seed, shape)
    cells = np.random.RandomState(seed).permutation(links.size).astype(np.int32)   # border and empty ids included
    want = oracle.cubic_extract_iso_pts(links, level, mask, cells, n_sample, thresh)
    got = _C.cubic_extract_iso_pts(torch.from_numpy(links).cuda(), torch.from_numpy(level).cuda()[:, None],
                                   torch.from_numpy(mask).cuda()[:, None], torch.from_numpy(cells).cuda(),
                                   n_sample, thresh).cpu().numpy()
    assert got.shape == want.shape and (want != 0).any()
    assert np.array_equal((got == 0).all(-1), (want == 0).all(-1))
    assert np.array_equal(got, want)            # same cubic, solver and trilerp arithmetic: bit for bit


def test_errors_and_empty(oracle):
    import torch
    from alphasurf_amd import csrc_api as _C
    links, level, mask = scene(3, (6, 6, 6))
    lk, lv, mk = torch.from_numpy(links).cuda(), torch.from_numpy(level).cuda(), torch.from_numpy(mask).cuda()
    none = torch.zeros(0, dtype=torch.int32, device="cuda")
    assert _C.cubic_extract_iso_pts(lk, lv, mk, none, 3, 0.0).shape == (0, 27, 3)
    ids = torch.arange(10, dtype=torch.int32, device="cuda")
    with pytest.raises(RuntimeError):
        _C.cubic_extract_iso_pts(lk, lv, mk, ids, 1, 0.0)          # n_sample - 1 divides
    with pytest.raises(RuntimeError):
        _C.cubic_extract_iso_pts(lk, lv, mk, ids.long(), 3, 0.0)
    with pytest.raises(RuntimeError):
        _C.cubic_extract_iso_pts(lk, lv.cpu(), mk, ids, 3, 0.0)


def test_extract_pts_through_the_grid(oracle):
    """SparseGrid.extract_pts / sample_surf_pts (svox2.py:4525-4580): every non-empty cell in batches, zero rows
    dropped, optional world coordinates -- the oracle's points of the same arrays, in the same order."""
    import torch
    import svox2
    grid = svox2.SparseGrid(reso=32, basis_dim=4, use_sphere_bound=False, device=torch.device("cuda:0"),
                            surface_type=svox2.SURFACE_TYPE_SDF, surface_init="sphere")
    gen = torch.Generator(device="cpu").manual_seed(0)
    grid.density_data.data[:] = torch.rand(grid.density_data.shape, generator=gen).cuda() * 2
    links = grid.links.cpu().numpy()
    cells = np.where(links.reshape(-1) >= 0)[0].astype(np.int32)
    for lv_set, thresh, n in ((0.0, 0.5, 3), (0.05, 1.0, 2)):
        want = oracle.cubic_extract_iso_pts(links, grid.surface_data.data.cpu().numpy().reshape(-1) - np.float32(lv_set),
                                            grid.density_data.data.cpu().numpy().reshape(-1), cells, n, thresh)
        want = want.reshape(-1, 3)
        want = want[~(want == 0).all(-1)]
        pts = grid.extract_pts(density_thresh=thresh, n_sample=n, batch_size=5000, surf_lv_set=lv_set)
        assert pts.shape[0] > 1000 and np.array_equal(pts.cpu().numpy(), want)
    world = grid.extract_pts(density_thresh=1.0, n_sample=2, batch_size=5000, surf_lv_set=0.05, to_world=True,
                             scene_scale=2.0)
    assert torch.allclose(world, grid.grid2world(pts) / 2.0)
    assert grid.extract_pts(density_thresh=5.0, n_sample=3).shape[0] == 0     # mask below the threshold everywhere
