"""CPU checks of the oracle's cubic_extract_iso_pts (svox2_kernel.cu:248-372).  Parity unpinned as a whole (the
reference's kernel needs CUDA); its cubic, root solver and trilerp are the renderer's pinned ones.  Checked here by
what the extraction must satisfy: every point lies in its cell, on its line, on the level set, above the mask
threshold; a plane has the known answer."""
import numpy as np


def dense_links(shape):
    return np.arange(int(np.prod(shape)), dtype=np.int32).reshape(shape)


def trilerp(vol, p):
    l = np.minimum(np.floor(p).astype(int), np.asarray(vol.shape) - 2)
    w = p - l
    acc = 0.0
    for c in range(8):
        i, j, k = (c >> 2) & 1, (c >> 1) & 1, c & 1
        acc += vol[l[0] + i, l[1] + j, l[2] + k] * (w[0] if i else 1 - w[0]) * (w[1] if j else 1 - w[1]) * \
            (w[2] if k else 1 - w[2])
    return acc


def test_plane_has_the_known_answer(oracle):
    shape = (5, 6, 7)
    links = dense_links(shape)
    xs = np.arange(shape[0], dtype=np.float32)[:, None, None] * np.ones(shape, np.float32)
    level = (xs - 2.25).reshape(-1)                      # plane x = 2.25
    mask = np.ones(level.shape, np.float32)
    cells = links[:-1, :-1, :-1].reshape(-1)
    n = 3
    out = oracle.cubic_extract_iso_pts(links, level, mask, cells, n, 0.5)
    assert out.shape == (cells.size, 3 * n * n, 3)
    got = out.reshape(-1, 3)
    got = got[~(got == 0).all(axis=1)]
    # only the x-lines of the cells with x = 2 cross the plane: n*n points per such cell
    assert got.shape[0] == (shape[1] - 1) * (shape[2] - 1) * n * n
    assert np.allclose(got[:, 0], 2.25, atol=1e-6)
    # the slot layout: [i*n*3 + j*3 + dir], x-lines (dir 0) sit at (y + i/(n-1), z + j/(n-1))
    cid = int(np.where(cells == links[2, 1, 3])[0][0])
    for i in range(n):
        for j in range(n):
            assert np.allclose(out[cid, i * n * 3 + j * 3 + 0], [2.25, 1 + i / (n - 1), 3 + j / (n - 1)], atol=1e-6)
            assert not out[cid, i * n * 3 + j * 3 + 1].any() and not out[cid, i * n * 3 + j * 3 + 2].any()
    # mask below the threshold, or a threshold above it: nothing
    assert not oracle.cubic_extract_iso_pts(links, level, mask, cells, n, 1.5).any()


def test_points_lie_on_the_level_set_inside_their_cells(oracle):
    rs = np.random.RandomState(5)
    shape = (9, 8, 10)
    links = np.full(shape, -1, np.int32)
    keep = rs.rand(*shape) < 0.9
    links[keep] = rs.permutation(int(keep.sum())).astype(np.int32)
    g = np.stack(np.meshgrid(*[np.arange(s) for s in shape], indexing="ij"), -1).astype(np.float64)
    sdf_vol = np.linalg.norm(g - np.array([4.2, 3.6, 4.9]), axis=-1) - 2.7 + rs.randn(*shape) * 0.05
    dens_vol = rs.rand(*shape) * 2
    level = np.zeros(int(keep.sum()), np.float32)
    mask = np.zeros_like(level)
    level[links[keep]] = sdf_vol[keep]
    mask[links[keep]] = dens_vol[keep]
    cells = np.arange(links.size, dtype=np.int32)          # every cell id, including border and empty ones
    n, thresh = 4, 0.8
    out = oracle.cubic_extract_iso_pts(links, level, mask, cells, n, thresh)
    hit = ~(out == 0).all(axis=-1)
    assert 50 < hit.sum()
    sv, dv = sdf_vol.astype(np.float32).astype(np.float64), dens_vol.astype(np.float32).astype(np.float64)
    for cid, slot in zip(*np.where(hit)):
        x, y, z = np.unravel_index(cid, shape)
        assert x < shape[0] - 1 and y < shape[1] - 1 and z < shape[2] - 1
        assert (links[x:x + 2, y:y + 2, z:z + 2] >= 0).all()
        p = out[cid, slot].astype(np.float64)
        loc = p - [x, y, z]
        assert (loc >= -1e-6).all() and (loc <= 1 + 1e-6).all()
        i, j, d = slot // (3 * n), (slot // 3) % n, slot % 3
        fixed = [a for a in range(3) if a != d]
        assert np.allclose(loc[fixed], [i / (n - 1), j / (n - 1)], atol=1e-6)
        assert abs(trilerp(sv, np.minimum(p, np.array(shape) - 1 - 1e-9))) < 1e-5
        assert trilerp(dv, np.minimum(p, np.array(shape) - 1 - 1e-9)) >= thresh - 1e-5
