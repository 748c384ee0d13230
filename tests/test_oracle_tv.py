"""CPU check of the oracle's dense TV value (loss_kernel.cu:73-117).  Parity unpinned -- the reference's kernel
needs CUDA and no test of it holds an expected value -- so the vectorised oracle is checked against a cell-by-cell
float64 loop written from the same source lines, and on cases with a known answer."""
import numpy as np


def cell_loop(links, data, start, end, ignore_edge):
    X, Y, Z = links.shape
    sc = [X / 256.0, Y / 256.0, Z / 256.0]
    total = 0.0
    for x in range(X - 1):
        for y in range(Y - 1):
            for z in range(Z - 1):
                l0 = links[x, y, z]
                if ignore_edge and l0 == 0:
                    continue
                for idx in range(start, end):
                    v0 = float(data[l0, idx]) if l0 >= 0 else 0.0
                    null = v0 if ignore_edge else 0.0
                    s = 1e-5
                    for k, ln in enumerate((links[x + 1, y, z], links[x, y + 1, z], links[x, y, z + 1])):
                        v = float(data[ln, idx]) if ln >= 0 else null
                        s += ((v - v0) * sc[k]) ** 2
                    total += np.sqrt(s)
    return total / ((X - 1) * (Y - 1) * (Z - 1))


def test_tv_value_matches_the_cell_loop(oracle):
    rs = np.random.RandomState(3)
    shape = (7, 9, 6)
    links = np.full(shape, -1, np.int32)
    keep = rs.rand(*shape) < 0.7
    links[keep] = rs.permutation(int(keep.sum())).astype(np.int32)
    data = rs.randn(int(keep.sum()), 5).astype(np.float32)
    assert (links == 0).sum() == 1
    for start, end, edge in ((0, 1, False), (0, 5, True), (2, 4, False), (2, 4, True)):
        got = oracle.tv_value(links, data, start, end, edge)
        want = cell_loop(links, data, start, end, edge)
        assert abs(got - want) <= 1e-6 * abs(want), (start, end, edge)


def test_tv_value_known_answers(oracle):
    # constant field: every term is sqrt(1e-5); two channels double it
    links = np.arange(4 * 4 * 4, dtype=np.int32).reshape(4, 4, 4)
    data = np.full((64, 2), 2.5, np.float32)
    eps = float(np.sqrt(np.float32(1e-5)))
    assert abs(oracle.tv_value(links, data, 0, 1, False) - eps) < 1e-7
    assert abs(oracle.tv_value(links, data, 0, 2, False) - 2 * eps) < 1e-7
    # a unit ramp along x on a 256-wide axis scale: |d| * X/256
    data[:, 0] = (np.arange(64) // 16).astype(np.float32)
    want = float(np.sqrt(1e-5 + (4 / 256.0) ** 2))
    assert abs(oracle.tv_value(links, data, 0, 1, False) - want) < 1e-6
    # ignore_edge drops cell (0,0,0), whose own link is 0; the empty cell (1,0,0) reads 0 at its centre and sees
    # its neighbours (values 2, 1, 1) as they are
    links2 = links.copy()
    links2[1, 0, 0] = -1
    t = float(np.sqrt(1e-5 + (2 * 4 / 256.0) ** 2 + 2 * (4 / 256.0) ** 2))
    assert abs(oracle.tv_value(links2, data, 0, 1, True) - (25 * want + t) / 27) < 1e-6
