"""CPU checks of the oracle's restatement of the per-iteration regularisers (N2).  The normal and
eikonal losses are pinned by gradients the reference's own PyTorch implementations produced
(tests/golden/gen_golden_reg.py); the TV and sparsify kernels have no PyTorch counterpart in the
reference and are checked against finite differences of the loss they differentiate."""
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN_DIR, grad_close


@pytest.fixture(scope="module")
def reg():
    z = dict(np.load(os.path.join(GOLDEN_DIR, "reg_r16.npz")))
    meta = json.loads(bytes(z.pop("meta")).decode())
    return z, meta


@pytest.mark.parametrize("name", ["l1", "l2", "l1_con", "l2_con_empty"])
def test_normal_loss_matches_reference(oracle, reg, name):
    z, meta = reg
    m = meta["normal_" + name]
    g, mask = oracle.surface_normal_grad_sparse(z["links"], z["surface"], z["rand_cells"], m["scaling"],
                                                lv_set=0.0, con_check=m["connectivity_check"],
                                                ignore_empty=m["ignore_empty"], use_l1=m["use_l1"])
    ref = z["normal_" + name]
    assert np.abs(ref).max() > 0
    grad_close(g, ref, 2e-4, name)
    assert ((np.abs(g.reshape(-1)) > 0) <= (mask > 0)).all()


def test_eikonal_matches_reference(oracle, reg):
    z, meta = reg
    g, loss, _ = oracle.surface_eikonal_grad_sparse(z["links"], z["surface"], z["rand_cells"],
                                                    meta["eikonal"]["scaling"], float(z["h"]))
    grad_close(g, z["eikonal"], 2e-5, "eikonal grad")
    assert abs(loss - meta["eikonal"]["loss"]) <= 1e-5 * abs(meta["eikonal"]["loss"])


def tv_loss(links, data, rand_cells, scale, idx, edge_value=None):
    """sum over cells of scale / n * sqrt(1e-9 + |forward difference|^2) with the per-axis weights
    size/256 folded in the way the kernel applies them (to the difference AFTER the norm)."""
    L = links
    sx, sy, sz = L.shape
    tot = 0.0
    for c in rand_cells:
        z = c % sz
        y = (c // sz) % sy
        x = c // (sz * sy)
        l0 = L[x, y, z]
        l1 = L[x, y, z + 1] if z + 1 < sz else 0
        l2 = L[x, y + 1, z] if y + 1 < sy else 0
        l3 = L[x + 1, y, z] if x + 1 < sx else 0
        d = 0.0 if edge_value is None else edge_value
        v0 = data[l0, idx] if l0 >= 0 else d
        v = [data[l, idx] if l >= 0 else d for l in (l3, l2, l1)]
        tot += np.sqrt(1e-9 + sum((float(a) - float(v0)) ** 2 for a in v))
    return scale * tot / len(rand_cells)


@pytest.mark.parametrize("surf", [False, True])
def test_tv_is_the_gradient_of_total_variation(oracle, reg, surf):
    """With cubic grids the axis weights are equal, so the kernel's output is (size/256) times the
    gradient of the TV loss above (loss_kernel.cu:782-790)."""
    z, _ = reg
    links = z["links"]
    rs = np.random.RandomState(3)
    n = z["surface"].shape[0]
    data = rs.randn(n, 2).astype(np.float32)
    cells = z["rand_cells"][::7][:300]
    kw = dict(surf=True, edge_value=0.3) if surf else {}
    g, mask = oracle.tv_grad_sparse(links, data, cells, 2.0, **kw)
    w = links.shape[0] / 256.0
    for idx in (0, 1):
        direction = np.sign(g[:, idx]).astype(np.float32)
        eps = 1e-3
        p, m = data.copy(), data.copy()
        p[:, idx] += eps * direction
        m[:, idx] -= eps * direction
        ev = 0.3 if surf else None
        fd = (tv_loss(links, p, cells, 2.0, idx, ev) - tv_loss(links, m, cells, 2.0, idx, ev)) / (2 * eps)
        an = float((g[:, idx].astype(np.float64) * direction).sum()) / w
        assert abs(fd - an) <= 2e-3 * abs(an), (idx, fd, an)
    assert ((np.abs(g).sum(1) > 0) <= (mask > 0)).all()


def test_surf_tv_alpha_dependency_boost(oracle, reg):
    z, _ = reg
    n = z["surface"].shape[0]
    cells = z["rand_cells"][:500]
    dens = np.full((n, 1), 0.02, np.float32)      # max raw alpha 0.02 < 0.1: gradient / max(0.2, 0.1)
    g0, _ = oracle.tv_grad_sparse(z["links"], z["surface"], cells, 1.0, surf=True, edge_value=-1.0,
                                  density=dens, alpha_dependency=False)
    g1, _ = oracle.tv_grad_sparse(z["links"], z["surface"], cells, 1.0, surf=True, edge_value=-1.0,
                                  density=dens, alpha_dependency=True)
    np.testing.assert_allclose(g1, g0 / 0.2, rtol=1e-4, atol=1e-12)


def test_sparsify_is_the_gradient_of_log_alpha(oracle, reg):
    z, _ = reg
    rs = np.random.RandomState(4)
    n = z["surface"].shape[0]
    alpha = rs.uniform(0.05, 2.0, size=(n, 1)).astype(np.float32)
    cells = z["rand_cells"][:1000]
    ga, gs, mask = oracle.alpha_surf_sparsify_grad_sparse(z["links"], alpha, z["surface"], cells, 0.5, 0.25,
                                                          True, 1.0, 0.1, -0.2)
    lk = z["links"].reshape(-1)[cells]
    lk = lk[lk >= 0]
    want_a = np.zeros_like(alpha)
    np.add.at(want_a[:, 0], lk, np.where(alpha[lk, 0] > 0.1, 0.5 / alpha[lk, 0], 0.0))   # d/da 0.5 log a
    np.testing.assert_allclose(ga, want_a, rtol=1e-5, atol=1e-9)
    sel = (z["surface"][lk, 0] > -0.2) & (alpha[lk, 0] < 1.0)
    want_s = np.zeros_like(alpha)
    np.add.at(want_s[:, 0], lk[sel], 0.25 / alpha[lk[sel], 0])
    np.testing.assert_allclose(gs, want_s, rtol=1e-5, atol=1e-9)
    assert mask[lk].all()
