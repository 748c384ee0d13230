"""The device side of the multi-GPU exchange on its own (`pytest -m gpu`): asurf_rows_list must name the mask's rows in
increasing order (every rank derives the same list from the same union mask), and pack -> unpack is the identity on
those rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,frac", [(1, 1.0), (4095, 0.5), (4096, 0.0), (4097, 1.0), (1309130, 0.21), (100003, 0.003)])
def test_rows_list_is_the_ordered_nonzero(n, frac):
    import torch
    from alphasurf_amd import csrc_api as _C
    rs = np.random.RandomState(n % 1000)
    mask = torch.from_numpy(rs.rand(n) < frac).cuda()
    rows, count = _C.rows_list(mask)
    want = torch.nonzero(mask, as_tuple=False).view(-1).to(torch.int32)
    assert int(count.item()) == want.numel()
    assert torch.equal(rows[: want.numel()], want)


def test_pack_unpack_round_trip():
    import torch
    from alphasurf_amd import csrc_api as _C
    rs = np.random.RandomState(5)
    n, D = 50021, 27
    mask = torch.from_numpy(rs.rand(n) < 0.3).cuda()
    sh = torch.from_numpy(rs.randn(n, D).astype(np.float32)).cuda()
    de = torch.from_numpy(rs.randn(n, 1).astype(np.float32)).cuda()
    su = torch.from_numpy(rs.randn(n, 1).astype(np.float32)).cuda()
    rows, count = _C.rows_list(mask)
    U = int(count.item())
    packed = torch.empty(U * (D + 2), device="cuda")
    _C.rows_pack(rows, U, sh, de, su, packed)
    body = packed.view(U, D + 2)
    idx = rows[:U].long()
    assert torch.equal(body[:, :D], sh[idx]) and torch.equal(body[:, D], de[idx, 0]) and torch.equal(body[:, D + 1], su[idx, 0])
    sh2, de2, su2 = torch.zeros_like(sh), torch.zeros_like(de), torch.zeros_like(su)
    m2 = torch.zeros(n, dtype=torch.bool, device="cuda")
    _C.rows_unpack(rows, U, packed * 2.0, sh2, de2, su2, m2)
    assert torch.equal(m2, mask)
    assert torch.equal(sh2[idx], 2 * sh[idx]) and torch.equal(de2[idx], 2 * de[idx]) and torch.equal(su2[idx], 2 * su[idx])
    assert float(sh2[~mask].abs().max()) == 0.0
