"""SparseGrid.save / load round trips (svox2.py:4693-4838), CPU only: the checkpoint keys and what the loader
restores -- in particular fake_sample_std in every form the training script leaves it in (a scheduled float, a
tensor, a trainable Parameter)."""
import numpy as np
import pytest
import torch

from alphasurf_amd import grid as G


def _grid(trainable=False):
    torch.manual_seed(0)
    g = G.SparseGrid(reso=6, basis_dim=4, device="cpu", surface_type=G.CONSTANTS["SURFACE_TYPE_SDF"],
                     surface_init="sphere", trainable_fake_sample_std=trainable, use_octree=False,
                     radius=[1.0, 1.5, 2.0], center=[0.1, 0.0, -0.2])
    with torch.no_grad():
        g.density_data.normal_()
        g.sh_data.normal_()
    g.level_set_data = torch.tensor([-0.1, 0.0, 0.1])
    return g


@pytest.mark.parametrize("kind", ["float", "tensor0", "tensor11", "parameter"])
def test_fake_sample_std_round_trip(tmp_path, kind):
    g = _grid(trainable=(kind == "parameter"))
    if kind == "float":
        g.fake_sample_std = 0.37
    elif kind == "tensor0":
        g.fake_sample_std = torch.tensor(0.37)
    elif kind == "tensor11":
        g.fake_sample_std = torch.tensor([[0.37]])
    else:
        with torch.no_grad():
            g.fake_sample_std.fill_(0.37)
    path = str(tmp_path / "ckpt.npz")
    g.save(path, step_id=123)
    h = G.SparseGrid.load(path, device="cpu")
    assert torch.is_tensor(h.fake_sample_std) and not isinstance(h.fake_sample_std, torch.nn.Parameter)
    assert abs(float(h.fake_sample_std.reshape(-1)[0]) - 0.37) < 1e-7
    assert h._to_cpp().fake_sample_std == pytest.approx(0.37, abs=1e-7)
    assert h.step_id == 123
    assert h.surface_type == g.surface_type and h.basis_type == g.basis_type and h.basis_dim == 4
    assert torch.equal(h.links, g.links) and h.links.dtype == torch.int32
    assert torch.equal(h.density_data.data, g.density_data.data)
    assert torch.equal(h.surface_data.data, g.surface_data.data)
    assert torch.equal(h.level_set_data, g.level_set_data)
    # sh_data is stored as float16 (svox2.py:4706)
    assert torch.allclose(h.sh_data.data, g.sh_data.data, atol=2e-3, rtol=1e-3)
    assert torch.allclose(h.radius, g.radius) and torch.allclose(h.center, g.center)
    assert h.capacity == g.capacity


def test_checkpoint_without_fake_sample_std_keeps_none(tmp_path):
    g = _grid()
    path = str(tmp_path / "c.npz")
    g.save(path)
    h = G.SparseGrid.load(path)
    assert h.fake_sample_std is None


def test_legacy_data_layout(tmp_path):
    """svox2.py:4755-4759: one `data` array, density in column 0; no surface."""
    g = _grid()
    data = np.concatenate([g.density_data.data.numpy(), g.sh_data.data.numpy()], axis=1)
    path = str(tmp_path / "old.npz")
    np.savez(path, data=data, links=g.links.numpy(), radius=g.radius.numpy(), center=g.center.numpy())
    h = G.SparseGrid.load(path)
    assert h.surface_type == G.CONSTANTS["SURFACE_TYPE_NONE"] and h.surface_data is None
    assert torch.equal(h.density_data.data, g.density_data.data)
    assert torch.equal(h.sh_data.data, g.sh_data.data)


def test_learned_basis_checkpoint_is_rejected(tmp_path):
    g = _grid()
    path = str(tmp_path / "b.npz")
    g.save(path)
    z = dict(np.load(path))
    z["basis_type"] = np.array(G.CONSTANTS["BASIS_TYPE_3D_TEXTURE"])
    np.savez(path, **z)
    with pytest.raises(NotImplementedError):
        G.SparseGrid.load(path)


def test_links_epoch_is_unique_per_grid_and_per_links_change():
    """asurf_grid::links_epoch must never repeat inside a process: grids that are built and freed back to back reuse
    their id() (and often the links pointer and version), so the value comes from one global counter."""
    seen = set()
    for _ in range(5):
        g = _grid()
        ep = g._to_cpp().links_epoch
        assert ep > 0 and ep not in seen
        assert g._to_cpp().links_epoch == ep          # stable while links stay what they are
        seen.add(ep)
        g.links = g.links.clone()                      # reassignment: a new value
        ep2 = g._to_cpp().links_epoch
        assert ep2 not in seen
        seen.add(ep2)
        del g
