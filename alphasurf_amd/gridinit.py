"""Grid construction helpers (numpy): link layouts and SDF initialisations of the
reference's SparseGrid constructor, restated (svox2/svox2.py:647-681, 779-793;
svox2/utils.py:48-83)."""
import numpy as np


def _expand_bits(v):
    v = v.astype(np.int64)
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v


def morton_code_3(x, y, z):
    """Interleave bits: x -> bit 2, y -> bit 1, z -> bit 0 (svox2/utils.py:66-70)."""
    return (_expand_bits(x) << 2) + (_expand_bits(y) << 1) + _expand_bits(z)


def is_pow2(x):
    return x > 0 and (x & (x - 1)) == 0


def init_links(reso, use_z_order=False):
    """Dense link volume: Morton order for power-of-two cubes when use_z_order, else
    raster order (svox2.py:647-651)."""
    reso = [int(r) for r in reso]
    if use_z_order and reso[0] == reso[1] == reso[2] and is_pow2(reso[0]):
        a = np.arange(reso[0], dtype=np.int64)
        X, Y, Z = np.meshgrid(a, a, a, indexing="ij")
        return morton_code_3(X, Y, Z).astype(np.int32)
    return np.arange(reso[0] * reso[1] * reso[2], dtype=np.int32).reshape(reso)


def sphere_surface_init(links):
    """surface_init='sphere' (svox2.py:779-793): signed distance to the nearest of the
    concentric shells of radius 0.5, 2.5, 4.5, ... voxels around reso/2."""
    reso = links.shape
    coords = np.stack(np.meshgrid(*[np.arange(r) for r in reso], indexing="ij"), 0).reshape(3, -1).T
    center = np.asarray(reso, np.float32) / 2
    rs = np.sqrt(((coords.astype(np.float32) - center) ** 2).sum(-1)).astype(np.float32)
    rmax = np.sqrt(np.sum((np.asarray(reso, np.float32) / 2) ** 2))
    shells = (np.arange(0, rmax, 2, dtype=np.float32) + 0.5).astype(np.float32)
    dists = rs[:, None] - shells[None, :]
    pick = np.abs(dists).argmin(-1)
    out = np.zeros((int(links.max()) + 1, 1), np.float32)
    out[links.reshape(-1).astype(np.int64), 0] = dists[np.arange(dists.shape[0]), pick]
    return out


def sphere_bound_links(links):
    """use_sphere_bound (svox2.py:653-679): keep only vertices inside the unit sphere (plus a
    sqrt(3)/reso margin) and renumber the kept rows in the order of their original row ids."""
    reso = links.shape
    gsz = np.asarray(reso, np.float32)
    axes = [(np.arange(r, dtype=np.float32) - 0.5) for r in reso]
    X, Y, Z = np.meshgrid(*axes, indexing="ij")
    pts = np.stack([X, Y, Z], -1).reshape(-1, 3)
    pts = (1.0 / gsz - 1.0) + pts * (2.0 / gsz)
    keep = np.linalg.norm(pts, axis=-1) <= 1.0 + (3 ** 0.5) / gsz.max()
    flat = links.reshape(-1).astype(np.int64)
    n3 = flat.shape[0]
    data_mask = np.zeros(n3, np.int32)
    data_mask[flat[keep]] = 1
    rank = np.cumsum(data_mask) - 1
    out = np.full(n3, -1, np.int32)
    out[keep] = rank[flat[keep]].astype(np.int32)
    return out.reshape(reso)
