"""Seeded synthetic workloads: sparse SDF shells, pinhole cameras.

numpy only; used by bench.py, the tests and __graft_entry__.smoke().  No dataset files.
"""
import numpy as np


def sphere_union_sdf(reso, n_spheres=8, seed=1234, center_range=(0.3, 0.7), radius_range=(0.10, 0.22)):
    """SDF (in voxel units, |grad| ~ 1) of a union of random spheres sampled at the
    grid vertices of a reso^3 volume."""
    rs = np.random.RandomState(seed)
    centers = rs.uniform(center_range[0], center_range[1], size=(n_spheres, 3)) * reso
    radii = rs.uniform(radius_range[0], radius_range[1], size=n_spheres) * reso
    ax = np.arange(reso, dtype=np.float32)
    sdf = np.full((reso, reso, reso), 1e9, np.float32)
    # evaluate slab by slab to bound memory at large reso
    for x0 in range(0, reso, 32):
        xs = ax[x0:x0 + 32]
        X, Y, Z = np.meshgrid(xs, ax, ax, indexing="ij")
        d = np.full(X.shape, 1e9, np.float32)
        for c, r in zip(centers, radii):
            dist = np.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) - r
            d = np.minimum(d, dist.astype(np.float32))
        sdf[x0:x0 + 32] = d
    return sdf


def _dilate(mask, it):
    for _ in range(it):
        m = mask.copy()
        m[1:] |= mask[:-1]
        m[:-1] |= mask[1:]
        m[:, 1:] |= mask[:, :-1]
        m[:, :-1] |= mask[:, 1:]
        m[:, :, 1:] |= mask[:, :, :-1]
        m[:, :, :-1] |= mask[:, :, 1:]
        mask = m
    return mask


def make_sparse_scene(reso, basis_dim, band=3.0, dilate=2, seed=1234, level_set=(0.0,),
                      n_spheres=8, fake_sample_std=1.0, trunc_a=1.0, center_range=(0.3, 0.7),
                      radius_range=(0.10, 0.22)):
    """A pruned-grid look-alike: vertices within `band` voxels of the surface (dilated
    `dilate` times) are active, rows are numbered in raster order (svox2.py:2940-2945),
    everything else links to -1.  Parameter distributions follow the reference's own
    gradcheck (test/test_render_gradcheck_surface.py:71-77)."""
    sdf = sphere_union_sdf(reso, n_spheres, seed, center_range, radius_range)
    active = _dilate(np.abs(sdf) < band, dilate)
    n = int(active.sum())
    links = np.full((reso, reso, reso), -1, np.int32)
    links[active] = np.arange(n, dtype=np.int32)
    rs = np.random.RandomState(seed + 1)
    surface = sdf[active].reshape(n, 1).astype(np.float32)
    density = rs.normal(0.5, 0.1, size=(n, 1)).astype(np.float32)
    sh = rs.normal(0.0, 0.1, size=(n, 3 * basis_dim)).astype(np.float32)
    sh[:, 0] = 0.5
    return dict(
        links=links, density=density, surface=surface, sh=sh,
        level_set=np.asarray(level_set, np.float32),
        # radius 1, center 0: _offset * gsz = 0.5 * reso, _scaling * gsz = 0.5 * reso
        offset=np.full(3, 0.5 * reso, np.float32), scaling=np.full(3, 0.5 * reso, np.float32),
        basis_dim=int(basis_dim), fake_sample_std=float(fake_sample_std),
        truncated_vol_render_a=float(trunc_a), surface_type=102,
    )


def look_at_c2w(eye, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """OpenCV-convention camera-to-world (x right, y down, z forward)."""
    eye = np.asarray(eye, np.float64)
    fwd = np.asarray(target, np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, eye
    return c2w


def orbit_camera(k, n_views=8, radius=4.0 * 2.0 / 3.0, elevation_deg=30.0):
    az = 2 * np.pi * k / n_views
    el = np.deg2rad(elevation_deg)
    eye = radius * np.array([np.cos(az) * np.cos(el), np.sin(az) * np.cos(el), np.sin(el)])
    return look_at_c2w(eye)


def pinhole_rays(c2w, width, height, fx=1111.11, fy=None, scale_focal=True):
    """Rays of a pinhole camera in image (row-major) order, like Camera.gen_rays
    (svox2/svox2.py:211-237).  With scale_focal the 800-px default focal is scaled to
    the requested width so the field of view stays that of the NeRF-synthetic cameras."""
    fy = fx if fy is None else fy
    if scale_focal:
        fx, fy = fx * width / 800.0, fy * height / 800.0
    yy, xx = np.meshgrid(np.arange(height, dtype=np.float64) + 0.5,
                         np.arange(width, dtype=np.float64) + 0.5, indexing="ij")
    d = np.stack([(xx - width * 0.5) / fx, (yy - height * 0.5) / fy, np.ones_like(xx)], -1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    d = d.reshape(-1, 3) @ c2w[:3, :3].T
    o = np.broadcast_to(c2w[:3, 3], d.shape)
    return np.ascontiguousarray(o, np.float32), np.ascontiguousarray(d, np.float32)


def random_rays(n, seed, spread=3.0, jitter=0.3):
    """Random rays aimed roughly at the origin (test/test_render_gradcheck_surface.py:112-118)."""
    rs = np.random.RandomState(seed)
    o = (rs.randn(n, 3) * spread).astype(np.float32)
    d = (-o + rs.randn(n, 3).astype(np.float32) * jitter)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    return o, d.astype(np.float32)


# surface_cuda_syn.yaml render options (opt/configs/surface_cuda_syn.yaml + config_util.py:224-250)
SYN_OPTS = dict(
    background_brightness=1.0, step_size=0.5, sigma_thresh=-1e4, stop_thresh=-1e4, near_clip=0.0,
    use_spheric_clip=False, last_sample_opaque=False, surf_fake_sample=False,
    surf_fake_sample_min_vox_len=0.1, limited_fake_sample=True, no_surf_grad_from_sh=False,
    alpha_activation_type=1, fake_sample_l_dist=False, fake_sample_normalize_surf=True,
    only_outward_intersect=True, truncated_vol_render=True, trunc_vol_weight_min=1e-10,
)
# bench scene: 16 spheres spread over the volume -> ~1.3 M active voxels at 256^3 (8 %),
# inside the 1-1.7 M range SURVEY.md section 8 assumes for a pruned lego-like grid
BENCH_SCENE = dict(n_spheres=16, center_range=(0.2, 0.8), radius_range=(0.08, 0.2), seed=1234)
SYN_FUSED = dict(lambda_l2=1.0, lambda_l1=0.0, lambda_l_entropy=1e-4, lambda_conv_mode_samp=1e-6,
                 l_dist_max_sample=64, l_di_alpha_thresh=0.1, surf_sparse_alpha_thresh=0.15)


def make_sparse_scene_torch(reso, basis_dim, device, band=3.0, dilate=2, seed=1234, level_set=(0.0,),
                            n_spheres=8, fake_sample_std=1.0, trunc_a=1.0, center_range=(0.3, 0.7),
                            radius_range=(0.10, 0.22)):
    """make_sparse_scene built on the device with torch (same geometry, torch's RNG for the
    parameters): for the 512^3 / 768^3 configurations, where the numpy version needs tens of GB
    of host temporaries.  Returns a dict of device tensors plus the host-side scalars."""
    import torch
    rs = np.random.RandomState(seed)
    centers = rs.uniform(center_range[0], center_range[1], size=(n_spheres, 3)) * reso
    radii = rs.uniform(radius_range[0], radius_range[1], size=n_spheres) * reso
    ax = torch.arange(reso, dtype=torch.float32, device=device)
    sdf = torch.empty((reso, reso, reso), dtype=torch.float32, device=device)
    for x0 in range(0, reso, 64):
        xs = ax[x0:x0 + 64]
        d = torch.full((xs.numel(), reso, reso), 1e9, dtype=torch.float32, device=device)
        for c, r in zip(centers, radii):
            dist = torch.sqrt((xs[:, None, None] - float(c[0])) ** 2 + (ax[None, :, None] - float(c[1])) ** 2 +
                              (ax[None, None, :] - float(c[2])) ** 2) - float(r)
            d = torch.minimum(d, dist)
        sdf[x0:x0 + 64] = d
    mask = sdf.abs() < band
    for _ in range(dilate):
        m = mask.clone()
        m[1:] |= mask[:-1]
        m[:-1] |= mask[1:]
        m[:, 1:] |= mask[:, :-1]
        m[:, :-1] |= mask[:, 1:]
        m[:, :, 1:] |= mask[:, :, :-1]
        m[:, :, :-1] |= mask[:, :, 1:]
        mask = m
    flat = mask.reshape(-1)
    ids = torch.cumsum(flat.to(torch.int32), 0, dtype=torch.int32) - 1
    links = torch.where(flat, ids, torch.full_like(ids, -1)).reshape(reso, reso, reso)
    n = int(flat.sum().item())
    gen = torch.Generator(device=device)
    gen.manual_seed(seed + 1)
    surface = sdf[mask].reshape(n, 1).contiguous()
    density = (torch.randn((n, 1), generator=gen, device=device) * 0.1 + 0.5)
    sh = torch.randn((n, 3 * basis_dim), generator=gen, device=device) * 0.1
    sh[:, 0] = 0.5
    return dict(
        links=links, density=density, surface=surface, sh=sh,
        level_set=torch.tensor(level_set, dtype=torch.float32, device=device),
        offset=np.full(3, 0.5 * reso, np.float32), scaling=np.full(3, 0.5 * reso, np.float32),
        basis_dim=int(basis_dim), fake_sample_std=float(fake_sample_std),
        truncated_vol_render_a=float(trunc_a), surface_type=102,
    )
