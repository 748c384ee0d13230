"""ctypes binding of the CPU oracle (oracle/libasurf_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under alphasurf_amd/ may import this module.

Scenes are plain dicts of numpy arrays (see `make_scene`); render options are dicts
keyed by the reference's RenderOptions field names (svox2/svox2.py:33-67).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libasurf_oracle.so")

c_f = C.POINTER(C.c_float)
c_i = C.POINTER(C.c_int32)
c_u8 = C.POINTER(C.c_uint8)


class Grid(C.Structure):
    _fields_ = [
        ("links", c_i), ("density_data", c_f), ("surface_data", c_f), ("sh_data", c_f),
        ("level_set_data", c_f), ("capacity", C.c_int64), ("size", C.c_int32 * 3),
        ("level_set_num", C.c_int32), ("basis_dim", C.c_int32), ("basis_type", C.c_int32),
        ("surface_type", C.c_int32), ("offset", C.c_float * 3), ("scaling", C.c_float * 3),
        ("fake_sample_std", C.c_float), ("truncated_vol_render_a", C.c_float),
        ("background_nlayers", C.c_int32),
    ]


class Rays(C.Structure):
    _fields_ = [("origins", c_f), ("dirs", c_f), ("masks", c_u8), ("n_rays", C.c_int64)]


def setup_rays(sc, origins, dirs, opts=None):
    """float32 [Q, 8]: origin, unit direction (grid coordinates), tmin, tmax as the kernels' prologue computes them."""
    r = _rays(origins, dirs)
    g, o = _grid(sc), make_opts(opts)
    out = np.zeros((r.n_rays, 8), np.float32)
    _check(lib().oracle_setup_rays(C.byref(g), C.byref(r), C.byref(o), _f(out)), "oracle_setup_rays")
    return out


class RayVox(C.Structure):
    _fields_ = [("voxel_ls", c_i), ("vox_start_i", c_i), ("vox_num", c_i), ("n_vox", C.c_int64)]


OPT_FIELDS = [
    ("background_brightness", C.c_float, 1.0), ("step_size", C.c_float, 0.5),
    ("sigma_thresh", C.c_float, 1e-10), ("stop_thresh", C.c_float, 1e-7),
    ("near_clip", C.c_float, 0.0), ("use_spheric_clip", C.c_int32, 0),
    ("last_sample_opaque", C.c_int32, 0), ("surf_fake_sample", C.c_int32, 0),
    ("surf_fake_sample_min_vox_len", C.c_float, 0.1), ("limited_fake_sample", C.c_int32, 0),
    ("no_surf_grad_from_sh", C.c_int32, 0), ("alpha_activation_type", C.c_int32, 1),
    ("fake_sample_l_dist", C.c_int32, 1), ("fake_sample_normalize_surf", C.c_int32, 1),
    ("only_outward_intersect", C.c_int32, 0), ("truncated_vol_render", C.c_int32, 0),
    ("trunc_vol_weight_min", C.c_float, 0.0),
]


class RenderOptions(C.Structure):
    _fields_ = [(n, t) for n, t, _ in OPT_FIELDS]


class Grads(C.Structure):
    _fields_ = [("grad_density_out", c_f), ("grad_sh_out", c_f), ("grad_surface_out", c_f),
                ("grad_fake_sample_std_out", c_f), ("mask_out", c_u8)]


FUSED_FIELDS = [
    ("beta_loss", C.c_float, 0.0), ("sparsity_loss", C.c_float, 0.0),
    ("fused_surf_norm_reg_scale", C.c_float, 0.0), ("fused_surf_norm_reg_con_check", C.c_int32, 1),
    ("fused_surf_norm_reg_ignore_empty", C.c_int32, 0), ("lambda_l2", C.c_float, 1.0),
    ("lambda_l1", C.c_float, 0.0), ("lambda_l_dist", C.c_float, 0.0),
    ("lambda_l_entropy", C.c_float, 0.0), ("no_norm_weight_l_entropy", C.c_int32, 0),
    ("lambda_l_dist_a", C.c_float, 0.0), ("lambda_l_entropy_a", C.c_float, 0.0),
    ("lambda_l_samp_dist", C.c_float, 0.0), ("lambda_l_di", C.c_float, 0.0),
    ("l_di_alpha_thresh", C.c_float, 0.0), ("surf_sparse_alpha_thresh", C.c_float, 0.0),
    ("lambda_inplace_surf_sparse", C.c_float, 0.0), ("lambda_inwards_norm_loss", C.c_float, 0.0),
    ("lambda_conv_mode_samp", C.c_float, 0.0), ("l_dist_max_sample", C.c_int32, 64),
    ("global_n_rays", C.c_int64, 0),
]


class FusedParams(C.Structure):
    _fields_ = [("rgb_gt", c_f)] + [(n, t) for n, t, _ in FUSED_FIELDS]


class Stats(C.Structure):
    _fields_ = [("n_traversed", c_i), ("n_active", c_i), ("n_samples", c_i),
                ("max_dump", C.c_int32), ("dump_t", c_f), ("dump_alpha", c_f),
                ("dump_weight", c_f), ("dump_voxel", c_i), ("dump_flags", c_i), ("n_surf", c_i)]


_lib = None


def build(force=False):
    """Compile oracle/libasurf_oracle.so with g++ (building the checker is not using it)."""
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "asurf_oracle.cpp"))):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "libasurf_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_trace_voxel_lists.restype = C.c_int64
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _f(a):
    return a.ctypes.data_as(c_f) if a is not None else None


def _i(a):
    return a.ctypes.data_as(c_i) if a is not None else None


def _u8(a):
    return a.ctypes.data_as(c_u8) if a is not None else None


def make_scene(links, density, surface, sh, level_set, offset, scaling, basis_dim,
               fake_sample_std=1.0, truncated_vol_render_a=1.0, surface_type=102):
    """Normalise dtypes/contiguity; returns the dict every binding here consumes."""
    sc = dict(
        links=np.ascontiguousarray(links, dtype=np.int32),
        density=np.ascontiguousarray(density, dtype=np.float32).reshape(-1, 1),
        surface=np.ascontiguousarray(surface, dtype=np.float32).reshape(-1, 1),
        sh=np.ascontiguousarray(sh, dtype=np.float32),
        level_set=np.ascontiguousarray(level_set, dtype=np.float32).reshape(-1),
        offset=np.asarray(offset, dtype=np.float32).reshape(3),
        scaling=np.asarray(scaling, dtype=np.float32).reshape(3),
        basis_dim=int(basis_dim),
        fake_sample_std=float(fake_sample_std),
        truncated_vol_render_a=float(truncated_vol_render_a),
        surface_type=int(surface_type),
    )
    assert sc["sh"].shape == (sc["density"].shape[0], 3 * sc["basis_dim"])
    return sc


def _grid(sc):
    g = Grid()
    g.links = _i(sc["links"])
    g.density_data = _f(sc["density"])
    g.surface_data = _f(sc["surface"])
    g.sh_data = _f(sc["sh"])
    g.level_set_data = _f(sc["level_set"])
    g.capacity = sc["density"].shape[0]
    g.size = (C.c_int32 * 3)(*sc["links"].shape)
    g.level_set_num = sc["level_set"].shape[0]
    g.basis_dim = sc["basis_dim"]
    g.basis_type = 1
    g.surface_type = sc["surface_type"]
    g.offset = (C.c_float * 3)(*[float(x) for x in sc["offset"]])
    g.scaling = (C.c_float * 3)(*[float(x) for x in sc["scaling"]])
    g.fake_sample_std = sc["fake_sample_std"]
    g.truncated_vol_render_a = sc["truncated_vol_render_a"]
    g.background_nlayers = 0
    return g


def _rays(origins, dirs):
    origins = np.ascontiguousarray(origins, dtype=np.float32)
    dirs = np.ascontiguousarray(dirs, dtype=np.float32)
    masks = np.ones(origins.shape[0], dtype=np.uint8)
    r = Rays(_f(origins), _f(dirs), _u8(masks), origins.shape[0])
    r._keep = (origins, dirs, masks)
    return r


def make_opts(opts=None):
    o = RenderOptions()
    opts = dict(opts or {})
    opts.pop("backend", None)
    for n, t, d in OPT_FIELDS:
        v = opts.pop(n, d)
        setattr(o, n, t(int(v)) if t is C.c_int32 else t(float(v)))
    opts.pop("random_sigma_std", None)
    opts.pop("random_sigma_std_background", None)
    assert not opts, f"unknown render options {list(opts)}"
    return o


def make_fused(rgb_gt, fused=None):
    p = FusedParams()
    fused = dict(fused or {})
    p.rgb_gt = _f(rgb_gt)
    for n, t, d in FUSED_FIELDS:
        v = fused.pop(n, d)
        setattr(p, n, t(int(v)) if t in (C.c_int32, C.c_int64) else t(float(v)))
    assert not fused, f"unknown fused params {list(fused)}"
    return p


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


def zero_grads(sc, with_std=True):
    n = sc["density"].shape[0]
    return dict(
        density=np.zeros((n, 1), np.float32),
        sh=np.zeros_like(sc["sh"]),
        surface=np.zeros((n, 1), np.float32),
        fake_sample_std=np.zeros((1,), np.float32) if with_std else None,
        mask=np.zeros((n,), np.uint8),
    )


def _grads(gr):
    return Grads(_f(gr["density"]), _f(gr["sh"]), _f(gr["surface"]),
                 _f(gr.get("fake_sample_std")), _u8(gr.get("mask")))


def _stats(q, max_dump=0):
    st = dict(n_traversed=np.zeros(q, np.int32), n_active=np.zeros(q, np.int32),
              n_samples=np.zeros(q, np.int32))
    st["n_surf"] = np.zeros(q, np.int32)
    s = Stats(_i(st["n_traversed"]), _i(st["n_active"]), _i(st["n_samples"]), max_dump)
    s.n_surf = _i(st["n_surf"])
    if max_dump:
        st["dump_t"] = np.zeros((q, max_dump), np.float32)
        st["dump_alpha"] = np.zeros((q, max_dump), np.float32)
        st["dump_weight"] = np.zeros((q, max_dump), np.float32)
        st["dump_voxel"] = np.full((q, max_dump, 3), -1, np.int32)
        st["dump_flags"] = np.zeros((q, max_dump), np.int32)
        s.dump_t, s.dump_alpha, s.dump_weight = _f(st["dump_t"]), _f(st["dump_alpha"]), _f(st["dump_weight"])
        s.dump_voxel, s.dump_flags = _i(st["dump_voxel"]), _i(st["dump_flags"])
    return s, st


def render_forward(sc, origins, dirs, opts=None, max_dump=0):
    """volume_render_surf_trav: returns (rgb[Q,3], stats dict)."""
    r = _rays(origins, dirs)
    q = r.n_rays
    rgb = np.zeros((q, 3), np.float32)
    s, st = _stats(q, max_dump)
    g, o = _grid(sc), make_opts(opts)
    _check(lib().oracle_volume_render_surf_trav(C.byref(g), C.byref(r), C.byref(o), _f(rgb),
                                                C.byref(s)), "forward")
    return rgb, st


def render_backward(sc, origins, dirs, opts, grad_out, color_cache, grads=None):
    """volume_render_surf_trav_backward: accumulates into (and returns) the grads dict."""
    r = _rays(origins, dirs)
    grads = grads if grads is not None else zero_grads(sc)
    grad_out = np.ascontiguousarray(grad_out, np.float32)
    color_cache = np.ascontiguousarray(color_cache, np.float32)
    g, o, gg = _grid(sc), make_opts(opts), _grads(grads)
    _check(lib().oracle_volume_render_surf_trav_backward(
        C.byref(g), C.byref(r), C.byref(o), _f(grad_out), _f(color_cache), C.byref(gg)), "backward")
    return grads


def render_fused(sc, origins, dirs, opts, rgb_gt, fused=None, grads=None, max_dump=0):
    """volume_render_surf_trav_fused: returns (rgb, grads dict, stats dict)."""
    r = _rays(origins, dirs)
    q = r.n_rays
    grads = grads if grads is not None else zero_grads(sc)
    rgb_gt = np.ascontiguousarray(rgb_gt, np.float32)
    rgb = np.zeros((q, 3), np.float32)
    s, st = _stats(q, max_dump)
    g, o, gg, fp = _grid(sc), make_opts(opts), _grads(grads), make_fused(rgb_gt, fused)
    _check(lib().oracle_volume_render_surf_trav_fused(
        C.byref(g), C.byref(r), C.byref(o), C.byref(fp), _f(rgb), C.byref(gg), C.byref(s)), "fused")
    return rgb, grads, st


def trace_voxel_lists(sc, origins, dirs, opts=None):
    r = _rays(origins, dirs)
    q = r.n_rays
    g, o = _grid(sc), make_opts(opts)
    start = np.zeros(q, np.int32)
    num = np.zeros(q, np.int32)
    total = lib().oracle_trace_voxel_lists(C.byref(g), C.byref(r), C.byref(o), None,
                                           C.c_int64(0), _i(start), _i(num))
    vox = np.zeros((max(total, 1), 3), np.int32)
    lib().oracle_trace_voxel_lists(C.byref(g), C.byref(r), C.byref(o), _i(vox), C.c_int64(total),
                                   _i(start), _i(num))
    return vox[:total], start, num


def setup_rays(sc, origins, dirs, opts=None):
    """float32 [Q, 8]: origin, unit direction (grid coordinates), tmin, tmax as the kernels' prologue computes them."""
    r = _rays(origins, dirs)
    g, o = _grid(sc), make_opts(opts)
    out = np.zeros((r.n_rays, 8), np.float32)
    _check(lib().oracle_setup_rays(C.byref(g), C.byref(r), C.byref(o), _f(out)), "oracle_setup_rays")
    return out


class RayVox(C.Structure):
    _fields_ = [("voxel_ls", c_i), ("vox_start_i", c_i), ("vox_num", c_i), ("n_vox", C.c_int64)]


def _ray_vox(voxel_ls, start, num):
    voxel_ls = np.ascontiguousarray(voxel_ls, np.int32).reshape(-1, 3)
    start = np.ascontiguousarray(start, np.int32)
    num = np.ascontiguousarray(num, np.int32)
    rv = RayVox(_i(voxel_ls) if voxel_ls.size else None, _i(start), _i(num), voxel_ls.shape[0])
    rv._keep = (voxel_ls, start, num)
    return rv


def surface_forward(sc, origins, dirs, ray_vox, opts=None):
    """Legacy volume_render_surface: ray_vox = (voxel_ls[V,3], vox_start_i[Q], vox_num[Q])."""
    r = _rays(origins, dirs)
    rgb = np.zeros((r.n_rays, 3), np.float32)
    g, o, rv = _grid(sc), make_opts(opts), _ray_vox(*ray_vox)
    _check(lib().oracle_volume_render_surface(C.byref(g), C.byref(r), C.byref(rv), C.byref(o), _f(rgb)),
           "surface forward")
    return rgb


def surface_backward(sc, origins, dirs, ray_vox, opts, grad_out, color_cache, grads=None):
    r = _rays(origins, dirs)
    grads = grads if grads is not None else zero_grads(sc)
    grad_out = np.ascontiguousarray(grad_out, np.float32)
    color_cache = np.ascontiguousarray(color_cache, np.float32)
    g, o, gg, rv = _grid(sc), make_opts(opts), _grads(grads), _ray_vox(*ray_vox)
    _check(lib().oracle_volume_render_surface_backward(
        C.byref(g), C.byref(r), C.byref(rv), C.byref(o), _f(grad_out), _f(color_cache), C.byref(gg)),
        "surface backward")
    return grads


def surface_fused(sc, origins, dirs, ray_vox, opts, rgb_gt, beta_loss=0.0, sparsity_loss=0.0, grads=None):
    r = _rays(origins, dirs)
    grads = grads if grads is not None else zero_grads(sc)
    rgb_gt = np.ascontiguousarray(rgb_gt, np.float32)
    rgb = np.zeros((r.n_rays, 3), np.float32)
    g, o, gg, rv = _grid(sc), make_opts(opts), _grads(grads), _ray_vox(*ray_vox)
    _check(lib().oracle_volume_render_surface_fused(
        C.byref(g), C.byref(r), C.byref(rv), C.byref(o), _f(rgb_gt), C.c_float(beta_loss),
        C.c_float(sparsity_loss), _f(rgb), C.byref(gg)), "surface fused")
    return rgb, grads


def cuvol_forward(sc, origins, dirs, opts=None):
    """volume_render_cuvol (render_lerp_kernel_cuvol.cu:1120-1165); sc["surface"] is not read."""
    r = _rays(origins, dirs)
    rgb = np.zeros((r.n_rays, 3), np.float32)
    g, o = _grid(sc), make_opts(opts)
    _check(lib().oracle_volume_render_cuvol(C.byref(g), C.byref(r), C.byref(o), _f(rgb)), "cuvol forward")
    return rgb


def cuvol_backward(sc, origins, dirs, opts, grad_out, color_cache, grads=None):
    r = _rays(origins, dirs)
    grads = grads if grads is not None else zero_grads(sc)
    grad_out = np.ascontiguousarray(grad_out, np.float32)
    color_cache = np.ascontiguousarray(color_cache, np.float32)
    g, o, gg = _grid(sc), make_opts(opts), _grads(grads)
    _check(lib().oracle_volume_render_cuvol_backward(
        C.byref(g), C.byref(r), C.byref(o), _f(grad_out), _f(color_cache), C.byref(gg)), "cuvol backward")
    return grads


def cuvol_fused(sc, origins, dirs, opts, rgb_gt, beta_loss=0.0, sparsity_loss=0.0, grads=None):
    r = _rays(origins, dirs)
    grads = grads if grads is not None else zero_grads(sc)
    rgb_gt = np.ascontiguousarray(rgb_gt, np.float32)
    rgb = np.zeros((r.n_rays, 3), np.float32)
    g, o, gg = _grid(sc), make_opts(opts), _grads(grads)
    _check(lib().oracle_volume_render_cuvol_fused(
        C.byref(g), C.byref(r), C.byref(o), _f(rgb_gt), C.c_float(beta_loss), C.c_float(sparsity_loss),
        _f(rgb), C.byref(gg)), "cuvol fused")
    return rgb, grads


def tv_value(links, data, start_dim, end_dim, ignore_edge):
    """tv_kernel + its launcher (loss_kernel.cu:73-117, 1214-1247): mean over the (X-1)(Y-1)(Z-1) cells of
    the per-channel sqrt(1e-5 + |scaled forward differences|^2), summed over channels; the sum itself in
    float64 (the reference's float atomics have no fixed order).  PARITY UNPINNED: no golden vector -- the
    reference's kernel needs CUDA and its tests hold no expected value."""
    links = np.asarray(links)
    data = np.asarray(data, np.float32)
    X, Y, Z = links.shape
    l000 = links[:-1, :-1, :-1]
    sc = [np.float32(n) * np.float32(1.0 / 256.0) for n in (X, Y, Z)]
    nb = [links[1:, :-1, :-1], links[:-1, 1:, :-1], links[:-1, :-1, 1:]]
    keep = (l000 != 0) if ignore_edge else np.ones(l000.shape, bool)
    total = 0.0
    for idx in range(start_dim, end_dim):
        col = data[:, idx]
        v0 = np.where(l000 >= 0, col[np.maximum(l000, 0)], np.float32(0))
        null = v0 if ignore_edge else np.zeros_like(v0)
        acc = np.full(v0.shape, np.float32(1e-5), np.float32)
        for k in range(3):
            vk = np.where(nb[k] >= 0, col[np.maximum(nb[k], 0)], null).astype(np.float32)
            d = ((vk - v0) * sc[k]).astype(np.float32)
            acc = (acc + d * d).astype(np.float32)
        total += float(np.sqrt(acc, dtype=np.float32)[keep].astype(np.float64).sum())
    return total / float(l000.size)


def weight_render(data, links, c2w, fx, fy, cx, cy, width, height, step_size, stop_thresh, last_sample_opaque,
                  offset, scaling, out=None):
    """grid_weight_render (links=None, data = dense [X,Y,Z] volume) / sparse_grid_weight_render
    (data = density rows): max into `out` [X,Y,Z] (zeros when not given).  Parity unpinned (see the .h)."""
    data = np.ascontiguousarray(data, np.float32)
    size = np.asarray(data.shape if links is None else links.shape, np.int32)
    out = np.zeros(tuple(size), np.float32) if out is None else out
    lk = None if links is None else np.ascontiguousarray(links, np.int32)
    c2w = np.ascontiguousarray(np.asarray(c2w, np.float32)[:3, :4])
    off, scl = np.ascontiguousarray(offset, np.float32), np.ascontiguousarray(scaling, np.float32)
    _check(lib().oracle_weight_render(
        _f(data), None if lk is None else _i(lk), _i(size), _f(c2w), C.c_float(fx), C.c_float(fy), C.c_float(cx),
        C.c_float(cy), C.c_int32(width), C.c_int32(height), C.c_float(step_size), C.c_float(stop_thresh),
        C.c_int32(int(bool(last_sample_opaque))), _f(off), _f(scl), _f(out)), "weight_render")
    return out


def cubic_extract_iso_pts(links, level_data, mask_data, cell_ids, n_sample, density_thresh):
    """cubic_extract_iso_pts (svox2_kernel.cu:248-372): [n_cells, 3 n_sample^2, 3] grid-space points, zeros where
    no root passes.  Parity unpinned as a whole (see the .h)."""
    lk = np.ascontiguousarray(links, np.int32)
    size = np.asarray(lk.shape, np.int32)
    lv = np.ascontiguousarray(level_data, np.float32).reshape(-1)
    mk = np.ascontiguousarray(mask_data, np.float32).reshape(-1)
    ids = np.ascontiguousarray(cell_ids, np.int32)
    out = np.zeros((ids.shape[0], 3 * n_sample * n_sample, 3), np.float32)
    _check(lib().oracle_cubic_extract_iso_pts(_i(lk), _i(size), _f(lv), _f(mk), _i(ids), C.c_int64(ids.shape[0]),
                                              C.c_int32(n_sample), C.c_float(density_thresh), _f(out)),
           "cubic_extract_iso_pts")
    return out


CUVOL_EVAL_MODES = {"expected_term": 0, "mode_term": 1, "med_term": 2, "sigma_thresh": 3}


def cuvol_eval(sc, origins, dirs, opts, what, param=0.0, max_sample=1):
    """Depth renders of the cuvol march: what in CUVOL_EVAL_MODES; med_term returns (depths, sigmas)."""
    r = _rays(origins, dirs)
    mode = CUVOL_EVAL_MODES[what]
    shape = (r.n_rays, max_sample) if mode == 2 else (r.n_rays,)
    out0, out1 = np.zeros(shape, np.float32), np.zeros(shape if mode == 2 else (1,), np.float32)
    g, o = _grid(sc), make_opts(opts)
    _check(lib().oracle_render_eval_cuvol(C.byref(g), C.byref(r), C.byref(o), C.c_int32(mode), C.c_float(param),
                                          C.c_int32(max_sample), _f(out0), _f(out1)), "cuvol eval")
    return (out0, out1) if mode == 2 else out0


def set_legacy_f64_surface_grad(on):
    lib().oracle_set_legacy_f64_surface_grad(int(bool(on)))


def test_cubic_root_grad(fs, st_ids, grad_in, use_vieta=True):
    fs = np.ascontiguousarray(fs, np.float64)
    st_ids = np.ascontiguousarray(st_ids, np.int32)
    grad_fs = np.ascontiguousarray(np.broadcast_to(np.asarray(grad_in, np.float32).reshape(-1, 1),
                                                   (fs.shape[0], 4))).copy()
    st = np.zeros((fs.shape[0], 3), np.float32)
    _check(lib().oracle_test_cubic_root_grad(fs.ctypes.data_as(C.POINTER(C.c_double)), _i(st_ids),
                                             _f(grad_fs), _f(st), C.c_int64(fs.shape[0]),
                                             C.c_int32(int(use_vieta))), "test_cubic_root_grad")
    return grad_fs, st


def eval_sh(basis_dim, dirs):
    dirs = np.ascontiguousarray(dirs, np.float32)
    out = np.zeros((dirs.shape[0], basis_dim), np.float32)
    lib().oracle_eval_sh(C.c_int32(basis_dim), _f(dirs), C.c_int64(dirs.shape[0]), _f(out))
    return out


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(n):
    lib().oracle_set_num_threads(C.c_int(int(n)))


# ---- per-iteration regularisers (asurf_oracle_reg.cpp) ----
def _links_size(links):
    links = np.ascontiguousarray(links, np.int32)
    return links, (C.c_int32 * 3)(*links.shape)


def tv_grad_sparse(links, data, rand_cells, scale, start_dim=0, end_dim=None, ignore_edge=False,
                   ignore_last_z=False, mask=None, grad=None, surf=False, density=None, edge_value=0.0,
                   alpha_dependency=False):
    """tv_grad_sparse (surf=False) / surf_tv_grad_sparse (surf=True); returns (grad, mask)."""
    links, size = _links_size(links)
    data = np.ascontiguousarray(data, np.float32)
    rc = np.ascontiguousarray(rand_cells, np.int32)
    end_dim = data.shape[1] if end_dim is None else end_dim
    grad = np.zeros_like(data) if grad is None else grad
    mask = np.zeros(data.shape[0], np.uint8) if mask is None else mask
    dens = np.ascontiguousarray(density, np.float32) if density is not None else None
    f = lib().oracle_tv_grad_sparse
    f.argtypes = [c_i, c_i, c_f, C.c_int32, c_f, C.c_int32, c_i, C.c_int64, c_u8, C.c_int32, C.c_int32,
                  C.c_float, C.c_int32, C.c_float, C.c_int32, C.c_int32, C.c_int32, c_f]
    _check(f(_i(links), size, _f(data), data.shape[1], _f(dens) if dens is not None else None,
             dens.shape[1] if dens is not None else 0, _i(rc), rc.shape[0], _u8(mask), start_dim, end_dim,
             scale, int(ignore_edge), edge_value, int(ignore_last_z), int(surf), int(alpha_dependency),
             _f(grad)), "tv_grad_sparse")
    return grad, mask


def surface_normal_grad_sparse(links, surface, rand_cells, scale, lv_set=0.0, con_check=True,
                               ignore_empty=False, use_l1=False, mask=None, grad=None):
    links, size = _links_size(links)
    surface = np.ascontiguousarray(surface, np.float32)
    rc = np.ascontiguousarray(rand_cells, np.int32)
    grad = np.zeros_like(surface) if grad is None else grad
    mask = np.zeros(surface.shape[0], np.uint8) if mask is None else mask
    f = lib().oracle_surface_normal_grad_sparse
    f.argtypes = [c_i, c_i, c_f, c_i, C.c_int64, c_u8, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_int32, c_f]
    _check(f(_i(links), size, _f(surface), _i(rc), rc.shape[0], _u8(mask), lv_set, scale, int(con_check),
             int(ignore_empty), int(use_l1), _f(grad)), "surface_normal_grad_sparse")
    return grad, mask


def alpha_surf_sparsify_grad_sparse(links, alpha, surf, rand_cells, scale_alpha, scale_surf, surf_decrease,
                                    surf_thresh, alpha_bound, surf_bound):
    links, size = _links_size(links)
    alpha = np.ascontiguousarray(alpha, np.float32)
    surf = np.ascontiguousarray(surf, np.float32)
    rc = np.ascontiguousarray(rand_cells, np.int32)
    ga, gs = np.zeros_like(alpha), np.zeros_like(surf)
    mask = np.zeros(alpha.shape[0], np.uint8)
    f = lib().oracle_alpha_surf_sparsify_grad_sparse
    f.argtypes = [c_i, c_i, c_f, c_f, c_i, C.c_int64, c_u8, C.c_float, C.c_float, C.c_int32, C.c_float,
                  C.c_float, C.c_float, c_f, c_f]
    _check(f(_i(links), size, _f(alpha), _f(surf), _i(rc), rc.shape[0], _u8(mask), scale_alpha, scale_surf,
             int(surf_decrease), surf_thresh, alpha_bound, surf_bound, _f(ga), _f(gs)), "alpha_surf_sparsify")
    return ga, gs, mask


def surface_eikonal_grad_sparse(links, surface, rand_cells, scale, h, with_mask=False):
    links, size = _links_size(links)
    surface = np.ascontiguousarray(surface, np.float32)
    rc = np.ascontiguousarray(rand_cells, np.int32)
    grad = np.zeros_like(surface)
    mask = np.zeros(surface.shape[0], np.uint8)
    loss = C.c_double(0)
    f = lib().oracle_surface_eikonal_grad_sparse
    f.argtypes = [c_i, c_i, c_f, c_i, C.c_int64, c_u8, C.c_float, C.c_float, c_f, C.POINTER(C.c_double)]
    _check(f(_i(links), size, _f(surface), _i(rc), rc.shape[0], _u8(mask) if with_mask else None, scale, h,
             _f(grad), C.byref(loss)), "surface_eikonal_grad_sparse")
    return grad, float(loss.value), mask


EVAL_MODES = dict(expected=0, mode=1, sigma_thresh=2, alpha=3, normal=4, extract_pts=5)


def render_eval(sc, origins, dirs, opts, what, param=0.0, max_sample=1):
    """Evaluation renders (surf_trav.cu:3944-4081): returns out[Q] (out[Q,3] for 'normal';
    (depths, alphas)[Q,max_sample] for 'extract_pts')."""
    r = _rays(origins, dirs)
    q = r.n_rays
    mode = EVAL_MODES[what]
    stride = 3 if mode == 4 else (max_sample if mode == 5 else 1)
    out0 = np.zeros((q, stride), np.float32)
    out1 = np.zeros((q, stride), np.float32) if mode == 5 else None
    g, o = _grid(sc), make_opts(opts)
    f = lib().oracle_render_eval
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_int32, c_f, c_f]
    _check(f(C.byref(g), C.byref(r), C.byref(o), mode, param, max_sample, _f(out0), _f(out1)), "render_eval")
    if mode == 5:
        return out0, out1
    return out0 if mode == 4 else out0[:, 0]
