"""ctypes binding of libasurf_hip.so (the C ABI declared in include/asurf.h).

This is the only way compute is reached: there is NO CPU fallback.  If the shared
library is missing or fails to load, every entry point raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# ASURF_LIB: development override to A/B another build of the same sources
LIB_PATH = os.environ.get("ASURF_LIB") or os.path.join(CSRC, "libasurf_hip.so")

c_f = C.POINTER(C.c_float)
c_i = C.POINTER(C.c_int32)
c_u8 = C.POINTER(C.c_uint8)


class AsurfGrid(C.Structure):
    _fields_ = [
        ("links", C.c_void_p), ("density_data", C.c_void_p), ("surface_data", C.c_void_p),
        ("sh_data", C.c_void_p), ("level_set_data", C.c_void_p), ("capacity", C.c_int64),
        ("size", C.c_int32 * 3), ("level_set_num", C.c_int32), ("basis_dim", C.c_int32),
        ("basis_type", C.c_int32), ("surface_type", C.c_int32), ("offset", C.c_float * 3),
        ("scaling", C.c_float * 3), ("fake_sample_std", C.c_float),
        ("truncated_vol_render_a", C.c_float), ("background_nlayers", C.c_int32),
        ("links_epoch", C.c_uint64),
    ]


class AsurfRays(C.Structure):
    _fields_ = [("origins", C.c_void_p), ("dirs", C.c_void_p), ("masks", C.c_void_p),
                ("n_rays", C.c_int64), ("image_width", C.c_int32), ("pad_", C.c_int32)]


class AsurfRayVox(C.Structure):
    _fields_ = [("voxel_ls", C.c_void_p), ("vox_start_i", C.c_void_p), ("vox_num", C.c_void_p),
                ("n_vox", C.c_int64)]


class AsurfCamera(C.Structure):
    _fields_ = [("c2w", C.c_float * 12), ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float),
                ("cy", C.c_float), ("width", C.c_int32), ("height", C.c_int32)]


RENDER_OPTION_FIELDS = [
    ("background_brightness", C.c_float), ("step_size", C.c_float), ("sigma_thresh", C.c_float),
    ("stop_thresh", C.c_float), ("near_clip", C.c_float), ("use_spheric_clip", C.c_int32),
    ("last_sample_opaque", C.c_int32), ("surf_fake_sample", C.c_int32),
    ("surf_fake_sample_min_vox_len", C.c_float), ("limited_fake_sample", C.c_int32),
    ("no_surf_grad_from_sh", C.c_int32), ("alpha_activation_type", C.c_int32),
    ("fake_sample_l_dist", C.c_int32), ("fake_sample_normalize_surf", C.c_int32),
    ("only_outward_intersect", C.c_int32), ("truncated_vol_render", C.c_int32),
    ("trunc_vol_weight_min", C.c_float),
]


class AsurfRenderOptions(C.Structure):
    _fields_ = RENDER_OPTION_FIELDS


class AsurfGrads(C.Structure):
    _fields_ = [("grad_density_out", C.c_void_p), ("grad_sh_out", C.c_void_p),
                ("grad_surface_out", C.c_void_p), ("grad_fake_sample_std_out", C.c_void_p),
                ("mask_out", C.c_void_p), ("mask_sh_out", C.c_void_p)]


FUSED_FIELDS = [
    ("rgb_gt", C.c_void_p), ("beta_loss", C.c_float), ("sparsity_loss", C.c_float),
    ("fused_surf_norm_reg_scale", C.c_float), ("fused_surf_norm_reg_con_check", C.c_int32),
    ("fused_surf_norm_reg_ignore_empty", C.c_int32), ("lambda_l2", C.c_float),
    ("lambda_l1", C.c_float), ("lambda_l_dist", C.c_float), ("lambda_l_entropy", C.c_float),
    ("no_norm_weight_l_entropy", C.c_int32), ("lambda_l_dist_a", C.c_float),
    ("lambda_l_entropy_a", C.c_float), ("lambda_l_samp_dist", C.c_float),
    ("lambda_l_di", C.c_float), ("l_di_alpha_thresh", C.c_float),
    ("surf_sparse_alpha_thresh", C.c_float), ("lambda_inplace_surf_sparse", C.c_float),
    ("lambda_inwards_norm_loss", C.c_float), ("lambda_conv_mode_samp", C.c_float),
    ("l_dist_max_sample", C.c_int32), ("global_n_rays", C.c_int64),
    ("order_rays", C.c_int32), ("image_width", C.c_int32),
]


class AsurfFusedParams(C.Structure):
    _fields_ = FUSED_FIELDS


N_COUNTERS = 4

EXPORTS = [
    "asurf_abi_version", "asurf_last_error", "asurf_workspace_bytes", "asurf_workspace_bytes_rec",
    "asurf_volume_render_surf_trav", "asurf_volume_render_surf_trav_backward",
    "asurf_volume_render_surf_trav_fused", "asurf_volume_render_surface",
    "asurf_volume_render_surface_backward", "asurf_volume_render_surface_fused",
    "asurf_test_cubic_root_grad", "asurf_profile_enable", "asurf_profile_read", "asurf_debug_read_timeline",
    "asurf_config_set",
    "asurf_profile_read_spans",
    "asurf_rmsprop_step", "asurf_sgd_step", "asurf_optim_scratch_bytes", "asurf_rmsprop_step_masked",
    "asurf_sgd_step_masked", "asurf_ray_voxel_count", "asurf_ray_voxel_fill",
    "asurf_tv", "asurf_tv_grad_sparse", "asurf_surf_tv_grad_sparse", "asurf_surface_normal_grad_sparse",
    "asurf_alpha_surf_sparsify_grad_sparse", "asurf_surface_eikonal_grad_sparse",
    "asurf_render_eval_surf_trav", "asurf_dilate", "asurf_accel_dist_prop_tmp_bytes", "asurf_accel_dist_prop",
    "asurf_sample_plane", "asurf_sample_plane_backward", "asurf_cubic_extract_iso_pts",
    "asurf_volume_render_cuvol", "asurf_volume_render_cuvol_backward", "asurf_volume_render_cuvol_fused",
    "asurf_grid_weight_render", "asurf_sparse_grid_weight_render", "asurf_render_eval_cuvol",
    "asurf_ray_order_tmp_bytes", "asurf_ray_order", "asurf_test_quotient",
    "asurf_rows_list_tmp_bytes", "asurf_rows_list", "asurf_rows_pack", "asurf_rows_unpack", "asurf_rows_unpack_add",
]

_lib = None


def build(force=False, verbose=False):
    """Compile csrc/ into libasurf_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC] + (["-B"] if force else []) + ["libasurf_hip.so"]
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return LIB_PATH


def lib():
    """The loaded library; raises if the HIP extension is missing (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` or `make -C {CSRC}`.  alphasurf_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.asurf_last_error.restype = C.c_char_p
        L.asurf_workspace_bytes.restype = C.c_size_t
        L.asurf_workspace_bytes.argtypes = [C.POINTER(C.c_int32), C.c_int64, C.c_int32]
        L.asurf_workspace_bytes_rec.restype = C.c_size_t
        L.asurf_workspace_bytes_rec.argtypes = [C.POINTER(C.c_int32), C.c_int64, C.c_int32, C.c_int32]
        vp = C.c_void_p
        L.asurf_volume_render_surf_trav.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, vp, vp]
        L.asurf_volume_render_surf_trav_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, vp]
        L.asurf_volume_render_surf_trav_fused.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp, vp]
        L.asurf_volume_render_surface.argtypes = [vp, vp, vp, vp, vp, vp]
        L.asurf_volume_render_surface_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
        L.asurf_volume_render_surface_fused.argtypes = [vp, vp, vp, vp, vp, C.c_float, C.c_float, vp, vp, vp]
        L.asurf_test_cubic_root_grad.argtypes = [vp, vp, vp, vp, C.c_int64, C.c_int32, vp]
        fl = C.c_float
        L.asurf_rmsprop_step.argtypes = [vp, vp, vp, C.c_int64, C.c_int32, vp, vp, C.c_int64, fl, fl, fl, fl, fl, vp]
        L.asurf_sgd_step.argtypes = [vp, vp, C.c_int64, C.c_int32, vp, vp, C.c_int64, fl, fl, vp]
        L.asurf_optim_scratch_bytes.restype = C.c_size_t
        L.asurf_optim_scratch_bytes.argtypes = [C.c_int64]
        L.asurf_rmsprop_step_masked.argtypes = [vp, vp, vp, C.c_int64, C.c_int32, vp, vp, C.c_size_t,
                                                fl, fl, fl, fl, fl, vp]
        L.asurf_sgd_step_masked.argtypes = [vp, vp, C.c_int64, C.c_int32, vp, vp, C.c_size_t, fl, fl, vp]
        L.asurf_ray_voxel_count.argtypes = [vp, vp, vp, C.c_int32, vp, vp]
        L.asurf_ray_voxel_fill.argtypes = [vp, vp, vp, C.c_int32, vp, vp, vp]
        i32, i64 = C.c_int32, C.c_int64
        L.asurf_tv.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp]
        L.asurf_cubic_extract_iso_pts.argtypes = [vp, vp, vp, vp, vp, i64, i32, fl, vp, vp]
        L.asurf_tv_grad_sparse.argtypes = [vp, vp, vp, i32, vp, i64, vp, i32, i32, fl, i32, i32, vp, vp]
        L.asurf_surf_tv_grad_sparse.argtypes = [vp, vp, vp, i32, vp, i32, vp, i64, vp, i32, i32, fl, i32, fl,
                                                i32, i32, vp, vp]
        L.asurf_surface_normal_grad_sparse.argtypes = [vp, vp, vp, vp, i64, vp, fl, fl, i32, i32, i32, vp, vp]
        L.asurf_alpha_surf_sparsify_grad_sparse.argtypes = [vp, vp, vp, vp, vp, i64, vp, fl, fl, i32, fl, fl,
                                                            fl, vp, vp, vp]
        L.asurf_surface_eikonal_grad_sparse.argtypes = [vp, vp, vp, vp, i64, vp, fl, fl, vp, vp, vp]
        L.asurf_render_eval_surf_trav.argtypes = [vp, vp, vp, i32, fl, i32, vp, vp, vp, C.c_size_t, vp]
        L.asurf_dilate.argtypes = [vp, vp, vp, vp]
        L.asurf_accel_dist_prop_tmp_bytes.restype = C.c_size_t
        L.asurf_accel_dist_prop_tmp_bytes.argtypes = [vp]
        L.asurf_accel_dist_prop.argtypes = [vp, vp, vp, C.c_size_t, vp]
        L.asurf_sample_plane.argtypes = [vp, vp, vp, i32, fl, vp, i64, vp, vp, vp, vp]
        L.asurf_sample_plane_backward.argtypes = [vp, vp, i32, vp, i64, vp, vp, vp, vp, vp]
        L.asurf_volume_render_cuvol.argtypes = [vp, vp, vp, vp, vp, C.c_size_t, vp]
        L.asurf_volume_render_cuvol_backward.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
        L.asurf_volume_render_cuvol_fused.argtypes = [vp, vp, vp, vp, fl, fl, vp, vp, vp, C.c_size_t, vp]
        L.asurf_grid_weight_render.argtypes = [vp, vp, vp, fl, fl, C.c_int32, vp, vp, vp, vp]
        L.asurf_sparse_grid_weight_render.argtypes = [vp, vp, vp, vp, fl, fl, vp, vp, vp, vp]
        L.asurf_render_eval_cuvol.argtypes = [vp, vp, vp, C.c_int32, fl, C.c_int32, vp, vp, vp, C.c_size_t, vp]
        L.asurf_ray_order_tmp_bytes.argtypes = [C.c_int64]
        L.asurf_ray_order_tmp_bytes.restype = C.c_size_t
        L.asurf_ray_order.argtypes = [vp, vp, C.c_int64, vp, vp, C.c_size_t, vp]
        L.asurf_test_quotient.argtypes = [vp, vp, C.c_int64, vp, vp, vp]
        L.asurf_config_set.argtypes = [C.c_char_p, C.c_char_p]
        L.asurf_rows_list_tmp_bytes.restype = C.c_size_t
        L.asurf_rows_list_tmp_bytes.argtypes = [C.c_int64]
        L.asurf_rows_list.argtypes = [vp, C.c_int64, vp, vp, vp, C.c_size_t, vp]
        L.asurf_rows_pack.argtypes = [vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp]
        L.asurf_rows_unpack.argtypes = [vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, vp]
        L.asurf_rows_unpack_add.argtypes = [vp, C.c_int64, C.c_int32, vp, vp, vp, vp, vp, vp]
        if L.asurf_abi_version() != 4:
            raise RuntimeError("libasurf_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc):
    """Non-zero return -> RuntimeError carrying the library's message (the reference raises
    RuntimeError through TORCH_CHECK for the same class of misuse)."""
    if rc != 0:
        raise RuntimeError(f"asurf: {lib().asurf_last_error().decode()} (code {rc})")


def config_set(name, value=None):
    """One of the library's internal switches (include/asurf.h, asurf_config_set): `value` None = default.  The
    library reads the ASURF_* environment once, when its first entry point runs; later changes go through here."""
    check(lib().asurf_config_set(name.encode(), None if value is None else str(value).encode()))


def profile_enable(on=True):
    check(lib().asurf_profile_enable(int(bool(on))))


def profile_read():
    """(masks_ms, forward_ms, backward_ms) of the last call, HIP-event timed on its stream."""
    ms = (C.c_float * 3)()
    check(lib().asurf_profile_read(ms, 3))
    return float(ms[0]), float(ms[1]), float(ms[2])


def profile_read_spans():
    """[(kernel name, ms)] of the last call, in launch order."""
    names = C.create_string_buffer(1024)
    ms = (C.c_float * 32)()
    L = lib()
    L.asurf_profile_read_spans.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_float), C.c_int]
    n = L.asurf_profile_read_spans(names, 1024, ms, 32)
    if n < 0:
        check(1)
    return list(zip(names.value.decode().split(","), [float(ms[i]) for i in range(n)]))


def workspace_bytes(size, n_rays, l_dist_max_sample, rec_depth=None):
    """Workspace size; `rec_depth`: sample records a ray for the fused call's backward (None: the library's 32)."""
    arr = (C.c_int32 * 3)(*[int(s) for s in size])
    if rec_depth is None:
        return int(lib().asurf_workspace_bytes(arr, int(n_rays), int(l_dist_max_sample)))
    return int(lib().asurf_workspace_bytes_rec(arr, int(n_rays), int(l_dist_max_sample), int(rec_depth)))
