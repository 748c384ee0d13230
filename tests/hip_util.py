"""Drive the HIP path (through alphasurf_amd.csrc_api -> C ABI) from numpy scene dicts."""
import numpy as np
import torch

from alphasurf_amd import csrc_api as _C

OPT_DEFAULTS = dict(
    background_brightness=1.0, step_size=0.5, sigma_thresh=1e-10, stop_thresh=1e-7, near_clip=0.0,
    use_spheric_clip=False, last_sample_opaque=False, surf_fake_sample=False,
    surf_fake_sample_min_vox_len=0.1, limited_fake_sample=False, no_surf_grad_from_sh=False,
    alpha_activation_type=1, fake_sample_l_dist=True, fake_sample_normalize_surf=True,
    only_outward_intersect=False, truncated_vol_render=False, trunc_vol_weight_min=0.0)

FUSED_ORDER = [
    ("beta_loss", 0.0), ("sparsity_loss", 0.0), ("fused_surf_norm_reg_scale", 0.0),
    ("fused_surf_norm_reg_con_check", True), ("fused_surf_norm_reg_ignore_empty", False),
    ("lambda_l2", 1.0), ("lambda_l1", 0.0), ("lambda_l_dist", 0.0), ("lambda_l_entropy", 0.0),
    ("no_norm_weight_l_entropy", False), ("lambda_l_dist_a", 0.0), ("lambda_l_entropy_a", 0.0),
    ("lambda_l_samp_dist", 0.0), ("lambda_l_di", 0.0), ("l_di_alpha_thresh", 0.0),
    ("surf_sparse_alpha_thresh", 0.0), ("lambda_inplace_surf_sparse", 0.0),
    ("lambda_inwards_norm_loss", 0.0), ("lambda_conv_mode_samp", 0.0), ("l_dist_max_sample", 64)]


def dev():
    return torch.device("cuda:0")


_links_keep = {}


def grid_spec(sc, links_epoch=0):
    d = dev()
    g = _C.SparseGridSpec()
    if links_epoch:
        # the promise of links_epoch is about one `links` tensor that stays where it is: keep it between calls
        key = (id(sc["links"]), links_epoch)
        if key not in _links_keep:
            _links_keep.clear()
            _links_keep[key] = (sc["links"], torch.from_numpy(sc["links"]).to(d))
        g.links_epoch = links_epoch
    g.density_data = torch.from_numpy(sc["density"]).to(d)
    g.surface_data = torch.from_numpy(sc["surface"]).to(d)
    g.sh_data = torch.from_numpy(sc["sh"]).to(d)
    g.links = _links_keep[(id(sc["links"]), links_epoch)][1] if links_epoch else torch.from_numpy(sc["links"]).to(d)
    g.level_set_data = torch.from_numpy(sc["level_set"]).to(d)
    g._offset = torch.from_numpy(np.asarray(sc["offset"], np.float32))
    g._scaling = torch.from_numpy(np.asarray(sc["scaling"], np.float32))
    g.basis_dim = sc["basis_dim"]
    g.basis_type = 1
    g.surface_type = sc.get("surface_type", 102)
    g.fake_sample_std = sc["fake_sample_std"]
    g.truncated_vol_render_a = sc["truncated_vol_render_a"]
    return g


def rays_spec(origins, dirs, image_width=0):
    d = dev()
    r = _C.RaysSpec()
    r.image_width = image_width
    r.origins = torch.from_numpy(np.ascontiguousarray(origins, np.float32)).to(d)
    r.dirs = torch.from_numpy(np.ascontiguousarray(dirs, np.float32)).to(d)
    r.masks = torch.ones(r.origins.shape[0], dtype=torch.bool, device=d)
    return r


def opt_spec(opts):
    o = _C.RenderOptions()
    full = dict(OPT_DEFAULTS)
    opts = dict(opts or {})
    opts.pop("backend", None)
    full.update(opts)
    for k, v in full.items():
        assert hasattr(o, k), k
        setattr(o, k, v)
    return o


def zero_grads(g, with_std=True):
    gr = _C.GridOutputGrads()
    gr.grad_density_out = torch.zeros_like(g.density_data)
    gr.grad_sh_out = torch.zeros_like(g.sh_data)
    gr.grad_surface_out = torch.zeros_like(g.surface_data)
    if with_std:
        gr.grad_fake_sample_std_out = torch.zeros(1, 1, device=dev())
    gr.mask_out = torch.zeros(g.density_data.shape[0], dtype=torch.bool, device=dev())
    gr.mask_sh_out = torch.zeros_like(gr.mask_out)   # (asurf_grads::mask_sh_out: must come back equal, grads_np)
    return gr


def grads_np(gr):
    if getattr(gr, "mask_sh_out", None) is not None:
        assert bool((gr.mask_sh_out == gr.mask_out).all()), "the second mask differs from the first"
    out = dict(density=gr.grad_density_out.cpu().numpy(), sh=gr.grad_sh_out.cpu().numpy(),
               surface=gr.grad_surface_out.cpu().numpy(), mask=gr.mask_out.cpu().numpy())
    if gr.grad_fake_sample_std_out is not None:
        out["fake_sample_std"] = gr.grad_fake_sample_std_out.cpu().numpy().reshape(-1)
    return out


def render_forward(sc, origins, dirs, opts=None, counters=False):
    g, r, o = grid_spec(sc), rays_spec(origins, dirs), opt_spec(opts)
    cnt = torch.zeros(4, dtype=torch.int64, device=dev()) if counters else None
    rgb = _C.volume_render_surf_trav(g, r, o, counters=cnt)
    torch.cuda.synchronize()
    return (rgb.cpu().numpy(), cnt.cpu().numpy()) if counters else rgb.cpu().numpy()


def render_backward(sc, origins, dirs, opts, grad_out, color_cache):
    g, r, o = grid_spec(sc), rays_spec(origins, dirs), opt_spec(opts)
    gr = zero_grads(g)
    _C.volume_render_surf_trav_backward(
        g, r, o, torch.from_numpy(np.ascontiguousarray(grad_out, np.float32)).to(dev()),
        torch.from_numpy(np.ascontiguousarray(color_cache, np.float32)).to(dev()), gr)
    torch.cuda.synchronize()
    return grads_np(gr)


def render_fused(sc, origins, dirs, opts, rgb_gt, fused=None, global_n_rays=0, counters=False, links_epoch=0,
                 rec_depth=None):
    g, r, o = grid_spec(sc, links_epoch), rays_spec(origins, dirs), opt_spec(opts)
    gr = zero_grads(g)
    fused = dict(fused or {})
    args = [fused.pop(k, dflt) for k, dflt in FUSED_ORDER]
    assert not fused, fused
    gt = torch.from_numpy(np.ascontiguousarray(rgb_gt, np.float32)).to(dev())
    # (NaN in front: the call owes every ray its colour -- the mirror hands it an uninitialised buffer, grid.py)
    rgb = torch.full_like(gt, float("nan"))
    cnt = torch.zeros(4, dtype=torch.int64, device=dev()) if counters else None
    _C.volume_render_surf_trav_fused(g, r, o, gt, *args, rgb, gr, global_n_rays=global_n_rays,
                                     counters=cnt, rec_depth=rec_depth)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(rgb).any()), "a ray was left without a colour"
    out = (rgb.cpu().numpy(), grads_np(gr))
    return out + (cnt.cpu().numpy(),) if counters else out


# ---- legacy `surface` backend ----
def ray_vox_spec(ray_vox):
    d = dev()
    rv = _C.RayVoxIntersecSpec()
    rv.voxel_ls = torch.from_numpy(np.ascontiguousarray(ray_vox[0], np.int32).reshape(-1, 3)).to(d)
    rv.vox_start_i = torch.from_numpy(np.ascontiguousarray(ray_vox[1], np.int32)).to(d)
    rv.vox_num = torch.from_numpy(np.ascontiguousarray(ray_vox[2], np.int32)).to(d)
    return rv


def voxel_lists(sc, origins, dirs, opts=None, skip_empty=False):
    g, r, o = grid_spec(sc), rays_spec(origins, dirs), opt_spec(opts)
    vox, start, num = _C.ray_voxel_lists(g, r, o, skip_empty=skip_empty)
    torch.cuda.synchronize()
    return vox.cpu().numpy(), start.cpu().numpy(), num.cpu().numpy()


def surface_forward(sc, origins, dirs, ray_vox, opts=None):
    g, r, o, rv = grid_spec(sc), rays_spec(origins, dirs), opt_spec(opts), ray_vox_spec(ray_vox)
    rgb = _C.volume_render_surface(g, r, rv, o)
    torch.cuda.synchronize()
    return rgb.cpu().numpy()


def surface_backward(sc, origins, dirs, ray_vox, opts, grad_out, color_cache):
    g, r, o, rv = grid_spec(sc), rays_spec(origins, dirs), opt_spec(opts), ray_vox_spec(ray_vox)
    gr = zero_grads(g, with_std=False)
    _C.volume_render_surface_backward(
        g, r, rv, o, torch.from_numpy(np.ascontiguousarray(grad_out, np.float32)).to(dev()),
        torch.from_numpy(np.ascontiguousarray(color_cache, np.float32)).to(dev()), gr)
    torch.cuda.synchronize()
    return grads_np(gr)


def surface_fused(sc, origins, dirs, ray_vox, opts, rgb_gt, beta_loss=0.0, sparsity_loss=0.0):
    g, r, o, rv = grid_spec(sc), rays_spec(origins, dirs), opt_spec(opts), ray_vox_spec(ray_vox)
    gr = zero_grads(g, with_std=False)
    gt = torch.from_numpy(np.ascontiguousarray(rgb_gt, np.float32)).to(dev())
    rgb = torch.zeros_like(gt)
    _C.volume_render_surface_fused(g, r, rv, o, gt, beta_loss, sparsity_loss, rgb, gr)
    torch.cuda.synchronize()
    return rgb.cpu().numpy(), grads_np(gr)


# ---- the Plenoxels `cuvol` renderer ----
def _cuvol_grads(g):
    gr = _C.GridOutputGrads()
    gr.grad_density_out = torch.zeros_like(g.density_data)
    gr.grad_sh_out = torch.zeros_like(g.sh_data)
    gr.mask_out = torch.zeros(g.density_data.shape[0], dtype=torch.bool, device=dev())
    gr.mask_sh_out = torch.zeros_like(gr.mask_out)   # (asurf_grads::mask_sh_out: must come back equal, grads_np)
    return gr


def _cuvol_grads_np(gr):
    return dict(density=gr.grad_density_out.cpu().numpy(), sh=gr.grad_sh_out.cpu().numpy(),
                mask=gr.mask_out.cpu().numpy())


def cuvol_forward(sc, origins, dirs, opts=None):
    g, r, o = grid_spec(sc), rays_spec(origins, dirs), opt_spec(opts)
    rgb = _C.volume_render_cuvol(g, r, o)
    torch.cuda.synchronize()
    return rgb.cpu().numpy()


def cuvol_backward(sc, origins, dirs, opts, grad_out, color_cache):
    g, r, o = grid_spec(sc), rays_spec(origins, dirs), opt_spec(opts)
    gr = _cuvol_grads(g)
    _C.volume_render_cuvol_backward(
        g, r, o, torch.from_numpy(np.ascontiguousarray(grad_out, np.float32)).to(dev()),
        torch.from_numpy(np.ascontiguousarray(color_cache, np.float32)).to(dev()), gr)
    torch.cuda.synchronize()
    return _cuvol_grads_np(gr)


def cuvol_fused(sc, origins, dirs, opts, rgb_gt, beta_loss=0.0, sparsity_loss=0.0, image_width=0):
    g, r, o = grid_spec(sc), rays_spec(origins, dirs, image_width), opt_spec(opts)
    gr = _cuvol_grads(g)
    gt = torch.from_numpy(np.ascontiguousarray(rgb_gt, np.float32)).to(dev())
    rgb = torch.zeros_like(gt)
    _C.volume_render_cuvol_fused(g, r, o, gt, beta_loss, sparsity_loss, rgb, gr)
    torch.cuda.synchronize()
    return rgb.cpu().numpy(), _cuvol_grads_np(gr)


def cuvol_eval(sc, origins, dirs, opts, what, param=0.0, max_sample=1):
    g, r, o = grid_spec(sc), rays_spec(origins, dirs), opt_spec(opts)
    fn = {"expected_term": _C.volume_render_expected_term, "mode_term": _C.volume_render_mode_term,
          "sigma_thresh": _C.volume_render_sigma_thresh}
    if what == "med_term":
        a, b = _C.volume_render_med_term(g, r, o, max_sample)
        torch.cuda.synchronize()
        return a.cpu().numpy(), b.cpu().numpy()
    out = fn[what](g, r, o, param)
    torch.cuda.synchronize()
    return out.cpu().numpy()


# ---- evaluation renders ----
def render_eval(sc, origins, dirs, opts, what, param=0.0, max_sample=1):
    g, r, o = grid_spec(sc), rays_spec(origins, dirs), opt_spec(opts)
    if what == "expected":
        out = _C.volume_render_expected_term_surf_trav(g, r, o)
    elif what == "mode":
        out = _C.volume_render_mode_term_surf_trav(g, r, o, param)
    elif what == "sigma_thresh":
        out = _C.volume_render_sigma_thresh_surf_trav(g, r, o, param)
    elif what == "alpha":
        out = _C.volume_render_alpha_surf_trav(g, r, o, param)
    elif what == "normal":
        out = _C.render_normal_surf_trav(g, r, o)
    else:
        d, a = _C.extract_pts_surf_trav(g, r, o, max_sample, param)
        torch.cuda.synchronize()
        return d.cpu().numpy(), a.cpu().numpy()
    torch.cuda.synchronize()
    return out.cpu().numpy()
