"""Host-side mirror of the slice of the reference's `svox2` Python API that sits on the hot path
(tier P0 + the cheap P1 pieces): `SparseGrid`, `Rays`, `Camera`,
`RenderOptions`, the autograd bridge and the fused training call, with the reference's names,
keyword arguments and side effects, so that `opt/opt.py` can import this package as `svox2`.

Own code; reference locations are cited for parity checking (paths relative to the reference
checkout, `svox2/svox2.py` unless stated).  Everything that leaves the hot path raises
NotImplementedError naming the SURVEY.md section 8(f) row it belongs to.
"""
import itertools
import os
from dataclasses import dataclass
from functools import reduce
from typing import List, Optional, Tuple, Union

import numpy as np
import torch
from torch import autograd, nn

from . import csrc_api as _C
from . import gridinit, shutils

# identities of `links` tensors as the renders see them (asurf_grid::links_epoch): one monotonic, process-wide
# counter -- every SparseGrid draws a fresh value when it is built and whenever its links change
_LINKS_EPOCHS = itertools.count(1)

# enums of data_spec.hpp:8-36 / svox2/defs.py
CONSTANTS = dict(
    BASIS_TYPE_SH=1, BASIS_TYPE_3D_TEXTURE=4, BASIS_TYPE_MLP=255,
    SURFACE_TYPE_NONE=100, SURFACE_TYPE_PLANE=101, SURFACE_TYPE_SDF=102, SURFACE_TYPE_UDF=103,
    SURFACE_TYPE_UDF_ALPHA=104, SURFACE_TYPE_UDF_FAKE_SAMPLE=105, SURFACE_TYPE_VOXEL_FACE=106,
    SIGMOID_FN=0, EXP_FN=1,
)
globals().update(CONSTANTS)


def _next_row(row, what):
    raise NotImplementedError(f"{what} is outside the hot-path scope of this build (SURVEY.md section 8f, {row})")


@dataclass
class RenderOptions:
    """Rendering options (:21-105); `backend` is consumed in Python only."""
    backend: str = "cuvol"
    background_brightness: float = 1.0
    step_size: float = 0.5
    sigma_thresh: float = 1e-10
    stop_thresh: float = 1e-7
    last_sample_opaque: bool = False
    near_clip: float = 0.0
    use_spheric_clip: bool = False
    random_sigma_std: float = 1.0
    random_sigma_std_background: float = 1.0
    surf_fake_sample: bool = False
    surf_fake_sample_min_vox_len: float = 0.1
    limited_fake_sample: bool = False
    no_surf_grad_from_sh: bool = False
    alpha_activation_type: int = CONSTANTS["EXP_FN"]
    fake_sample_l_dist: bool = True
    fake_sample_normalize_surf: bool = True
    only_outward_intersect: bool = False
    truncated_vol_render: bool = False
    trunc_vol_weight_min: float = 0.0

    def _to_cpp(self, randomize: bool = False):
        opt = _C.RenderOptions()
        for name in ("background_brightness", "step_size", "sigma_thresh", "stop_thresh", "near_clip",
                     "use_spheric_clip", "last_sample_opaque", "surf_fake_sample",
                     "surf_fake_sample_min_vox_len", "limited_fake_sample", "no_surf_grad_from_sh",
                     "alpha_activation_type", "fake_sample_l_dist", "fake_sample_normalize_surf",
                     "only_outward_intersect", "truncated_vol_render", "trunc_vol_weight_min"):
            setattr(opt, name, getattr(self, name))
        return opt


@dataclass
class Rays:
    """:108-133"""
    origins: torch.Tensor
    dirs: torch.Tensor
    masks: Optional[torch.Tensor] = None
    image_width: int = 0   # extension (asurf_rays::image_width): > 0 says the rays are a row-major image of this width

    def __post_init__(self):
        if self.masks is None:
            self.masks = torch.ones_like(self.origins[..., 0]).bool()

    def _to_cpp(self):
        spec = _C.RaysSpec()
        spec.origins, spec.dirs, spec.masks = self.origins, self.dirs, self.masks
        spec.image_width = int(self.image_width or 0)
        return spec

    def __getitem__(self, key):
        return Rays(self.origins[key], self.dirs[key], self.masks[key])

    @property
    def is_cuda(self) -> bool:
        return self.origins.is_cuda and self.dirs.is_cuda and self.masks.is_cuda


@dataclass
class RayVoxIntersecs:
    """:135-156 (legacy `surface` backend)"""
    voxel_ls: torch.Tensor
    vox_start_i: torch.Tensor
    vox_num: torch.Tensor

    def _to_cpp(self):
        spec = _C.RayVoxIntersecSpec()
        spec.voxel_ls, spec.vox_start_i, spec.vox_num = self.voxel_ls, self.vox_start_i, self.vox_num
        return spec

    def __getitem__(self, key):
        return RayVoxIntersecs(self.voxel_ls[key], self.vox_start_i[key], self.vox_num[key])

    @property
    def is_cuda(self) -> bool:
        return self.voxel_ls.is_cuda and self.vox_start_i.is_cuda and self.vox_num.is_cuda


@dataclass
class Camera:
    """Pinhole camera, OpenCV convention (:159-237)."""
    c2w: torch.Tensor
    fx: float = 1111.11
    fy: Optional[float] = None
    cx: Optional[float] = None
    cy: Optional[float] = None
    width: int = 800
    height: int = 800
    ndc_coeffs: Union[Tuple[float, float], List[float]] = (-1.0, -1.0)

    @property
    def fx_val(self):
        return self.fx

    @property
    def fy_val(self):
        return self.fx if self.fy is None else self.fy

    @property
    def cx_val(self):
        return self.width * 0.5 if self.cx is None else self.cx

    @property
    def cy_val(self):
        return self.height * 0.5 if self.cy is None else self.cy

    @property
    def using_ndc(self):
        return self.ndc_coeffs[0] > 0.0

    @property
    def is_cuda(self) -> bool:
        return self.c2w.is_cuda

    def _to_cpp(self):
        spec = _C.CameraSpec()
        spec.c2w = self.c2w
        spec.fx, spec.fy, spec.cx, spec.cy = self.fx_val, self.fy_val, self.cx_val, self.cy_val
        spec.width, spec.height = self.width, self.height
        spec.ndc_coeffx, spec.ndc_coeffy = self.ndc_coeffs[0], self.ndc_coeffs[1]
        return spec

    def gen_rays(self) -> Rays:
        """Rays through the pixel centres in row-major order; directions are formed in float64
        and normalised before the rotation, like the reference (:211-237)."""
        if self.using_ndc:
            _next_row("N4", "NDC cameras")
        dev = self.c2w.device
        ys = torch.arange(self.height, dtype=torch.float64, device=dev) + 0.5
        xs = torch.arange(self.width, dtype=torch.float64, device=dev) + 0.5
        yy, xx = torch.meshgrid(ys, xs, indexing="ij")
        d = torch.stack(((xx - self.cx_val) / self.fx_val, (yy - self.cy_val) / self.fy_val,
                         torch.ones_like(xx)), dim=-1).reshape(-1, 3)
        d = d / d.norm(dim=-1, keepdim=True)
        d = (d @ self.c2w[:3, :3].double().T).float()
        o = self.c2w[None, :3, 3].expand(d.shape[0], -1).contiguous()
        return Rays(o, d.contiguous())


class _SurfTravRenderFunction(autograd.Function):
    """Autograd bridge of the non-fused path (:349-430): forward = volume_render_<backend>,
    backward = volume_render_<backend>_backward into freshly zeroed dense gradients."""

    @staticmethod
    def forward(ctx, data_density, data_surface, data_sh, data_basis, data_background,
                data_fake_sample_std, grid, rays, opt, backend):
        color = getattr(_C, f"volume_render_{backend}")(grid, rays, opt)
        ctx.save_for_backward(color)
        ctx.grid, ctx.rays, ctx.opt, ctx.backend = grid, rays, opt, backend
        return color

    @staticmethod
    def backward(ctx, grad_out):
        (color_cache,) = ctx.saved_tensors
        g = ctx.grid
        holder = _C.GridOutputGrads()
        holder.grad_density_out = torch.zeros_like(g.density_data.data)
        holder.grad_surface_out = torch.zeros_like(g.surface_data.data)
        holder.grad_sh_out = torch.zeros_like(g.sh_data.data)
        holder.grad_fake_sample_std_out = torch.zeros_like(g.density_data.data[:1])
        getattr(_C, f"volume_render_{ctx.backend}_backward")(
            g, ctx.rays, ctx.opt, grad_out.contiguous(), color_cache, holder)
        ctx.grid = ctx.rays = ctx.opt = None
        need = ctx.needs_input_grad
        return (holder.grad_density_out if need[0] else None,
                holder.grad_surface_out if need[1] else None,
                holder.grad_sh_out if need[2] else None, None, None,
                holder.grad_fake_sample_std_out if need[5] else None, None, None, None, None)


class _TotalVariationFunction(autograd.Function):
    """svox2.py:507-546: the dense TV value (_C.tv) with _C.tv_grad as its backward."""

    @staticmethod
    def forward(ctx, data, links, start_dim, end_dim, use_logalpha, logalpha_delta, ignore_edge, ndc_coeffs):
        tv = _C.tv(links, data, start_dim, end_dim, use_logalpha, logalpha_delta, ignore_edge,
                   ndc_coeffs[0], ndc_coeffs[1])
        ctx.save_for_backward(links, data)
        ctx.args = (start_dim, end_dim, use_logalpha, logalpha_delta, ignore_edge, ndc_coeffs)
        return tv

    @staticmethod
    def backward(ctx, grad_out):
        links, data = ctx.saved_tensors
        start_dim, end_dim, use_logalpha, logalpha_delta, ignore_edge, ndc_coeffs = ctx.args
        if not ctx.needs_input_grad[0]:
            return (None,) * 8
        grad_grid = torch.zeros_like(data)
        # the reference calls tv_grad with scale 1.0 and drops grad_out (svox2.py:534-538); kept
        _C.tv_grad(links, data, start_dim, end_dim, 1.0, use_logalpha, logalpha_delta, ignore_edge,
                   ndc_coeffs[0], ndc_coeffs[1], grad_grid)
        return (grad_grid,) + (None,) * 7


class _VolumeRenderFunction(autograd.Function):
    """Autograd bridge of the surface-less renderers (:280-347; here: `cuvol`)."""

    @staticmethod
    def forward(ctx, data_density, data_sh, data_basis, data_background, grid, rays, opt, backend):
        color = getattr(_C, f"volume_render_{backend}")(grid, rays, opt)
        ctx.save_for_backward(color)
        ctx.grid, ctx.rays, ctx.opt, ctx.backend = grid, rays, opt, backend
        return color

    @staticmethod
    def backward(ctx, grad_out):
        (color_cache,) = ctx.saved_tensors
        g = ctx.grid
        holder = _C.GridOutputGrads()
        holder.grad_density_out = torch.zeros_like(g.density_data.data)
        holder.grad_sh_out = torch.zeros_like(g.sh_data.data)
        getattr(_C, f"volume_render_{ctx.backend}_backward")(
            g, ctx.rays, ctx.opt, grad_out.contiguous(), color_cache, holder)
        ctx.grid = ctx.rays = ctx.opt = None
        need = ctx.needs_input_grad
        return (holder.grad_density_out if need[0] else None, holder.grad_sh_out if need[1] else None,
                None, None, None, None, None, None)


class _SurfaceRenderFunction(autograd.Function):
    """Autograd bridge of the legacy voxel-list backend (:432-511)."""

    @staticmethod
    def forward(ctx, data_density, data_surface, data_sh, data_basis, data_background, grid, rays,
                ray_vox, opt, backend):
        color = getattr(_C, f"volume_render_{backend}")(grid, rays, ray_vox, opt)
        ctx.save_for_backward(color)
        ctx.grid, ctx.rays, ctx.ray_vox, ctx.opt, ctx.backend = grid, rays, ray_vox, opt, backend
        return color

    @staticmethod
    def backward(ctx, grad_out):
        (color_cache,) = ctx.saved_tensors
        g = ctx.grid
        holder = _C.GridOutputGrads()
        holder.grad_density_out = torch.zeros_like(g.density_data.data)
        holder.grad_surface_out = torch.zeros_like(g.surface_data.data)
        holder.grad_sh_out = torch.zeros_like(g.sh_data.data)
        getattr(_C, f"volume_render_{ctx.backend}_backward")(
            g, ctx.rays, ctx.ray_vox, ctx.opt, grad_out.contiguous(), color_cache, holder)
        ctx.grid = ctx.rays = ctx.ray_vox = ctx.opt = None
        need = ctx.needs_input_grad
        return (holder.grad_density_out if need[0] else None,
                holder.grad_surface_out if need[1] else None,
                holder.grad_sh_out if need[2] else None, None, None, None, None, None, None, None)


class _SampleGridAutogradFunction(autograd.Function):
    """Autograd bridge of point sampling (:241-278)."""

    @staticmethod
    def forward(ctx, data_density, data_sh, grid, points, want_colors):
        assert not points.requires_grad, "Point gradient not supported"
        out_density, out_sh = _C.sample_grid(grid, points, want_colors)
        ctx.save_for_backward(points)
        ctx.grid, ctx.want_colors = grid, want_colors
        return out_density, out_sh

    @staticmethod
    def backward(ctx, grad_out_density, grad_out_sh):
        (points,) = ctx.saved_tensors
        g = ctx.grid
        grad_density = torch.zeros_like(g.density_data.data)
        grad_sh = torch.zeros_like(g.sh_data.data)
        _C.sample_grid_backward(g, points, grad_out_density.contiguous(),
                                grad_out_sh.contiguous() if ctx.want_colors else grad_out_sh, grad_density, grad_sh,
                                ctx.want_colors)
        return (grad_density if ctx.needs_input_grad[0] else None,
                grad_sh if ctx.needs_input_grad[1] else None, None, None, None)


# rays per call of the whole-image renders when the caller names no batch size
_IMAGE_RAYS = 1 << 20


def _image_batch(batch_size):
    """Rays per call of the whole-image renders.  The reference's `batch_size` (default 5000) guards its [Q, 64] sample
    buffers; here rays are independent and every call pays for the occupancy pyramid of the whole grid once, so a
    caller that names no batch size gets up to _IMAGE_RAYS rays a call (ASURF_IMAGE_RAYS overrides) -- at about
    1.8 KB of cached workspace per ray for these forward renders (tables, candidate pool: 2 GB per stream at 2^20
    rays).  (A FUSED call keeps sample records as well: 4.5 KB a ray with 32 records, 9.7 KB with the 128 the binding
    asks for when `surf_fake_sample` is on -- 6.2 GB at 640 k rays, csrc_api.volume_render_surf_trav_fused.)
    A batch size the caller passes is a memory bound and is kept as it is."""
    if batch_size is not None:
        return max(1, int(batch_size))
    return max(1, int(os.environ.get("ASURF_IMAGE_RAYS", _IMAGE_RAYS)))


class SparseGrid(nn.Module):
    """Sparse voxel grid with a per-vertex surface scalar (:560-990).  Only what the surf_trav
    hot path needs is built: SH basis, SDF/UDF surface types, no MSI background."""

    def __init__(self, reso: Union[int, List[int], Tuple[int, int, int]] = 128,
                 radius: Union[float, List[float]] = 1.0,
                 center: Union[float, List[float]] = [0.0, 0.0, 0.0],
                 basis_type: int = CONSTANTS["BASIS_TYPE_SH"], basis_dim: int = 9, basis_reso: int = 16,
                 use_z_order: bool = False, use_sphere_bound: bool = False, mlp_posenc_size: int = 0,
                 mlp_width: int = 16, background_nlayers: int = 0, background_reso: int = 256,
                 device: Union[torch.device, str] = "cpu",
                 surface_type: int = CONSTANTS["SURFACE_TYPE_NONE"], surface_init: str = None,
                 use_octree: bool = True, trainable_fake_sample_std: bool = False,
                 force_alpha: bool = False):
        super().__init__()
        if basis_type != CONSTANTS["BASIS_TYPE_SH"]:
            _next_row("N4", "learned bases (3D texture / MLP)")
        if background_nlayers != 0:
            _next_row("N4", "MSI background layers")
        assert shutils.isqrt(basis_dim) is not None, "basis_dim (SH) must be a square number"
        assert 1 <= basis_dim <= shutils.MAX_SH_BASIS, f"basis_dim 1-{shutils.MAX_SH_BASIS} supported"
        self.basis_type, self.surface_type = basis_type, surface_type
        self.basis_dim = basis_dim
        self.step_id = 0
        self.force_alpha = force_alpha
        self.mlp_posenc_size, self.mlp_width = mlp_posenc_size, mlp_width
        self.background_nlayers, self.background_reso = 0, background_reso
        reso = [reso] * 3 if isinstance(reso, int) else list(reso)
        assert len(reso) == 3, "reso must be an integer or indexable object of 3 ints"
        if isinstance(radius, (float, int)):
            radius = [radius] * 3
        self.radius = torch.as_tensor(radius, dtype=torch.float32, device="cpu")
        self.center = torch.as_tensor(center, dtype=torch.float32, device="cpu")
        self._offset = 0.5 * (1.0 - self.center / self.radius)   # :644-645
        self._scaling = 0.5 / self.radius

        links = gridinit.init_links(reso, use_z_order=use_z_order)
        if use_sphere_bound:
            links = gridinit.sphere_bound_links(links)
        self.capacity = int(links.max()) + 1
        n = self.capacity
        self.density_data = nn.Parameter(torch.zeros(n, 1, dtype=torch.float32, device=device))
        self.sh_data = nn.Parameter(torch.zeros(n, basis_dim * 3, dtype=torch.float32, device=device))
        empty = dict(dtype=torch.float32, device=device)
        self.basis_data = nn.Parameter(torch.empty(0, 0, 0, 0, **empty), requires_grad=False)
        self.background_data = nn.Parameter(torch.empty(0, 0, 0, **empty), requires_grad=False)
        self.register_buffer("links", torch.from_numpy(links).to(device))
        self.trainable_data = ["density_data", "sh_data", "basis_data", "background_data"]

        self.surface_data = None
        self.fake_sample_std = None
        self.truncated_vol_render_a = 1
        if trainable_fake_sample_std:
            self.fake_sample_std = nn.Parameter(torch.tensor([[1.0]], dtype=torch.float32, device=device))
            self.trainable_data.append("fake_sample_std")
        self.level_set_data = None
        if surface_type == CONSTANTS["SURFACE_TYPE_SDF"]:
            self.level_set_data = torch.tensor([0.0], device=device)
            if surface_init is None:
                surf = np.zeros((n, 1), np.float32)
            elif surface_init == "sphere":
                surf = gridinit.sphere_surface_init(links)
            else:
                _next_row("N4", f"surface_init={surface_init!r}")
            self.surface_data = nn.Parameter(torch.from_numpy(surf).to(device))
            self.trainable_data.append("surface_data")
        elif surface_type == CONSTANTS["SURFACE_TYPE_UDF"]:
            if surface_init is not None:
                _next_row("N4", f"UDF surface_init={surface_init!r}")
            self.level_set_data = torch.tensor([64.0], device=device)
            self.surface_data = nn.Parameter(torch.zeros(n, 1, dtype=torch.float32, device=device))
            self.trainable_data.append("surface_data")
        elif surface_type != CONSTANTS["SURFACE_TYPE_NONE"]:
            _next_row("N4", f"surface_type={surface_type}")

        self.opt = RenderOptions()
        self.sparse_grad_indexer: Optional[torch.Tensor] = None
        self.sparse_sh_grad_indexer: Optional[torch.Tensor] = None
        self.sparse_background_indexer = None
        self.density_rms = self.surface_rms = self.fake_sample_std_rms = self.sh_rms = None
        self.background_rms = self.basis_rms = None
        self.use_octree = use_octree
        self._C = _C

    # ---- small properties (:992-1014) ----
    @property
    def data_dim(self):
        return self.sh_data.size(1) + 1

    @property
    def use_background(self):
        return False

    @property
    def shape(self):
        return list(self.links.shape) + [self.data_dim]

    def _grid_size(self):
        return torch.tensor(self.links.shape, device="cpu", dtype=torch.float32)

    def world2grid(self, points):
        """:4492-4506 (this fork has no -0.5 voxel-centre shift)"""
        gsz = self._grid_size()
        return torch.addcmul((self._offset * gsz).to(points.device), points,
                             (self._scaling * gsz).to(points.device))

    def grid2world(self, points):
        gsz = self._grid_size()
        roffset = self.radius * (-1.0) + self.center   # :4517-4519
        rscaling = 2.0 * self.radius / gsz
        return torch.addcmul(roffset.to(points.device), points, rscaling.to(points.device))

    def sample_surf_pts(self, cell_ids, n_sample: int = 5, density_thresh: float = 0, surf_lv_set: float = 0.):
        """Iso-surface points inside the listed cells along axis-parallel lines (:4525-4550 ->
        cubic_extract_iso_pts); a surface-less grid extracts the density level set at density_thresh."""
        has_surf = self.surface_data is not None
        level = (self.surface_data.data - surf_lv_set) if has_surf else (self.density_data.data - density_thresh)
        pts = _C.cubic_extract_iso_pts(self.links, level.contiguous(), self.density_data.data,
                                       cell_ids.to(device=self.links.device, dtype=torch.int).contiguous(),
                                       n_sample, density_thresh if has_surf else -100.)
        pts = pts.view(-1, 3)
        return pts[~((pts == 0.).all(dim=-1))]

    def extract_pts(self, density_thresh: float = 0, n_sample: int = 5, batch_size: int = 100000,
                    scene_scale: float = 1., to_world: bool = False, surf_lv_set: float = 0.):
        """Point cloud of the surface over every non-empty cell (:4552-4580)."""
        with torch.no_grad():
            cell_ids = self._non_empty_ids()
            parts = [self.sample_surf_pts(cell_ids[i:i + batch_size], n_sample, density_thresh, surf_lv_set)
                     for i in range(0, cell_ids.shape[0], batch_size)]
            pts = torch.cat(parts, dim=0) if parts else torch.zeros(0, 3, device=self.links.device)
            if to_world:
                pts = self.grid2world(pts) / scene_scale
            return pts

    def trunc_vol_render_rw(self, intersect_ids):
        """Truncated-hann re-weighting of the i-th intersection (:4740-4746)."""
        u = torch.clamp(self.truncated_vol_render_a - intersect_ids, 0.0, 1.0)
        return torch.clamp_min(0.5 * (1.0 - torch.cos(torch.pi * u)), self.opt.trunc_vol_weight_min)

    # ---- spec packing (:6234-6293) ----
    def _to_cpp(self, grid_coords: bool = False, replace_basis_data=None):
        g = _C.SparseGridSpec()
        g.density_data = self.density_data
        g.surface_type = self.surface_type
        if self.surface_type != CONSTANTS["SURFACE_TYPE_NONE"]:
            g.surface_data = self.surface_data
            g.level_set_data = self.level_set_data
        g.sh_data = self.sh_data
        g.links = self.links
        if grid_coords:
            g._offset = torch.zeros_like(self._offset)
            g._scaling = torch.ones_like(self._offset)
        else:
            gsz = self._grid_size()
            g._offset = self._offset * gsz
            g._scaling = self._scaling * gsz
        g.basis_dim = self.basis_dim
        g.basis_type = self.basis_type
        if self.fake_sample_std is not None:
            g.fake_sample_std = float(self.fake_sample_std.item()) if torch.is_tensor(self.fake_sample_std) \
                else float(self.fake_sample_std)
        g.truncated_vol_render_a = self.truncated_vol_render_a
        # (no reference counterpart) the renders keep their list of non-empty brick rows while `links` stays what it is
        if "_links_epoch" not in self.__dict__:
            self._bump_links_epoch()
        g.links_epoch = self.__dict__["_links_epoch"]
        return g

    def _get_data_grads(self):
        ret = {}
        for name in self.trainable_data:
            p = getattr(self, name)
            if not p.requires_grad:
                ret[name] = torch.zeros_like(p.data)
                continue
            if p.grad is None or p.grad.shape != p.data.shape:
                p.grad = torch.zeros_like(p.data)
            ret[name] = p.grad
        return ret

    # ---- rendering (:3390-3638) ----
    def _check_backend(self, no_surface: bool = False) -> str:
        """The renderer that serves this call: `no_surface` swaps the surface renderers for the
        Plenoxels one (:3554-3557, :3657-3659)."""
        backend = self.opt.backend
        if backend not in ("surf_trav", "surface", "cuvol"):
            _next_row("N4", f"renderer backend {backend!r}")
        if no_surface and backend in ("surf_trav", "surface"):
            backend = "cuvol"
        if backend != "cuvol" and self.surface_data is None:
            raise RuntimeError("the surf_trav backend needs a surface grid (surface_type SDF/UDF)")
        return backend

    def ray_voxel_lists(self, rays: Rays, skip_empty: bool = True) -> RayVoxIntersecs:
        """Per-ray voxel lists of the legacy backend (what :3415-3428 builds from
        find_ray_voxels_intersection), from the surf_trav DDA on the GPU."""
        voxel_ls, start, num = _C.ray_voxel_lists(self._to_cpp(), rays._to_cpp(), self.opt._to_cpp(),
                                                  skip_empty=skip_empty)
        return RayVoxIntersecs(voxel_ls, start, num)

    def find_ray_voxels_intersection(self, t, origins, dirs):
        """(l int64[VV,3], ray_ids int64[VV]) for GRID-space rays (:1515-1594; `t` is unused, as
        at the reference's call sites :3421 / :3572 which pass None)."""
        spec = self._to_cpp(grid_coords=True)
        r = Rays(origins.float().contiguous(), dirs.float().contiguous())
        voxel_ls, start, num = _C.ray_voxel_lists(spec, r._to_cpp(), self.opt._to_cpp(), skip_empty=False)
        ray_ids = torch.repeat_interleave(torch.arange(num.shape[0], device=num.device), num.long())
        return voxel_ls.long(), ray_ids

    def volume_render(self, rays: Rays, use_kernel: bool = True, randomize: bool = False,
                      return_raylen: bool = False, **kwargs):
        """Differentiable render of a ray batch; returns {'rgb': [Q,3]} (:3390-3474)."""
        if not use_kernel or return_raylen:
            raise NotImplementedError("the pure-PyTorch debugging renderer is not part of this build; "
                                      "the HIP path is the only implementation (no CPU fallback)")
        backend = self._check_backend(kwargs.get("no_surface", False))
        assert rays.is_cuda and self.links.is_cuda
        if backend == "cuvol":  # :3456-3466
            rgb = _VolumeRenderFunction.apply(
                self.density_data, self.sh_data, None, None, self._to_cpp(), rays._to_cpp(),
                self.opt._to_cpp(randomize=randomize), backend)
            return {"rgb": rgb}
        if backend == "surface":  # :3410-3437
            ray_vox = self.ray_voxel_lists(rays)
            rgb = _SurfaceRenderFunction.apply(
                self.density_data, self.surface_data, self.sh_data, None, None, self._to_cpp(),
                rays._to_cpp(), ray_vox._to_cpp(), self.opt._to_cpp(randomize=randomize), self.opt.backend)
            return {"rgb": rgb}
        rgb = _SurfTravRenderFunction.apply(
            self.density_data, self.surface_data, self.sh_data, None, None, self.fake_sample_std,
            self._to_cpp(), rays._to_cpp(), self.opt._to_cpp(randomize=randomize), self.opt.backend)
        return {"rgb": rgb}

    def volume_render_fused(self, rays: Rays, rgb_gt: torch.Tensor, randomize: bool = False,
                            beta_loss: float = 0.0, sparsity_loss: float = 0.0,
                            fused_surf_norm_reg_scale: float = 0.0,
                            fused_surf_norm_reg_con_check: bool = True,
                            fused_surf_norm_reg_ignore_empty: bool = False,
                            lambda_l2: float = 1.0, lambda_l1: float = 0.0, lambda_l_dist: float = 0.0,
                            lambda_l_entropy: float = 0.0, no_norm_weight_l_entropy: bool = False,
                            lambda_l_dist_a: float = 0.0, lambda_l_entropy_a: float = 0.0,
                            lambda_l_samp_dist: float = 0.0, lambda_l_di: float = 0.0,
                            l_di_alpha_thresh: float = 0.0, surf_sparse_alpha_thresh: float = 0.0,
                            lambda_inplace_surf_sparse: float = 0.0, lambda_inwards_norm_loss: float = 0.0,
                            lambda_conv_mode_samp: float = 0.0, l_dist_max_sample: int = 64,
                            no_surface: bool = False, global_n_rays: int = 0,
                            coherent_order: Optional[bool] = None, image_width: int = 0):
        """Render + fused loss gradients accumulated into the parameters' .grad (:3477-3638).
        `image_width` (extension): the batch is a row-major image of that many rays a row (asurf_fused_params::image_width:
        large batches are then rendered in the order of their 8 x 8 pixel tiles, transparently).
        `global_n_rays` (extension) = ray count of the whole data-parallel batch; `coherent_order` (extension): True
        renders a shuffled batch in a coherent order and returns the colours in the caller's order, False renders it as
        given, None leaves it to the library (asurf_fused_params::order_rays: surf_trav batches of 65 536 rays and
        more are ordered inside the call when the last batch arrived shuffled)."""
        assert self.sh_data.is_cuda and rays.is_cuda, "the HIP extension is required for fused"
        perm = None
        in_library = self._check_backend(no_surface) == "surf_trav" and rays.origins.shape[0] >= 65536
        order_rays = 0 if coherent_order is None else (2 if coherent_order else 1)
        if coherent_order and not in_library and rays.origins.shape[0] > 1:
            perm = _C.ray_order(rays.origins.contiguous(), rays.dirs.contiguous()).long()
            rays = Rays(rays.origins[perm].contiguous(), rays.dirs[perm].contiguous())
            rgb_gt = rgb_gt[perm].contiguous()

        def finish(rgb_sorted):
            if perm is None:
                return {"rgb": rgb_sorted}
            rgb = torch.empty_like(rgb_sorted)
            rgb[perm] = rgb_sorted
            return {"rgb": rgb}
        backend = self._check_backend(no_surface)
        grad = self._get_data_grads()
        # (surf_trav writes every ray's colour -- k_comp or the one-kernel path, a miss gets the background: no fill)
        rgb_out = torch.zeros_like(rgb_gt) if backend != "surf_trav" else torch.empty_like(rgb_gt)
        # (a fresh mask per call, :3535 -- zeroed as 8-byte words: torch's bool fill is seven times slower)
        n_rows = self.density_data.size(0)
        # (surf_trav: the mask of the SH step -- the reference's clone of this one, taken before the regularisers add
        # their rows, :3655 -- is a second fresh mask the call marks as well (asurf_grads::mask_sh_out): both come out of
        # ONE fill, and the copy's launch is gone)
        words = (n_rows + 7) // 8
        both = torch.zeros(((2 if backend == "surf_trav" else 1) * words,), dtype=torch.int64,
                           device=self.density_data.device)
        self.sparse_grad_indexer = both[:words].view(torch.bool)[:n_rows]
        sh_indexer = both[words:].view(torch.bool)[:n_rows] if backend == "surf_trav" else None
        holder = _C.GridOutputGrads()
        holder.grad_density_out = grad["density_data"]
        if "surface_data" in grad:
            holder.grad_surface_out = grad["surface_data"]
        holder.grad_sh_out = grad["sh_data"]
        holder.mask_out = self.sparse_grad_indexer
        holder.mask_sh_out = sh_indexer
        if backend == "cuvol":  # :3625-3634
            rspec = rays._to_cpp()
            if image_width:
                rspec.image_width = int(image_width)
            _C.volume_render_cuvol_fused(self._to_cpp(), rspec, self.opt._to_cpp(randomize=randomize),
                                         rgb_gt, beta_loss, sparsity_loss, rgb_out, holder)
            self.sparse_sh_grad_indexer = self.sparse_grad_indexer.clone()
            return finish(rgb_out)
        if backend == "surface":  # :3560-3597
            ray_vox = self.ray_voxel_lists(rays)
            _C.volume_render_surface_fused(
                self._to_cpp(), rays._to_cpp(), ray_vox._to_cpp(), self.opt._to_cpp(randomize=randomize),
                rgb_gt, beta_loss, sparsity_loss, rgb_out, holder)
            self.sparse_sh_grad_indexer = self.sparse_grad_indexer.clone()
            return finish(rgb_out)
        if "fake_sample_std" in grad:
            holder.grad_fake_sample_std_out = grad["fake_sample_std"]
        _C.volume_render_surf_trav_fused(
            self._to_cpp(), rays._to_cpp(), self.opt._to_cpp(randomize=randomize), rgb_gt, beta_loss,
            sparsity_loss, fused_surf_norm_reg_scale, fused_surf_norm_reg_con_check,
            fused_surf_norm_reg_ignore_empty, lambda_l2, lambda_l1, lambda_l_dist, lambda_l_entropy,
            no_norm_weight_l_entropy, lambda_l_dist_a, lambda_l_entropy_a, lambda_l_samp_dist,
            lambda_l_di, l_di_alpha_thresh, surf_sparse_alpha_thresh, lambda_inplace_surf_sparse,
            lambda_inwards_norm_loss, lambda_conv_mode_samp, l_dist_max_sample, rgb_out, holder,
            global_n_rays=global_n_rays, order_rays=order_rays, image_width=image_width)
        self.sparse_sh_grad_indexer = sh_indexer
        return finish(rgb_out)

    def volume_render_image(self, camera: Camera, use_kernel: bool = True, randomize: bool = False,
                            batch_size: Optional[int] = None, return_raylen: bool = False, debug_pixels=None,
                            **kwargs):
        """Whole-image render in ray batches (:3641-3691): no image kernel is registered for
        surf_trav in the reference either (svox2.cpp:153); `no_surface=True` renders with cuvol
        (the reference's volume_render_cuvol_image, :3657-3659)."""
        rays = camera.gen_rays()
        kw = dict(use_kernel=use_kernel, randomize=randomize, no_surface=kwargs.get("no_surface", False))
        if debug_pixels is not None:
            ids = debug_pixels[:, 0] + debug_pixels[:, 1] * camera.width
            return self.volume_render(rays[ids], **kw)["rgb"]
        # (rays per call: _image_batch)
        step = _image_batch(batch_size)
        # (whole rows a call: the cuvol kernels then take the batch by pixel tiles, asurf_rays::image_width)
        whole_rows = camera.width if (step % camera.width == 0 or step >= camera.height * camera.width) else 0
        parts = [self.volume_render(Rays(rays.origins[s:s + step], rays.dirs[s:s + step], rays.masks[s:s + step],
                                         image_width=whole_rows), **kw)["rgb"]
                 for s in range(0, camera.height * camera.width, step)]
        return torch.cat(parts, dim=0).view(camera.height, camera.width, -1)

    # ---- point sampling and resampling (:1040-1113, 3963-4180 -> svox2_kernel.cu) ----
    def sample(self, points: torch.Tensor, use_kernel: bool = True, grid_coords: bool = False,
               want_colors: bool = True):
        """(density[N,1], sh[N,D]) by trilinear interpolation, border padding, empty vertices = 0 (:1040-1113).
        `grid_coords`: points are grid coordinates in [-0.5, reso-0.5]."""
        if not use_kernel:
            raise NotImplementedError("the PyTorch debugging path is not part of this build (no CPU fallback)")
        assert points.is_cuda and self.links.is_cuda
        return _SampleGridAutogradFunction.apply(self.density_data, self.sh_data,
                                                 self._to_cpp(grid_coords=grid_coords), points.contiguous(),
                                                 want_colors)

    def sample_surface(self, points: torch.Tensor, use_kernel: bool = True, grid_coords: bool = False,
                       want_colors: bool = True, want_surfaces: bool = True, default_surf: float = -1):
        """(sh, surface) at points; empty surface vertices read as `default_surf` (:1115-1146)."""
        if not use_kernel:
            raise NotImplementedError("the PyTorch debugging path is not part of this build (no CPU fallback)")
        assert points.is_cuda and self.links.is_cuda
        return _C.sample_grid_sh_surf(self._to_cpp(grid_coords=grid_coords), points.contiguous(), want_colors,
                                      want_surfaces, default_surf)

    def resample(self, reso, sigma_thresh: float = 5.0, weight_thresh: float = 0.01, dilate: int = 2,
                 cameras=None, use_z_order: bool = False, accelerate: bool = True,
                 weight_render_stop_thresh: float = 0.2, max_elements: int = 0, batch_size: int = 720720):
        """Resample density and SH onto a grid of resolution `reso`, keep the vertices whose density reaches
        `sigma_thresh` -- or, given `cameras`, whose largest compositing weight over those views reaches
        `weight_thresh` (grid_weight_render) -- dilated `dilate` times, and renumber them in raster order
        (:3963-4180).  Morton order is not offered."""
        if use_z_order:
            _next_row("N4", "resample into Morton order")
        if self.surface_data is not None:
            raise RuntimeError("resample() drops the surface plane: use resample_surface() on a surface grid")
        with torch.no_grad():
            device = self.links.device
            reso = [reso] * 3 if isinstance(reso, int) else list(reso)
            assert len(reso) == 3, "reso must be an integer or indexable object of 3 ints"
            cur = self.links.shape
            fac = [0.5 * cur[i] / reso[i] for i in range(3)]
            ax = [torch.linspace(fac[i] - 0.5, cur[i] - fac[i] - 0.5, reso[i], dtype=torch.float32) for i in range(3)]
            points = torch.stack(torch.meshgrid(*ax, indexing="ij"), dim=-1).view(-1, 3).to(device)
            dens = torch.cat([self.sample(points[i:i + batch_size], grid_coords=True, want_colors=False)[0]
                              for i in range(0, len(points), batch_size)], dim=0).view(reso)
            if cameras is not None:  # :4058-4098
                gsz = torch.tensor(reso, dtype=torch.float32)
                offset, scaling = self._offset * gsz - 0.5, self._scaling * gsz
                max_wt = torch.zeros(reso, dtype=torch.float32, device=device)
                dens = dens.contiguous()
                for cam in cameras:
                    _C.grid_weight_render(dens, cam._to_cpp(), 0.5, weight_render_stop_thresh, False, offset,
                                          scaling, max_wt)
                mask = max_wt >= weight_thresh
                if 0 < max_elements < max_wt.numel() and max_elements < int(torch.count_nonzero(mask)):
                    bound = torch.topk(max_wt.view(-1), k=max_elements, sorted=False).values.min().item()
                    weight_thresh = max(weight_thresh, bound)
                    mask = max_wt >= weight_thresh
                del max_wt
            else:
                mask = dens >= sigma_thresh
                if 0 < max_elements < dens.numel() and max_elements < int(torch.count_nonzero(mask)):
                    bound = torch.topk(dens.view(-1), k=max_elements, sorted=False).values.min().item()
                    sigma_thresh = max(sigma_thresh, bound)
                    mask = dens >= sigma_thresh
                if self.opt.last_sample_opaque:
                    mask[:, :, -1] = True
            for _ in range(int(dilate)):
                mask = _C.dilate(mask.contiguous())
            mask = mask.view(-1)
            kept_pts = points[mask]
            sh = torch.cat([self.sample(kept_pts[i:i + batch_size], grid_coords=True, want_colors=True)[1]
                            for i in range(0, len(kept_pts), batch_size)], dim=0) if len(kept_pts) else \
                torch.empty_like(self.sh_data.data[:0])
            links = torch.cumsum(mask.to(torch.int32), dim=-1).int() - 1
            links[~mask] = -1
            self.density_data = nn.Parameter(dens.view(-1)[mask].view(-1, 1).contiguous())
            self.sh_data = nn.Parameter(sh.contiguous())
            self.links = links.view(reso).contiguous()
            self.density_rms = self.sh_rms = None
            self.sparse_grad_indexer = self.sparse_sh_grad_indexer = None
            self._data_to_links = None
            if accelerate:
                self.accelerate()

    def resample_surface(self, reso, alpha_thresh: float = 1e-6, weight_thresh: float = 0.01, dilate: int = 2,
                         cameras=None, use_z_order: bool = False, accelerate: bool = True,
                         weight_render_stop_thresh: float = 0.2, max_elements: int = 0, batch_size: int = 720720,
                         alpha_empty_val: float = -20.0, non_expanding: bool = False):
        """Resample a surface grid (raw opacity, SH and the surface field) onto resolution `reso` (:4172-4420):
        vertices whose raw opacity reaches `alpha_thresh` (its logit under the sigmoid activation) -- or, given
        `cameras`, whose largest compositing weight reaches `weight_thresh` -- are kept after `dilate` dilations;
        `non_expanding` also drops vertices that fall into cells the old surface field does not cover.
        Returns the kept fraction.  Morton order is not offered."""
        if use_z_order:
            _next_row("N4", "resample into Morton order")
        if self.surface_data is None:
            raise RuntimeError("resample_surface() needs a surface grid: use resample()")
        with torch.no_grad():
            device = self.links.device
            reso = [reso] * 3 if isinstance(reso, int) else list(reso)
            assert len(reso) == 3, "reso must be an integer or indexable object of 3 ints"
            cur = self.links.shape
            ax = [torch.linspace(0.0, cur[i] - 0.5, reso[i], dtype=torch.float32) for i in range(3)]  # :4225-4243
            points = torch.stack(torch.meshgrid(*ax, indexing="ij"), dim=-1).view(-1, 3).to(device)
            sigmoid = self.opt.alpha_activation_type == CONSTANTS["SIGMOID_FN"]
            spec = self._to_cpp(grid_coords=True)
            raw = torch.cat([_C.sample_grid_raw_alpha(spec, points[i:i + batch_size].contiguous(),
                                                      alpha_empty_val if sigmoid else 0.0)
                             for i in range(0, len(points), batch_size)], dim=0).view(reso)
            if cameras is not None:
                gsz = torch.tensor(reso, dtype=torch.float32)
                max_wt = torch.zeros(reso, dtype=torch.float32, device=device)
                raw = raw.contiguous()
                for cam in cameras:
                    _C.grid_weight_render(raw, cam._to_cpp(), 0.5, weight_render_stop_thresh, False,
                                          self._offset * gsz - 0.5, self._scaling * gsz, max_wt)
                mask = max_wt >= weight_thresh
                if 0 < max_elements < max_wt.numel() and max_elements < int(torch.count_nonzero(mask)):
                    bound = torch.topk(max_wt.view(-1), k=max_elements, sorted=False).values.min().item()
                    weight_thresh = max(weight_thresh, bound)
                    mask = max_wt >= weight_thresh
                del max_wt
            else:
                if sigmoid:
                    alpha_thresh = float(np.log(alpha_thresh / (1.0 - alpha_thresh)))
                mask = raw >= alpha_thresh
                if 0 < max_elements < raw.numel() and max_elements < int(torch.count_nonzero(mask)):
                    bound = torch.topk(raw.view(-1), k=max_elements, sorted=False).values.min().item()
                    alpha_thresh = max(alpha_thresh, bound)
                    mask = raw >= alpha_thresh
                if self.opt.last_sample_opaque:
                    mask[:, :, -1] = True
            for _ in range(int(dilate)):
                mask = _C.dilate(mask.contiguous())
            mask = mask.view(-1).bool()
            raw = raw.view(-1)[mask]
            kept_pts = points[mask]
            default_surf = float("nan") if non_expanding else -1.0
            parts = [self.sample_surface(kept_pts[i:i + batch_size], grid_coords=True, want_colors=True,
                                         default_surf=default_surf) for i in range(0, len(kept_pts), batch_size)]
            sh = torch.cat([p[0] for p in parts], dim=0) if parts else torch.empty_like(self.sh_data.data[:0])
            surf = torch.cat([p[1] for p in parts], dim=0) if parts else torch.empty_like(self.surface_data.data[:0])
            if non_expanding:  # :4357-4368
                valid = ~torch.isnan(surf)[:, 0]
                ids = torch.arange(mask.shape[0], device=device)[mask]
                mask[ids] = valid
                sh, surf, raw = sh[valid], surf[valid], raw[valid]
            ratio = float(torch.count_nonzero(mask)) / mask.numel()
            links = torch.cumsum(mask.to(torch.int32), dim=-1).int() - 1
            links[~mask] = -1
            self.density_data = nn.Parameter(raw.view(-1, 1).contiguous())
            self.sh_data = nn.Parameter(sh.contiguous())
            self.surface_data = nn.Parameter(surf.view(-1, 1).contiguous())
            self.links = links.view(reso).contiguous()
            self.capacity = int(raw.shape[0])
            self.density_rms = self.sh_rms = self.surface_rms = None
            self.sparse_grad_indexer = self.sparse_sh_grad_indexer = None
            self._data_to_links = None
            if accelerate:
                self.accelerate()
            return ratio

    # ---- grid maintenance (:2859-2957, 4483-4490 -> misc_kernel.cu) ----
    def accelerate(self):
        """Distance-encode the empty links for empty-space skipping (:4483-4490).  The surf_trav
        kernels only test the sign of a link (the reference compiles its own use of the distances
        out, surf_trav.cu:31), so rendering is unchanged by it."""
        assert self.links.is_cuda, "the HIP extension is required for accelerate"
        _C.accel_dist_prop(self.links)
        self.__dict__["_non_empty_cache"] = None  # links were rewritten through a raw pointer
        self._bump_links_epoch()

    def safe_fetch_data(self, xyz, data, default=0):
        """data rows at integer grid coordinates, `default` outside the grid / at empty links (:2859-2872)."""
        out = torch.full([xyz.shape[0], data.shape[-1]], default, device=data.device, dtype=data.dtype)
        shape = torch.tensor(self.links.shape, device=xyz.device)
        inside = (xyz >= 0).all(dim=-1) & (xyz < shape[None, :]).all(dim=-1)
        x, y, z = xyz[inside].unbind(-1)
        links = self.links[x, y, z]
        valid = links >= 0
        idx = torch.arange(out.shape[0], device=data.device)[inside][valid]
        out[idx] = data[links[valid].long(), :]
        return out

    def prune_grid(self, density_raw_thres=1, dilate=2, prune_surf=True):
        """Keep the vertices whose raw opacity exceeds the threshold and (prune_surf) whose surface value
        changes sign against one of its 26 neighbours, dilated `dilate` times; renumber the rows in
        raster order (:2874-2957).  Returns the kept fraction."""
        with torch.no_grad():
            device = self.density_data.device
            reso = tuple(self.links.shape)
            ax = [torch.arange(r, device=device) for r in reso]
            coords = torch.stack(torch.meshgrid(*ax, indexing="ij"), dim=-1).view(-1, 3)
            density = self.safe_fetch_data(coords, self.density_data.data)
            sh = self.safe_fetch_data(coords, self.sh_data.data)
            valid = (density > density_raw_thres).view(reso)
            surface = None
            if self.surface_data is not None:
                surface = self.safe_fetch_data(coords, self.surface_data.data)
                if prune_surf:
                    off = torch.tensor([-1, 0, 1], device=device)
                    offsets = torch.stack(torch.meshgrid(off, off, off, indexing="ij"), dim=-1).view(-1, 3)
                    crossing = torch.zeros_like(coords[:, :1]).bool()
                    for i in range(offsets.shape[0]):
                        nb = self.safe_fetch_data(coords + offsets[i, None, :], self.surface_data.data)
                        crossing |= (surface == 0) | (nb == 0) | (torch.sign(surface) != torch.sign(nb))
                    valid &= crossing.view(reso)
            for _ in range(int(dilate)):
                valid = _C.dilate(valid.contiguous())
            valid = valid.view(-1)
            self.density_data = nn.Parameter(density[valid])
            self.sh_data = nn.Parameter(sh[valid])
            if surface is not None:
                self.surface_data = nn.Parameter(surface[valid])
            links = torch.cumsum(valid.to(torch.int32), dim=-1).int() - 1
            links[~valid] = -1
            self.links = links.view(reso).contiguous()
            self.density_rms = self.surface_rms = self.sh_rms = None
            self.sparse_grad_indexer = self.sparse_sh_grad_indexer = None
            self._data_to_links = None
            kept = float(torch.count_nonzero(valid).item()) / valid.numel()
            self.accelerate()
            return kept

    def init_surface_from_density(self, density_lvs=[0.5], reset_alpha=False, alpha_rescale=None,
                                  surface_rescale=0.1, alpha_clip_thresh=1e-6, reset_all=False,
                                  prune_threshold=1e-8, dilate=2, init_type="density", weight_init_cams=None,
                                  visibility_pruning_scale=0, mask_pruning_rays=None, zero_lv_density=None):
        """Start the surface field from a trained density grid (:2958-3227), `init_type='density'` with the
        exponential activation (the live configuration, opt.py:458-475): surface = density - level, pruned,
        then rescaled so that the mean forward-difference gradient norm is 1 (an SDF-like scale); the level
        sets are shifted and rescaled with it.  Any other `init_type` weights the density by the largest
        transmittance `weight_init_cams` see in front of the vertex first (sparse_grid_weight_render, :3118-3147).
        The visibility / mask-pruning passes need sparse_grid_visbility_render_surf / sparse_grid_mask_render
        (N4, not built)."""
        if visibility_pruning_scale > 0 or mask_pruning_rays is not None:
            _next_row("N4", "init_surface_from_density with visibility / mask pruning")
        if init_type != "density" and not weight_init_cams:
            raise RuntimeError("init_surface_from_density: init_type != 'density' needs weight_init_cams")
        if self.opt.alpha_activation_type == CONSTANTS["SIGMOID_FN"]:
            raise NotImplementedError("init_surface_from_density: the reference raises for the sigmoid "
                                      "activation as well (svox2.py:3016)")
        with torch.no_grad():
            if zero_lv_density is None:
                zero_lv_density = density_lvs[len(density_lvs) // 2]
            device = self.density_data.device
            self.level_set_data = torch.tensor(density_lvs, device=device, dtype=torch.float32) - zero_lv_density
            self.surface_type = CONSTANTS["SURFACE_TYPE_SDF"] if self.surface_type == CONSTANTS["SURFACE_TYPE_NONE"] \
                else self.surface_type
            surface = self.density_data.detach().clone()
            if init_type != "density":
                gsz = self._grid_size()
                max_t = torch.zeros(tuple(self.links.shape), dtype=torch.float32, device=device)
                for cam in weight_init_cams:
                    _C.sparse_grid_weight_render(self._to_cpp(), cam._to_cpp(), 0.5, 0.0, self._offset * gsz,
                                                 self._scaling * gsz, max_t)
                surface = surface * max_t[self.links >= 0][:, None]
            self.surface_data = nn.Parameter((surface - zero_lv_density).to(torch.float32))
            if "surface_data" not in self.trainable_data:
                self.trainable_data.append("surface_data")
            self.prune_grid(prune_threshold, dilate, prune_surf=False)
            # mean |forward difference| / h over the cells whose +x, +y, +z neighbours exist
            cells = self._get_rand_cells_non_empty(1, contiguous=True).long()
            sx, sy, sz = self.links.shape
            z = cells % sz
            y = (cells // sz) % sy
            x = cells // (sz * sy)
            inner = ~((x >= sx - 1) | (y >= sy - 1) | (z >= sz - 1))
            x, y, z = x[inner], y[inner], z[inner]
            l000, l100 = self.links[x, y, z], self.links[x + 1, y, z]
            l010, l001 = self.links[x, y + 1, z], self.links[x, y, z + 1]
            ok = ~((l000 < 0) | (l100 < 0) | (l010 < 0) | (l001 < 0))
            s = self.surface_data.data
            s0 = s[l000[ok].long()]
            h = self._get_h()
            norm_grad = torch.sqrt(((s[l100[ok].long()] - s0) / h) ** 2 + ((s[l010[ok].long()] - s0) / h) ** 2 +
                                   ((s[l001[ok].long()] - s0) / h) ** 2)
            scale = norm_grad.mean()
            self.surface_data.data = self.surface_data.data / scale
            self.level_set_data = self.level_set_data / scale
            if alpha_rescale is not None:
                self.density_data.data *= alpha_rescale

    # ---- evaluation renders (:3693-3961 -> surf_trav.cu:3944-4081) ----
    def _eval_backend(self, kwargs):
        if kwargs.get("no_surface", False):
            _next_row("N4", "no_surface normal / alpha / point renders")
        self._check_backend()
        if self.opt.backend != "surf_trav":
            _next_row("N3", f"evaluation renders of backend {self.opt.backend!r}")

    def volume_render_depth(self, rays: Rays, sigma_thresh: Optional[float] = None, depth_type="mean",
                            weight_thresh: float = 0.0, **kwargs):
        """Depth along each ray (absolute length, not z-depth): expected termination, the max-weight
        sample (`depth_type='mode'`), or the first sample above `sigma_thresh` (:3693-3763)."""
        g, r, o = self._to_cpp(), rays._to_cpp(), self.opt._to_cpp()
        if self._check_backend(kwargs.get("no_surface", False)) == "cuvol":  # :3736-3772
            if sigma_thresh is not None:
                return _C.volume_render_sigma_thresh(g, r, o, sigma_thresh)
            if depth_type == "mean":
                return _C.volume_render_expected_term(g, r, o, weight_thresh)
            if depth_type == "mode":
                return _C.volume_render_mode_term(g, r, o, weight_thresh)
            if depth_type == "med":
                depths, sigmas = _C.volume_render_med_term(g, r, o, 1024)
                nan_mask = (depths == 0.0) | (sigmas < 0.5)
                depths[nan_mask] = torch.nan
                sigmas[nan_mask] = torch.nan
                ids = torch.nanmedian(sigmas, dim=-1).indices
                return depths[torch.arange(depths.shape[0], device=depths.device), ids]
            raise NotImplementedError(f"depth_type {depth_type} not recognised!")
        self._eval_backend(kwargs)
        if sigma_thresh is None:
            if depth_type == "mode":
                return _C.volume_render_mode_term_surf_trav(g, r, o, weight_thresh)
            return _C.volume_render_expected_term_surf_trav(g, r, o)
        return _C.volume_render_sigma_thresh_surf_trav(g, r, o, sigma_thresh)

    def volume_render_normal(self, rays: Rays, **kwargs):
        """:3765-3784"""
        self._eval_backend(kwargs)
        return _C.render_normal_surf_trav(self._to_cpp(), rays._to_cpp(), self.opt._to_cpp())

    def volume_render_alpha_map(self, rays: Rays, thresh: float, **kwargs):
        """:3786-3806"""
        self._eval_backend(kwargs)
        return _C.volume_render_alpha_surf_trav(self._to_cpp(), rays._to_cpp(), self.opt._to_cpp(), thresh)

    def _image_batches(self, camera: Camera, batch_size: int, fn):
        batch_size = _image_batch(batch_size)
        rays = camera.gen_rays()
        return torch.cat([fn(rays[s:s + batch_size]) for s in range(0, camera.height * camera.width, batch_size)],
                         dim=0)

    def volume_render_depth_image(self, camera: Camera, sigma_thresh: Optional[float] = None,
                                  batch_size: Optional[int] = None, **kwargs):
        """:3808-3826"""
        out = self._image_batches(camera, batch_size, lambda r: self.volume_render_depth(r, sigma_thresh, **kwargs))
        return out.view(camera.height, camera.width)

    def volume_render_normal_image(self, camera: Camera, batch_size: Optional[int] = None, **kwargs):
        """:3828-3845"""
        out = self._image_batches(camera, batch_size, lambda r: self.volume_render_normal(r, **kwargs))
        return out.view(camera.height, camera.width, 3)

    def volume_render_alpha_image(self, camera: Camera, thresh: float = -1, batch_size: Optional[int] = None, **kwargs):
        """:3847-3864"""
        out = self._image_batches(camera, batch_size, lambda r: self.volume_render_alpha_map(r, thresh, **kwargs))
        return out.view(camera.height, camera.width, 1)

    def volume_render_extract_pts(self, camera: Camera, sigma_thresh: Optional[float] = None,
                                  batch_size: Optional[int] = None, depth_type="mean", **kwargs):
        """One 3-D point per ray that has a depth (:3866-3912)."""
        rays = camera.gen_rays()
        depth = self._image_batches(
            camera, batch_size, lambda r: self.volume_render_depth(r, sigma_thresh, depth_type=depth_type, **kwargs))
        pts = rays.origins + rays.dirs * depth[:, None]
        return pts[(depth != 0.0) & (~torch.isnan(depth))]

    def volume_render_extract_pts_with_alpha(self, camera: Camera, sigma_thresh: Optional[float] = None,
                                             batch_size: Optional[int] = None, max_sample: int = 20, **kwargs):
        """Every intersection with alpha above `sigma_thresh`, up to max_sample per ray (:3914-3961)."""
        self._eval_backend(kwargs)
        rays = camera.gen_rays()
        all_pts, all_alpha = [], []
        batch_size = _image_batch(batch_size)
        for s in range(0, camera.height * camera.width, batch_size):
            rb = rays[s:s + batch_size]
            depths, alphas = _C.extract_pts_surf_trav(self._to_cpp(), rb._to_cpp(), self.opt._to_cpp(), max_sample,
                                                      sigma_thresh)
            mask = depths != 0.0
            all_alpha.append(alphas[mask])
            all_pts.append((rb.origins[:, None, :] + rb.dirs[:, None, :] * depths[..., None])[mask])
        return torch.cat(all_pts, dim=0), torch.cat(all_alpha, dim=0)

    # ---- per-iteration regularisers (:4947-5165, 5686-5766 -> loss_kernel.cu) ----
    def _get_sparse_grad_indexer(self):
        """:6295-6299"""
        if self.sparse_grad_indexer is None:
            return torch.empty((0,), dtype=torch.bool, device=self.density_data.device)
        return self.sparse_grad_indexer

    def _get_sparse_sh_grad_indexer(self):
        if self.sparse_sh_grad_indexer is None:
            return torch.empty((0,), dtype=torch.bool, device=self.density_data.device)
        return self.sparse_sh_grad_indexer

    def _get_h(self):
        """:3228-3232"""
        return 2.0 * float(self.radius.mean()) / float(self._grid_size().mean())

    def _get_rand_cells(self, sparse_frac: float, force: bool = False, contiguous: bool = True):
        """Cell ids for a stochastic regulariser pass (reference: svox2.py:6335-6352): None stands for "every cell";
        otherwise max(1, frac * cells) flat ids, either one run that starts at a random cell and wraps around the end
        of the volume, or independent draws.  The host / device generators are consulted exactly as the reference
        does (one numpy draw for the run's start, one torch draw for the independent ids), so seeded runs line up."""
        if not force and sparse_frac >= 1.0:
            return None
        if self.sparse_grad_indexer is not None and self.sparse_grad_indexer.dtype != torch.bool:
            raise AssertionError("sparse regularisers come after the render and before the optimizer step "
                                 "(the row indexer has already been turned into an index list)")
        n_cells = self.links.numel()
        count = max(1, int(sparse_frac * n_cells))
        dev = self.links.device
        if not contiguous:
            return torch.randint(0, n_cells, (count,), dtype=torch.int32, device=dev)
        if self.__dict__.get("_device_rng"):   # (inside a captured iteration: the start is drawn on the device)
            first = (torch.rand(1, device=dev) * n_cells).long().clamp_(max=n_cells - 1)
            return (torch.arange(count, dtype=torch.int64, device=dev) + first).remainder_(n_cells).to(torch.int32)
        first = int(np.random.randint(0, n_cells))
        if first + count <= n_cells:   # (no wrap: a view of the cached 0 .. n_cells-1 -- no kernel at all; the callers only read it)
            if n_cells <= (1 << 26):
                ids = self.__dict__.get("_all_cell_ids")
                if ids is None or ids.numel() != n_cells or ids.device != dev:
                    ids = torch.arange(n_cells, dtype=torch.int32, device=dev)
                    self.__dict__["_all_cell_ids"] = ids
                return ids[first:first + count]
            return torch.arange(first, first + count, dtype=torch.int32, device=dev)
        run = torch.arange(count, dtype=torch.int64, device=dev).add_(first)
        return run.remainder_(n_cells).to(torch.int32)

    def _get_rand_cells_non_empty(self, sparse_frac: float, frac_of_remaining: bool = True,
                                  contiguous: bool = True, borrow: bool = False):
        """:6354-6376: the same over the cells whose first corner exists.  `borrow`: the caller only reads the list
        (the regulariser kernels): it may be handed the cached tensor or a view of it instead of a copy (two copies of
        5 MB an iteration at 256^3 otherwise)."""
        grid_size = self.links.numel()
        non_empty_ids = self._non_empty_ids()
        if sparse_frac >= 1.0:
            return non_empty_ids if borrow else non_empty_ids.clone()  # never the cached tensor itself
        non_empty_num = len(non_empty_ids)
        if frac_of_remaining:
            sparse_num = int(non_empty_num * sparse_frac)
        else:
            sparse_num = min(int(sparse_frac * grid_size), int(non_empty_num))
        if contiguous:
            if self.__dict__.get("_device_rng"):   # (inside a captured iteration: the start is drawn on the device)
                span = non_empty_num - sparse_num + 1
                start = (torch.rand(1, device=non_empty_ids.device) * span).long().clamp_(max=span - 1)
                return non_empty_ids.index_select(0, torch.arange(sparse_num, device=non_empty_ids.device) + start)
            start = np.random.randint(0, non_empty_num - sparse_num + 1)
            run = non_empty_ids[start:start + sparse_num]
            return run if borrow else run.clone()
        return non_empty_ids[torch.randint(0, non_empty_num, (sparse_num,), dtype=torch.long,
                                           device=self.links.device)]

    def _non_empty_ids(self):
        """Flat ids of the cells whose first corner exists.  The reference recomputes this list in each of the
        four regulariser calls of an iteration (a compare and a compaction over the whole link volume: 0.1 ms
        each at 256^3); here it is kept until `links` is reassigned (__setattr__) or modified in place (_version)."""
        key = (self.links._version, tuple(self.links.shape))
        cached = self.__dict__.get("_non_empty_cache")
        if cached is None or cached[0] != key:
            cached = (key, torch.where(self.links.view(-1) >= 0)[0].int())
            self.__dict__["_non_empty_cache"] = cached
        return cached[1]

    def __setattr__(self, name, value):
        if name == "links":
            self.__dict__["_non_empty_cache"] = None
            self._bump_links_epoch()
        super().__setattr__(name, value)

    def _bump_links_epoch(self):
        """`links` was replaced, or written through a raw pointer: the spec's links_epoch (include/asurf.h) changes,
        and the next render rebuilds its list of non-empty brick rows.  In-place tensor writes need no call: the
        tensor's version counter is part of the value (csrc_api._pack_grid).  The value comes from ONE process-wide
        counter: an object's id() is reused as soon as the object is freed, so two successive grids of the same
        shape could otherwise present the same epoch (and often the same links pointer and version)."""
        self.__dict__["_links_epoch"] = next(_LINKS_EPOCHS)

    def capture_iteration(self, fn, rays: Rays, rgb_gt: torch.Tensor, warmup: int = 3):
        """(extension; the reference has no counterpart)  One training iteration -- `fn(rays, rgb_gt)`: the fused render,
        the regularisers, the optimizer steps, whatever the caller does per batch with THIS grid -- captured into one
        HIP graph and returned as a callable `step(rays, rgb_gt)` that copies a new batch of the same shape into the
        captured buffers and replays the graph.  At the training script's batch size the iteration is bound by the
        host (a few dozen launches and their Python glue: 0.37 ms per iteration against 0.25 ms of kernels); a replay
        is one call.  Requirements: shapes and `links` stay what they were (re-capture after up-sampling / pruning);
        nothing in `fn` reads device memory on the host (a trainable `fake_sample_std` does: `.item()`).  The
        random cell blocks of the sparse regularisers draw their start on the device inside the graph (torch's
        graph-safe generator) instead of from numpy.  Returns (step, outputs) with `outputs` what `fn` returned for
        the captured batch (its tensors are rewritten by every replay)."""
        assert rays.is_cuda and rgb_gt.is_cuda
        s_rays = Rays(rays.origins.clone(), rays.dirs.clone())
        s_gt = rgb_gt.clone()
        stream = torch.cuda.Stream(device=rgb_gt.device)
        stream.wait_stream(torch.cuda.current_stream(rgb_gt.device))
        self.__dict__["_device_rng"] = True
        try:
            with torch.cuda.stream(stream):
                for _ in range(max(1, warmup)):
                    fn(s_rays, s_gt)
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                outputs = fn(s_rays, s_gt)
        finally:
            self.__dict__["_device_rng"] = False

        def step(new_rays: Rays = None, new_gt: torch.Tensor = None):
            # (one multi-tensor copy instead of three launches; a caller that writes its batches into the captured
            # buffers itself -- `step.buffers` -- passes nothing and pays the replay alone)
            if new_rays is not None:
                torch._foreach_copy_([s_rays.origins, s_rays.dirs, s_gt], [new_rays.origins, new_rays.dirs, new_gt],
                                     non_blocking=True)
            graph.replay()
            return outputs
        step.buffers = (s_rays, s_gt)
        return step, outputs

    def _get_data_to_links(self):
        """:6378-6382"""
        d2l = getattr(self, "_data_to_links", None)
        if d2l is None or d2l.numel() != self.density_data.numel():
            d2l = torch.arange(self.links.numel(), dtype=torch.int, device=self.links.device)[self.links.view(-1) >= 0]
            self._data_to_links = d2l
        return d2l

    def tv(self, logalpha: bool = False, logalpha_delta: float = 2.0, ndc_coeffs=(-1.0, -1.0)):
        """svox2.py:4890-4909: total variation of the density plane (sum over channels, mean over cells)."""
        assert self.sh_data.is_cuda, "the HIP extension is required for total variation (no CPU fallback)"
        assert not logalpha, "No longer supported"
        if self.surface_type != SURFACE_TYPE_NONE:
            raise NotImplementedError
        return _TotalVariationFunction.apply(self.density_data, self.links, 0, 1, logalpha, logalpha_delta,
                                             False, ndc_coeffs)

    def tv_color(self, start_dim: int = 0, end_dim: Optional[int] = None, logalpha: bool = False,
                 logalpha_delta: float = 2.0, ndc_coeffs=(-1.0, -1.0)):
        """svox2.py:4911-4938: total variation of SH channels [start_dim, end_dim), edges ignored."""
        assert self.sh_data.is_cuda, "the HIP extension is required for total variation (no CPU fallback)"
        assert not logalpha, "No longer supported"
        n = self.sh_data.size(1)
        if end_dim is None:
            end_dim = n
        end_dim = end_dim + n if end_dim < 0 else end_dim
        start_dim = start_dim + n if start_dim < 0 else start_dim
        return _TotalVariationFunction.apply(self.sh_data, self.links, start_dim, end_dim, logalpha,
                                             logalpha_delta, True, ndc_coeffs)

    def inplace_tv_grad(self, grad: torch.Tensor, scaling: float = 1.0, sparse_frac: float = 0.01,
                        logalpha: bool = False, logalpha_delta: float = 2.0, ndc_coeffs=(-1.0, -1.0),
                        contiguous: bool = True):
        """TV of the raw opacity added into `grad` (:4947-4987)."""
        assert self.density_data.is_cuda and grad.is_cuda, "the HIP extension is required for total variation"
        assert not logalpha, "No longer supported"
        rand_cells = self._get_rand_cells(sparse_frac, contiguous=contiguous)
        if rand_cells is None:  # :4977-4986
            _C.tv_grad(self.links, self.density_data, 0, 1, scaling, logalpha, logalpha_delta, False, ndc_coeffs[0],
                       ndc_coeffs[1], grad)
            self.sparse_grad_indexer = None
            return
        if rand_cells.size(0) > 0:
            _C.tv_grad_sparse(self.links, self.density_data, rand_cells, self._get_sparse_grad_indexer(), 0, 1,
                              scaling, logalpha, logalpha_delta, False, self.opt.last_sample_opaque,
                              ndc_coeffs[0], ndc_coeffs[1], grad)

    def inplace_tv_color_grad(self, grad: torch.Tensor, start_dim: int = 0, end_dim: Optional[int] = None,
                              scaling: float = 1.0, sparse_frac: float = 0.01, logalpha: bool = False,
                              logalpha_delta: float = 2.0, ndc_coeffs=(-1.0, -1.0), contiguous: bool = True):
        """TV of SH coefficients [start_dim, end_dim) added into `grad` (:5768-5825)."""
        assert self.sh_data.is_cuda and grad.is_cuda, "the HIP extension is required for total variation"
        assert not logalpha, "No longer supported"
        n = self.sh_data.size(1)
        end_dim = n if end_dim is None else (end_dim + n if end_dim < 0 else end_dim)
        start_dim = start_dim + n if start_dim < 0 else start_dim
        rand_cells = self._get_rand_cells(sparse_frac, contiguous=contiguous)
        if rand_cells is None:  # :5813-5824
            _C.tv_grad(self.links, self.sh_data, start_dim, end_dim, scaling, logalpha, logalpha_delta, True,
                       ndc_coeffs[0], ndc_coeffs[1], grad)
            self.sparse_sh_grad_indexer = None
            return
        if rand_cells.size(0) > 0:
            _C.tv_grad_sparse(self.links, self.sh_data, rand_cells, self._get_sparse_sh_grad_indexer(),
                              start_dim, end_dim, scaling, logalpha, logalpha_delta, True, False,
                              ndc_coeffs[0], ndc_coeffs[1], grad)

    def inplace_l2_color_grad(self, grad: torch.Tensor, start_dim: int = 0, end_dim: Optional[int] = None,
                              scaling: float = 1.0):
        """L2 penalty on SH coefficients [start_dim, end_dim) added into `grad`, over the rows the last render
        touched (:5897-5928; plain tensor arithmetic in the reference too)."""
        with torch.no_grad():
            n = self.sh_data.size(1)
            end_dim = n if end_dim is None else (end_dim + n if end_dim < 0 else end_dim)
            start_dim = start_dim + n if start_dim < 0 else start_dim
            if self.sparse_sh_grad_indexer is None:
                grad[:, start_dim:end_dim] += (scaling / self.sh_data.size(0)) * self.sh_data[:, start_dim:end_dim]
            else:
                indexer = self._maybe_convert_sparse_grad_indexer(sh=True)
                nz = int(torch.count_nonzero(indexer)) if indexer.dtype == torch.bool else indexer.size(0)
                grad[indexer, start_dim:end_dim] += (scaling / nz) * self.sh_data[indexer, start_dim:end_dim]

    def inplace_tv_surface_grad(self, grad: torch.Tensor, scaling: float = 1.0, sparse_frac: float = 0.01,
                                ndc_coeffs=(-1.0, -1.0), alpha_dependency: bool = True,
                                ignore_edge: bool = True, edge_value: float = -1.0, contiguous: bool = True):
        """TV of the surface scalar added into `grad` (:5121-5163)."""
        assert self.surface_data.is_cuda and grad.is_cuda, "the HIP extension is required for total variation"
        rand_cells = self._get_rand_cells_non_empty(sparse_frac, contiguous=contiguous, borrow=True)
        if rand_cells.size(0) > 0:
            _C.surf_tv_grad_sparse(self.links, self.surface_data, self.density_data, rand_cells,
                                   self._get_sparse_grad_indexer(), 0, 1, scaling, ignore_edge, edge_value,
                                   False, ndc_coeffs[0], ndc_coeffs[1], alpha_dependency, grad)

    def inplace_surface_normal_grad(self, grad: torch.Tensor, scaling: float = 1.0, sparse_frac: float = 0.01,
                                    ndc_coeffs=(-1.0, -1.0), contiguous: bool = True, use_kernel: bool = True,
                                    connectivity_check: bool = True, ignore_empty: bool = False,
                                    use_l1: bool = False, **kwargs):
        """Smoothness of the surface normal between neighbouring cells (:5686-5731)."""
        if self.surface_data is None:
            return
        if not use_kernel:
            raise NotImplementedError("the PyTorch debugging path is not part of this build (no CPU fallback)")
        assert self.surface_data.is_cuda and grad.is_cuda, "the HIP extension is required"
        rand_cells = self._get_rand_cells_non_empty(sparse_frac, contiguous=contiguous, borrow=True)
        if rand_cells.size(0) > 0:
            _C.surface_normal_grad_sparse(self.links, self.surface_data, rand_cells,
                                          self._get_sparse_grad_indexer(), 0.0, 0, 1, scaling, 0.0,
                                          ndc_coeffs[0], ndc_coeffs[1], connectivity_check, ignore_empty,
                                          use_l1, grad)

    def inplace_surface_eikonal_grad(self, grad: torch.Tensor, scaling: float = 1.0, sparse_frac: float = 0.01,
                                     contiguous: bool = True, use_kernel: bool = True, **kwargs):
        """Eikonal loss; returns its value (:5733-5766).  The reference runs it through autograd
        (`use_kernel=False`, opt.py:1004-1009), which accumulates into surface_data.grad and leaves the
        sparse indexer alone; so does this (whatever `use_kernel` says: there is one implementation)."""
        if self.surface_data is None:
            return
        rand_cells = self._get_rand_cells_non_empty(sparse_frac, contiguous=contiguous, borrow=True)
        target = self._get_data_grads()["surface_data"]
        if rand_cells.size(0) == 0:
            return torch.zeros((), device=self.surface_data.device)
        return _C.surface_eikonal_grad_sparse(self.links, self.surface_data, rand_cells, None, scaling,
                                              self._get_h(), target)

    def inplace_alpha_surf_sparsify_grad(self, grad_alpha: torch.Tensor, grad_surf: torch.Tensor,
                                         scaling_alpha: float = 1.0, scaling_surf: float = 1.0,
                                         sparse_frac: float = 0.01, surf_sparse_decrease: bool = True,
                                         surf_sparse_thresh: float = 0.01, alpha_sparsify_bound: float = 1e-6,
                                         surf_sparsify_bound: float = -1, only_trained_cells: bool = False,
                                         trained_cells_mask: torch.Tensor = None, contiguous: bool = True):
        """log-alpha / surface sparsification (:5053-5119)."""
        assert self.density_data.is_cuda, "the HIP extension is required"
        if only_trained_cells:
            rand_cells = self._get_data_to_links()[trained_cells_mask].to(self.density_data.device)
        else:
            rand_cells = self._get_rand_cells_non_empty(sparse_frac, contiguous=contiguous, borrow=True)
        if self.opt.alpha_activation_type == CONSTANTS["SIGMOID_FN"]:
            surf_sparse_thresh = float(np.log(surf_sparse_thresh / (1.0 - surf_sparse_thresh)))
            alpha_sparsify_bound = float(np.log(alpha_sparsify_bound / (1.0 - alpha_sparsify_bound)))
        if rand_cells.size(0) > 0:
            _C.alpha_surf_sparsify_grad_sparse(self.links, self.density_data, self.surface_data,
                                               rand_cells.contiguous(), self._get_sparse_grad_indexer(),
                                               scaling_alpha, scaling_surf, surf_sparse_decrease,
                                               surf_sparse_thresh, alpha_sparsify_bound, surf_sparsify_bound,
                                               grad_alpha, grad_surf)

    # ---- optimizers (:5972-6140 -> optim_kernel.cu:14-265) ----
    def _maybe_convert_sparse_grad_indexer(self, sh=False, bg=False):
        """The row selector of an optimizer step (:6314-6333).  The reference turns the mask into an index list on
        the host when under 1/8 of the rows were touched -- `torch.count_nonzero(mask).item()`, a device -> host
        round trip in every optimizer step that stalls the launch queue of the whole iteration.  Here the mask
        itself goes to `_C.rmsprop_step` / `sgd_step`, which build the row list on the device
        (asurf_rmsprop_step_masked): the same rows are updated, nothing waits."""
        indexer = self.sparse_sh_grad_indexer if sh else self.sparse_grad_indexer
        if indexer is None:
            return torch.empty((), device=self.density_data.device)
        return indexer

    def _optim_step(self, name, rms_name, indexer, lr, beta, epsilon, optim):
        p = getattr(self, name)
        if optim == "rmsprop":
            rms = getattr(self, rms_name)
            if rms is None or rms.shape != p.shape:
                rms = torch.zeros_like(p.data)
                setattr(self, rms_name, rms)
            _C.rmsprop_step(p.data, rms, p.grad, indexer, beta, lr, epsilon, -1e9, lr)
        elif optim == "sgd":
            _C.sgd_step(p.data, p.grad, indexer, lr, lr)
        else:
            raise NotImplementedError(f"Unsupported optimizer {optim}")

    def optim_density_step(self, lr: float, beta: float = 0.9, epsilon: float = 1e-8, optim: str = "rmsprop"):
        self._optim_step("density_data", "density_rms", self._maybe_convert_sparse_grad_indexer(),
                         lr, beta, epsilon, optim)

    def optim_surface_step(self, lr: float, beta: float = 0.9, epsilon: float = 1e-8, optim: str = "rmsprop"):
        if self.surface_data is None:
            return
        self._optim_step("surface_data", "surface_rms", self._maybe_convert_sparse_grad_indexer(),
                         lr, beta, epsilon, optim)

    def optim_sh_step(self, lr: float, beta: float = 0.9, epsilon: float = 1e-8, optim: str = "rmsprop"):
        self._optim_step("sh_data", "sh_rms", self._maybe_convert_sparse_grad_indexer(sh=True),
                         lr, beta, epsilon, optim)

    def optim_fake_sample_std_step(self, lr: float, beta: float = 0.9, epsilon: float = 1e-8,
                                   optim: str = "rmsprop", lambda_l2: float = 0.0, lambda_l1: float = 0.0):
        """:6058-6107.  The reference passes `self.surface_rms` as the RMS state here (a slip: it
        reads and overwrites surface_rms[0]); this mirror keeps a dedicated state tensor."""
        if self.fake_sample_std is None:
            return
        p = self.fake_sample_std
        p.grad += 2 * lambda_l2 * p.data + lambda_l1 * torch.sign(p.data)
        self._optim_step("fake_sample_std", "fake_sample_std_rms",
                         torch.empty((), device=p.device), lr, beta, epsilon, optim)

    # ---- checkpointing (:4693-4838): same .npz keys ----
    def save(self, path: str, compress: bool = False, step_id: int = 0):
        data = dict(radius=self.radius.numpy(), center=self.center.numpy(),
                    links=self.links.cpu().numpy(), density_data=self.density_data.data.cpu().numpy(),
                    sh_data=self.sh_data.data.cpu().numpy().astype(np.float16),
                    basis_type=self.basis_type, surface_type=self.surface_type, step_id=step_id)
        if self.surface_data is not None:
            data["surface_data"] = self.surface_data.data.cpu().numpy()
            data["level_set_data"] = self.level_set_data.cpu().numpy()
        if self.fake_sample_std is not None:
            data["fake_sample_std"] = self.fake_sample_std.data.cpu().numpy() \
                if torch.is_tensor(self.fake_sample_std) else self.fake_sample_std
        (np.savez_compressed if compress else np.savez)(path, **data)

    @classmethod
    def load(cls, path: str, device: Union[torch.device, str] = "cpu"):
        """svox2.py:4749-4838, same keys.  The throw-away constructor grid is 2^3 instead of the reference's
        dense `list(links.shape)` volume (every tensor it would allocate is replaced below)."""
        z = np.load(path, allow_pickle=True)
        surface_type = CONSTANTS["SURFACE_TYPE_NONE"]
        surface_data = None
        if "data" in z.files:  # :4755-4759, the pre-surface layout: one array, density first
            all_data = z["data"]
            sh_data, density_data = all_data[..., 1:], all_data[..., :1]
        else:
            sh_data, density_data = z["sh_data"], z["density_data"]
            surface_type = int(z["surface_type"].item()) if "surface_type" in z.files else surface_type
            if surface_type != CONSTANTS["SURFACE_TYPE_NONE"]:
                surface_data = z["surface_data"]
        if "background_data" in z.files and z["background_data"].size > 0:
            _next_row("N4", "MSI background layers in a checkpoint")
        basis_type = int(z["basis_type"].item()) if "basis_type" in z.files else CONSTANTS["BASIS_TYPE_SH"]
        if "basis_data" in z.files and z["basis_data"].size > 0:
            basis_type = CONSTANTS["BASIS_TYPE_3D_TEXTURE"]  # :4819-4824
        links = z["links"]
        grid = cls(2, radius=z["radius"].tolist() if "radius" in z.files else [1.0, 1.0, 1.0],
                   center=z["center"].tolist() if "center" in z.files else [0.0, 0.0, 0.0],
                   basis_dim=sh_data.shape[1] // 3, use_z_order=False, device=device, basis_type=basis_type,
                   mlp_posenc_size=int(z["mlp_posenc_size"].item()) if "mlp_posenc_size" in z.files else 0,
                   mlp_width=int(z["mlp_width"].item()) if "mlp_width" in z.files else 16,
                   background_nlayers=0, surface_type=surface_type)
        if "step_id" in z.files:
            grid.step_id = int(z["step_id"].item())

        def dev(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device=device)
        grid.sh_data = nn.Parameter(dev(sh_data))
        grid.density_data = nn.Parameter(dev(density_data))
        if surface_data is not None:
            grid.surface_data = nn.Parameter(dev(surface_data))
        if "level_set_data" in z.files:
            grid.level_set_data = dev(z["level_set_data"])
        if "fake_sample_std" in z.files:  # any rank; a plain tensor, as in the reference (:4808-4809)
            grid.fake_sample_std = dev(z["fake_sample_std"])
        grid.links = torch.from_numpy(np.ascontiguousarray(links)).to(device=device)
        grid.capacity = grid.sh_data.size(0)
        if grid.links.is_cuda:
            grid.accelerate()
        return grid

    def __repr__(self):
        return (f"alphasurf_amd.SparseGrid(basis_dim={self.basis_dim}, reso={list(self.links.shape)}, "
                f"capacity={self.sh_data.size(0)})")


def _add_next_rows():
    rows = {
        "N2": ["inplace_tv_grad", "inplace_tv_surface_grad", "inplace_surface_normal_grad",
               "inplace_alpha_surf_sparsify_grad", "inplace_surface_eikonal_grad", "inplace_tv_color_grad",
               "inplace_surface_sign_change_grad", "inplace_tv_lumisphere_grad", "inplace_tv_background_grad",
               "inplace_l2_color_grad", "inplace_alpha_lap_grad", "_get_rand_cells_non_empty", "tv_basis"],
        "N3": ["volume_render_depth", "volume_render_depth_image", "volume_render_normal_image",
               "volume_render_alpha_image", "extract_pts"],
        "N4": ["init_surface_from_density", "accelerate", "resample", "resample_surface", "prune_grid",
               "sparsify_background", "extract_mesh", "reinit_learned_bases", "optim_basis_step",
               "optim_background_step", "sample", "_surface_render_gradcheck_lerp",
               "_surface_norm_match_loss_grad_check", "_surface_viscosity_loss_grad_check",
               "_eval_learned_bases", "_eval_basis_mlp"],
    }
    for row, names in rows.items():
        for name in names:
            if name in SparseGrid.__dict__:
                continue  # built
            def stub(self, *a, _n=name, _r=row, **k):
                _next_row(_r, f"SparseGrid.{_n}")
            stub.__name__ = name
            setattr(SparseGrid, name, stub)


_add_next_rows()

__all__ = ["SparseGrid", "Rays", "RayVoxIntersecs", "Camera", "RenderOptions"] + list(CONSTANTS)
