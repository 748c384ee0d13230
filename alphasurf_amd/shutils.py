"""Small host utilities the reference keeps in svox2/utils.py: SH basis evaluation on torch
tensors, Morton helpers, the extension loader and a HIP-event timer (own code; reference lines
cited for parity)."""
import math

import torch

MAX_SH_BASIS = 16  # the reference stops at 10 (utils.py:118); wave64 lanes make 16 natural here

_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
       0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
       -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


def isqrt(n: int):
    r = math.isqrt(n) if n >= 0 else -1
    return r if r * r == n else None


def is_pow2(x: int):
    return x > 0 and (x & (x - 1)) == 0


def eval_sh_bases(basis_dim: int, dirs: torch.Tensor):
    """SH basis values at unit directions, [..., basis_dim] (utils.py:119-166, up to degree 3)."""
    out = torch.empty((*dirs.shape[:-1], basis_dim), dtype=dirs.dtype, device=dirs.device)
    out[..., 0] = _C0
    if basis_dim > 1:
        x, y, z = dirs.unbind(-1)
        out[..., 1], out[..., 2], out[..., 3] = -_C1 * y, _C1 * z, -_C1 * x
        if basis_dim > 4:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            out[..., 4] = _C2[0] * xy
            out[..., 5] = _C2[1] * yz
            out[..., 6] = _C2[2] * (2.0 * zz - xx - yy)
            out[..., 7] = _C2[3] * xz
            out[..., 8] = _C2[4] * (xx - yy)
            if basis_dim > 9:
                out[..., 9] = _C3[0] * y * (3 * xx - yy)
                out[..., 10] = _C3[1] * xy * z
                out[..., 11] = _C3[2] * y * (4 * zz - xx - yy)
                out[..., 12] = _C3[3] * z * (2 * zz - 3 * xx - 3 * yy)
                out[..., 13] = _C3[4] * x * (4 * zz - xx - yy)
                out[..., 14] = _C3[5] * z * (xx - yy)
                out[..., 15] = _C3[6] * x * (xx - 3 * yy)
    return out


def gen_morton(D, device="cpu", dtype=torch.long):
    from . import gridinit
    return torch.from_numpy(gridinit.init_links([D] * 3, use_z_order=True)).to(device=device, dtype=dtype)


def _get_c_extension():
    """utils.py:32-45: the operator module, or a loud failure -- there is no slow path here."""
    from . import csrc_api, lib
    lib.lib()
    assert hasattr(csrc_api, "sample_grid")
    return csrc_api


class Timing:
    """`with Timing("msg"):` prints the elapsed GPU time (utils.py:615-635)."""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.start = torch.cuda.Event(enable_timing=True)
        self.end = torch.cuda.Event(enable_timing=True)
        self.start.record()

    def __exit__(self, *exc):
        self.end.record()
        torch.cuda.synchronize()
        print(self.name, "elapsed", self.start.elapsed_time(self.end), "ms")
