"""Data-parallel exchange for ray-sharded rendering.

The reference is single-GPU; this is new functionality.  Rays shard
across ranks, the grid is replicated, every rank calls the fused renderer with the GLOBAL
ray count in the loss normaliser, and the per-rank gradient buffers are then summed here.
After the exchange every rank holds the gradient of the whole batch and the union of the
touched-row masks, exactly what a single GPU would have produced on the concatenated rays
(up to float summation order).

Backend: torch.distributed -- "nccl" is RCCL over xGMI on ROCm; "gloo" for the CPU tests.
"""
import torch
import torch.distributed as dist


def _world(group):
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _reduce(t, op, group):
    """all_reduce in place; with the gloo backend (CPU tests, rehearsals) device tensors go
    through host memory."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)


def _staged(t, group):
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _all_to_all(out, inp, out_splits, in_splits, group):
    """all_to_all_single; device tensors go through host memory with the gloo backend."""
    if _staged(inp, group):
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h, inp.cpu(), output_split_sizes=out_splits, input_split_sizes=in_splits,
                               group=group)
        out.copy_(h)
    else:
        dist.all_to_all_single(out, inp, output_split_sizes=out_splits, input_split_sizes=in_splits, group=group)


def _all_gather(outs, inp, group):
    """all_gather into the list `outs`; device tensors go through host memory with the gloo backend."""
    if _staged(inp, group):
        hs = [torch.empty(o.shape, dtype=o.dtype) for o in outs]
        dist.all_gather(hs, inp.cpu(), group=group)
        for o, h in zip(outs, hs):
            o.copy_(h)
    else:
        dist.all_gather(outs, inp, group=group)


def all_reduce_dense(tensors, mask=None, group=None):
    """Plain sum all-reduce of every gradient buffer + OR of the mask."""
    if _world(group) == 1:
        return
    for t in tensors:
        if t is not None:
            _reduce(t, dist.ReduceOp.SUM, group)
    if mask is not None:
        m8 = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
        _reduce(m8, dist.ReduceOp.MAX, group)


def all_reduce_sparse(tensors, mask, group=None):
    """Sparse sum keyed by the touched-row mask (`mask_out`, render_util.cuh:138-142).

    Row ownership is a contiguous range per rank.  Step 1 (all-to-all): every rank sends the
    touched rows of each range to its owner, which sums them.  Step 2 (all-gather): owners
    publish their summed rows; every rank writes them back.  Per-rank volume is about
    2 * touched_rows * row_bytes instead of 2 * all_rows * row_bytes for the dense ring, and
    the all-to-all uses all xGMI links at once rather than one ring direction.
    """
    world = _world(group)
    if world == 1:
        return
    rank = dist.get_rank(group)
    tensors = [t for t in tensors if t is not None]
    n = mask.shape[0]
    dev = mask.device
    widths = [t.reshape(n, -1).shape[1] for t in tensors]
    W = sum(widths)
    # contiguous ownership ranges
    bounds = torch.linspace(0, n, world + 1, device="cpu").long().tolist()
    idx = torch.nonzero(mask.view(-1), as_tuple=False).view(-1)
    rows = torch.cat([t.reshape(n, -1)[idx] for t in tensors], dim=1) if idx.numel() else \
        torch.zeros((0, W), dtype=tensors[0].dtype, device=dev)
    owner_cuts = torch.searchsorted(idx, torch.tensor(bounds, device=dev))
    send_counts = (owner_cuts[1:] - owner_cuts[:-1]).to(torch.int64)
    recv_counts = torch.empty_like(send_counts)
    _all_to_all(recv_counts, send_counts, None, None, group)
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    recv_idx = torch.empty(sum(rc), dtype=idx.dtype, device=dev)
    recv_rows = torch.empty((sum(rc), W), dtype=rows.dtype, device=dev)
    _all_to_all(recv_idx, idx, rc, sc, group)
    _all_to_all(recv_rows, rows, rc, sc, group)
    # owner-side reduction over its range
    lo, hi = bounds[rank], bounds[rank + 1]
    acc = torch.zeros((hi - lo, W), dtype=rows.dtype, device=dev)
    acc.index_add_(0, recv_idx - lo, recv_rows)
    touched = torch.zeros(hi - lo, dtype=torch.bool, device=dev)
    touched[recv_idx - lo] = True
    own_idx = torch.nonzero(touched, as_tuple=False).view(-1)
    own_rows = acc[own_idx]
    own_idx = own_idx + lo
    # publish the summed rows
    cnt = torch.tensor([own_idx.numel()], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    _all_gather(cnts, cnt, group)
    cnts = [int(c.item()) for c in cnts]
    mx = max(max(cnts), 1)
    pad_idx = torch.zeros(mx, dtype=own_idx.dtype, device=dev)
    pad_rows = torch.zeros((mx, W), dtype=rows.dtype, device=dev)
    pad_idx[: own_idx.numel()] = own_idx
    pad_rows[: own_idx.numel()] = own_rows
    all_idx = [torch.empty_like(pad_idx) for _ in range(world)]
    all_rows = [torch.empty_like(pad_rows) for _ in range(world)]
    _all_gather(all_idx, pad_idx, group)
    _all_gather(all_rows, pad_rows, group)
    for r in range(world):
        k = cnts[r]
        if k == 0:
            continue
        ids, vals = all_idx[r][:k], all_rows[r][:k]
        c0 = 0
        for t, w in zip(tensors, widths):
            t.reshape(n, -1)[ids] = vals[:, c0:c0 + w]
            c0 += w
        mask.view(-1)[ids] = True


class GradArena:
    """The gradient buffers of one replica as views of ONE flat float32 allocation -- grad_sh [n, D] |
    grad_density [n, 1] | grad_surface [n, 1] | touched [n] | grad_fake_sample_std [1, 1] -- so that the dense
    exchange is a single ring all-reduce (and clearing them a single fill) instead of five collectives: over
    xGMI a ring all-reduce pays its latency per call, and the two [n, 1] planes and the mask are small next to
    it.  The kernels write `mask` (bool); `all_reduce` carries it through the `touched` plane."""

    ALIGN = 64  # floats: every view starts on a 256-byte boundary

    def __init__(self, n_rows, sh_dim, device):
        def up(x):
            return (x + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        sizes = [n_rows * sh_dim, n_rows, n_rows, n_rows, 1]
        offs, total = [], 0
        for sz in sizes:
            offs.append(total)
            total += up(sz)
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad_sh = self.flat[offs[0]:offs[0] + sizes[0]].view(n_rows, sh_dim)
        self.grad_density = self.flat[offs[1]:offs[1] + sizes[1]].view(n_rows, 1)
        self.grad_surface = self.flat[offs[2]:offs[2] + sizes[2]].view(n_rows, 1)
        self.touched = self.flat[offs[3]:offs[3] + sizes[3]]
        self.grad_fake_sample_std = self.flat[offs[4]:offs[4] + 1].view(1, 1)
        # the mask lives in 8-byte words: torch's fill kernel for bool takes 15 us for 1.3 M rows, for the same bytes
        # as int64 2 us (rocprofv3, profiles/r2_kernel_stats.csv)
        self._mask_words = torch.zeros((n_rows + 7) // 8, dtype=torch.int64, device=device)
        self.mask = self._mask_words.view(torch.bool)[:n_rows]

    def zero_mask(self):
        self._mask_words.zero_()

    def zero_(self):
        self.flat.zero_()
        self.zero_mask()

    def all_reduce(self, group=None):
        """Sum over the ranks in place; `mask` becomes the union of the ranks' masks."""
        if _world(group) == 1:
            return
        self.touched.copy_(self.mask)
        _reduce(self.flat, dist.ReduceOp.SUM, group)
        torch.gt(self.touched, 0, out=self.mask)


def all_reduce_grads(grad_sh, grad_density, grad_surface, mask, grad_fake_sample_std=None,
                     group=None, dense_threshold=0.125, mode="auto"):
    """Sum the sparse gradient buffers of a fused render across ranks.

    mode: "dense", "sparse" or "auto" (sparse when the fraction of touched rows on every rank
    is below `dense_threshold`, mirroring the reference's own switch between mask and index
    lists at svox2/svox2.py:6326-6332)."""
    if _world(group) == 1:
        return "single"
    if grad_fake_sample_std is not None:
        _reduce(grad_fake_sample_std, dist.ReduceOp.SUM, group)
    tensors = [grad_sh, grad_density, grad_surface]
    if mode == "auto":
        frac = mask.float().mean().reshape(1)
        _reduce(frac, dist.ReduceOp.MAX, group)
        mode = "sparse" if float(frac.item()) < dense_threshold else "dense"
    if mode == "sparse":
        all_reduce_sparse(tensors, mask, group=group)
    else:
        all_reduce_dense(tensors, mask, group=group)
    return mode


def shard_rays(n_rays, rank, world):
    """Contiguous ray range of this rank (contiguous image tiles keep touched-voxel sets compact)."""
    per = (n_rays + world - 1) // world
    lo = min(rank * per, n_rays)
    return lo, min(lo + per, n_rays)
