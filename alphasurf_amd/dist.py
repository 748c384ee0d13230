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


DENSE_ABOVE = 0.5   # union of touched rows above this fraction of the grid: the dense ring moves fewer bytes per row


def _union_rows(mask, group):
    """The union of the ranks' touched rows: `mask` all-reduced (max) in place, then the ordered list of the rows it
    names -- identical on every rank -- and its length, the ONE number the host reads per exchange."""
    m8 = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
    _reduce(m8, dist.ReduceOp.MAX, group)
    if mask.is_cuda:
        from . import csrc_api as _C
        rows, count = _C.rows_list(mask)
        return rows, int(count.item())
    rows = torch.nonzero(mask.view(-1), as_tuple=False).view(-1).to(torch.int32)
    return rows, int(rows.numel())


def all_reduce_sparse(tensors, mask, group=None, extra=None, dense_above=None, dense_flat=None):
    """Sum keyed by the UNION of the ranks' touched rows (`mask_out`, render_util.cuh:138-142).

    One small all-reduce (max) of the masks gives every rank the same union mask and therefore the same ordered row
    list; each rank packs its values of those rows (zeros where it touched nothing) into a dense [U, D + 2] buffer,
    ONE ring all-reduce sums the buffers, and the sums are written back (asurf_rows_list / _pack / _unpack on the
    device: include/asurf.h).  Per rank the ring moves 2 (N-1)/N * U * row_bytes instead of ... * all_rows * row_bytes,
    with one host read (U, the buffer's size) instead of the four of a variable-size all-to-all.
    `extra`: a tensor of a few floats (grad_fake_sample_std) summed in the same collective.
    `dense_above`: when the union exceeds this fraction of the rows the buffers are all-reduced as they are --
    `dense_flat`, when given, is the one allocation that holds them all (GradArena.flat) and is reduced instead.
    Returns "sparse" or "dense"."""
    if _world(group) == 1:
        return "single"
    if len(tensors) != 3 or tensors[0] is None or tensors[1] is None or tensors[0].dim() != 2:
        raise ValueError("all_reduce_sparse: tensors = [grad_sh [n, D], grad_density [n, 1], grad_surface [n, 1] or None] "
                         f"(got {[None if t is None else tuple(t.shape) for t in tensors]}); all_reduce_dense takes any list")
    grad_sh, grad_density, grad_surface = tensors
    n, D = grad_sh.shape[0], grad_sh.shape[1]
    if grad_density.numel() != n or (grad_surface is not None and grad_surface.numel() != n) or mask.numel() != n:
        raise ValueError("all_reduce_sparse: grad_density, grad_surface and mask must have one entry per row of grad_sh")
    rows, U = _union_rows(mask, group)
    if dense_above is not None and U > dense_above * n:
        if dense_flat is not None:     # the arena's buffers are ONE allocation: one ring instead of four
            _reduce(dense_flat, dist.ReduceOp.SUM, group)
            return "dense"
        for t in (grad_sh, grad_density, grad_surface, extra):
            if t is not None:
                _reduce(t, dist.ReduceOp.SUM, group)
        return "dense"
    W = D + 2
    n_extra = 0 if extra is None else extra.numel()
    packed = torch.empty(U * W + n_extra, dtype=grad_sh.dtype, device=grad_sh.device)
    if grad_sh.is_cuda:
        from . import csrc_api as _C
        _C.rows_pack(rows, U, grad_sh, grad_density, grad_surface, packed)
    elif U:
        idx = rows[:U].long()
        body = packed[:U * W].view(U, W)
        body[:, :D] = grad_sh[idx]
        body[:, D] = grad_density.view(-1)[idx]
        body[:, D + 1] = grad_surface.view(-1)[idx] if grad_surface is not None else 0.0
    if n_extra:
        packed[U * W:] = extra.view(-1)
    _reduce(packed, dist.ReduceOp.SUM, group)
    if grad_sh.is_cuda:
        _C.rows_unpack(rows, U, packed, grad_sh, grad_density, grad_surface, None)   # (the mask IS the union already)
    elif U:
        idx = rows[:U].long()
        body = packed[:U * W].view(U, W)
        grad_sh[idx] = body[:, :D]
        grad_density.view(-1)[idx] = body[:, D]
        if grad_surface is not None:
            grad_surface.view(-1)[idx] = body[:, D + 1]
    if n_extra:
        extra.view(-1).copy_(packed[U * W:])
    return "sparse"


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

    def all_reduce(self, group=None, mode="dense"):
        """Sum over the ranks in place; `mask` becomes the union of the ranks' masks.  mode "dense": one ring
        all-reduce of the whole allocation; "sparse": the rows of the union only (all_reduce_sparse); "auto": sparse
        unless the union exceeds DENSE_ABOVE of the rows."""
        if _world(group) == 1:
            return "single"
        if mode == "dense":
            self.touched.copy_(self.mask)
            _reduce(self.flat, dist.ReduceOp.SUM, group)
            torch.gt(self.touched, 0, out=self.mask)
            return "dense"
        # (auto beyond DENSE_ABOVE: the mask is the union already -- the flat buffer goes round in one ring)
        return all_reduce_sparse([self.grad_sh, self.grad_density, self.grad_surface], self.mask, group=group,
                                 extra=self.grad_fake_sample_std, dense_above=DENSE_ABOVE if mode == "auto" else None,
                                 dense_flat=self.flat)


def _pack_rows(rows, U, grad_sh, grad_density, grad_surface, packed):
    if grad_sh.is_cuda:
        from . import csrc_api as _C
        _C.rows_pack(rows, U, grad_sh, grad_density, grad_surface, packed)
    elif U:
        D = grad_sh.shape[1]
        idx = rows[:U].long()
        body = packed[:U * (D + 2)].view(U, D + 2)
        body[:, :D] = grad_sh[idx]
        body[:, D] = grad_density.view(-1)[idx]
        body[:, D + 1] = grad_surface.view(-1)[idx] if grad_surface is not None else 0.0


def _unpack_rows(rows, U, packed, grad_sh, grad_density, grad_surface, mask, add=False):
    if grad_sh.is_cuda:
        from . import csrc_api as _C
        _C.rows_unpack(rows, U, packed, grad_sh, grad_density, grad_surface, mask, add=add)
    elif U:
        D = grad_sh.shape[1]
        idx = rows[:U].long()
        body = packed[:U * (D + 2)].view(U, D + 2)
        if add:
            grad_sh[idx] += body[:, :D]
            grad_density.view(-1)[idx] += body[:, D]
            if grad_surface is not None:
                grad_surface.view(-1)[idx] += body[:, D + 1]
        else:
            grad_sh[idx] = body[:, :D]
            grad_density.view(-1)[idx] = body[:, D]
            if grad_surface is not None:
                grad_surface.view(-1)[idx] = body[:, D + 1]
        if mask is not None:
            mask.view(-1)[idx] = True if mask.dtype == torch.bool else 1


class PipelinedReduce:
    """The exchange of a step taken in PARTS, each part's rows travelling under the render of the next part.

    Nothing inside ONE fused call can overlap its exchange: the touched-row mask is complete only when the call's last
    kernel ends.  But a rank's batch may be rendered as two (or more) partial batches -- e.g. the upper and the lower
    half of its view, each with the global ray count in the loss normaliser, each into an arena of its own:

        px = PipelinedReduce(device)
        render(part_0 -> arena_0); px.submit(arena_0)     # queues the union-mask exchange of part 0 on a side stream
        render(part_1 -> arena_1)                         # queued at once: the device goes on rendering
        px.progress()                                     # host: reads part 0's row count, packs, starts its all-reduce
        px.submit(arena_1)
        px.finish(arena_0)                                # arena_0 <- sum over ranks and parts; its mask <- the union

    so that part 0's all-reduce runs while part 1 renders, and only the last part's exchange is exposed.  The parts'
    gradients must not share buffers (part 1's atomics would race with part 0's pack).  Every rank issues the same
    collectives in the same order (mask 0, rows 0, mask 1, rows 1).  With gloo (tests, rehearsals; CPU tensors too) the
    collectives are synchronous host round trips: same results, no overlap."""

    def __init__(self, device, group=None, max_parts=2, timed=False):
        self.device, self.group = torch.device(device), group
        self.side = torch.cuda.Stream(self.device) if self.device.type == "cuda" else None
        self.parts = []
        # per part slot, allocated once (VERDICT r3 #7b: no allocation inside a step): the pinned word the row count
        # lands in, and the packed buffer, sized on first use for EVERY row of the arena (the union can be no larger)
        self._hosts = [torch.empty(1, dtype=torch.int32, pin_memory=True) if self.side is not None else None
                       for _ in range(max_parts)]
        self._packed = [None] * max_parts
        # `timed`: events on the side stream around every part's collectives (bench.py's exchange_ms); bytes a rank
        # hands to the collectives of a step
        self.timed = timed and self.side is not None
        self._ev = []
        self.last_bytes = 0

    def _packed_for(self, slot, arena, n_floats):
        buf = self._packed[slot]
        cap = arena.grad_sh.shape[0] * (arena.grad_sh.shape[1] + 2) + arena.grad_fake_sample_std.numel()
        if buf is None or buf.numel() < cap or buf.device != arena.grad_sh.device:
            buf = torch.empty(cap, dtype=arena.grad_sh.dtype, device=arena.grad_sh.device)
            self._packed[slot] = buf
        return buf[:n_floats]

    def exchange_ms(self):
        """Device time of the last step's collectives and their pack / list kernels on the side stream (timed=True; call
        after a synchronisation)."""
        ms = sum(a.elapsed_time(b) for a, b in self._ev)
        self._ev = []
        return ms

    def _on_side(self):
        import contextlib
        return torch.cuda.stream(self.side) if self.side is not None else contextlib.nullcontext()

    def submit(self, arena):
        """After the part's render has been queued on the current stream: its mask -> union over the ranks -> ordered
        row list, all on the side stream; the list's length travels to pinned host memory.  Does not block."""
        if _world(self.group) == 1:
            self.parts.append(dict(arena=arena, single=True))
            return
        if self.side is None:      # CPU tensors: everything at once
            rows, U = _union_rows(arena.mask, self.group)
            self.parts.append(dict(arena=arena, rows=rows, U=U))
            return
        from . import csrc_api as _C
        slot = len(self.parts)
        if slot >= len(self._hosts):
            raise RuntimeError(f"PipelinedReduce(max_parts={len(self._hosts)}): a step of {slot + 1} parts")
        if slot == 0:
            self.last_bytes = 0
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        with self._on_side():
            self.side.wait_event(done)
            if self.timed:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(self.side)
            _reduce(arena.mask.view(torch.uint8), dist.ReduceOp.MAX, self.group)
            rows, count = _C.rows_list(arena.mask)
            host = self._hosts[slot]
            host.copy_(count, non_blocking=True)
            known = torch.cuda.Event(enable_timing=self.timed)
            known.record(self.side)
            if self.timed:
                self._ev.append((e0, known))
        self.last_bytes += arena.mask.numel()
        self.parts.append(dict(arena=arena, rows=rows, count=count, host=host, known=known, slot=slot))

    def progress(self):
        """Host side of every submitted part that has not been packed yet: wait for its row count (the side stream
        only), pack its rows, start their all-reduce.  Call it after the NEXT part's render has been queued."""
        for p in self.parts:
            if p.get("single") or "packed" in p:
                continue
            if "U" not in p:
                p["known"].synchronize()
                p["U"] = int(p["host"][0])
            U, a = p["U"], p["arena"]
            W = a.grad_sh.shape[1] + 2
            n_tail = a.grad_fake_sample_std.numel()
            with self._on_side():
                if self.timed:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(self.side)
                if "slot" in p:
                    packed = self._packed_for(p["slot"], a, U * W + n_tail)
                else:   # (CPU tensors)
                    packed = torch.empty(U * W + n_tail, dtype=a.grad_sh.dtype, device=a.grad_sh.device)
                _pack_rows(p["rows"], U, a.grad_sh, a.grad_density, a.grad_surface, packed)
                packed[U * W:].copy_(a.grad_fake_sample_std.view(-1))
                _reduce(packed, dist.ReduceOp.SUM, self.group)
                if self.timed:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(self.side)
                    self._ev.append((e0, e1))
            self.last_bytes += packed.numel() * packed.element_size()
            p.update(W=W, packed=packed)

    def finish(self, into):
        """All parts in: `into` (the first part's arena) receives the sums over ranks and parts, its mask the union of
        the parts' unions.  The current stream waits for the side stream."""
        self.progress()
        parts, self.parts = self.parts, []
        if parts and parts[0].get("single"):          # one rank: the parts' sums, no exchange
            for p in parts:
                if p["arena"] is not into:
                    into.flat.add_(p["arena"].flat)
                    into.mask.logical_or_(p["arena"].mask)
            return "single"
        assert parts and parts[0]["arena"] is into, "finish(into): `into` is the first part's arena"
        with self._on_side():
            std = None
            for i, p in enumerate(parts):
                U, W = p["U"], p["W"]
                _unpack_rows(p["rows"], U, p["packed"], into.grad_sh, into.grad_density, into.grad_surface, into.mask,
                             add=i > 0)
                tail = p["packed"][U * W:]
                std = tail.clone() if std is None else std.add_(tail)
            into.grad_fake_sample_std.view(-1).copy_(std)
        if self.side is not None:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        return "pipelined"


def all_reduce_grads(grad_sh, grad_density, grad_surface, mask, grad_fake_sample_std=None,
                     group=None, dense_threshold=DENSE_ABOVE, mode="auto"):
    """Sum the sparse gradient buffers of a fused render across ranks.

    mode: "dense", "sparse" or "auto" (sparse unless the UNION of the ranks' touched rows exceeds `dense_threshold` of
    the grid -- known after the masks' all-reduce, at no extra cost; the reference makes the same kind of switch
    between mask and index lists at svox2/svox2.py:6326-6332)."""
    if _world(group) == 1:
        return "single"
    tensors = [grad_sh, grad_density, grad_surface]
    if mode == "dense":
        if grad_fake_sample_std is not None:
            _reduce(grad_fake_sample_std, dist.ReduceOp.SUM, group)
        all_reduce_dense(tensors, mask, group=group)
        return "dense"
    return all_reduce_sparse(tensors, mask, group=group, extra=grad_fake_sample_std,
                             dense_above=dense_threshold if mode == "auto" else None)


def shard_rays(n_rays, rank, world):
    """Contiguous ray range of this rank (contiguous image tiles keep touched-voxel sets compact)."""
    per = (n_rays + world - 1) // world
    lo = min(rank * per, n_rays)
    return lo, min(lo + per, n_rays)
