"""The N>1 path on CPU: world_size-2 gloo runs of the gradient exchange used by ray-sharded
rendering (alphasurf_amd/dist.py).  Property: after the exchange every rank holds the SUM of
the per-shard gradients and the UNION of the touched-row masks."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_grads(rank, n, d, frac, seed):
    rs = np.random.RandomState(seed + rank)
    mask = rs.rand(n) < frac
    g_sh = np.zeros((n, d), np.float32)
    g_de = np.zeros((n, 1), np.float32)
    g_su = np.zeros((n, 1), np.float32)
    k = int(mask.sum())
    g_sh[mask] = rs.randn(k, d)
    g_de[mask] = rs.randn(k, 1)
    g_su[mask] = rs.randn(k, 1)
    return g_sh, g_de, g_su, mask


def _worker(rank, world, port, mode, frac, n, d, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from alphasurf_amd import dist as adist
        g_sh, g_de, g_su, mask = _shard_grads(rank, n, d, frac, 42)
        t_sh, t_de, t_su = torch.from_numpy(g_sh), torch.from_numpy(g_de), torch.from_numpy(g_su)
        t_m = torch.from_numpy(mask.copy())
        t_std = torch.tensor([[float(rank + 1)]])
        if mode == "pipelined":   # the step in two parts (dist.PipelinedReduce): each part a shard of its own
            parts = []
            for part in range(2):
                a = adist.GradArena(n, d, torch.device("cpu"))
                p_sh, p_de, p_su, p_m = _shard_grads(rank, n, d, frac, 42 + 1000 * part)
                a.grad_sh.copy_(torch.from_numpy(p_sh))
                a.grad_density.copy_(torch.from_numpy(p_de))
                a.grad_surface.copy_(torch.from_numpy(p_su))
                a.grad_fake_sample_std.fill_(float(rank + 1 + 10 * part))
                a.mask.copy_(torch.from_numpy(p_m.copy()))
                parts.append(a)
            px = adist.PipelinedReduce(torch.device("cpu"))
            px.submit(parts[0])
            px.progress()
            px.submit(parts[1])
            used = px.finish(parts[0])
            a = parts[0]
            exp = [_shard_grads(r, n, d, frac, 42 + 1000 * part) for r in range(world) for part in range(2)]
            ok = (np.allclose(a.grad_sh.numpy(), sum(e[0] for e in exp), atol=1e-5)
                  and np.allclose(a.grad_density.numpy(), sum(e[1] for e in exp), atol=1e-5)
                  and np.allclose(a.grad_surface.numpy(), sum(e[2] for e in exp), atol=1e-5)
                  and (a.mask.numpy() == np.logical_or.reduce([e[3] for e in exp])).all()
                  and float(a.grad_fake_sample_std.item()) == sum(r + 1 + 10 * part for r in range(world) for part in range(2)))
            out.put((rank, bool(ok), used))
            return
        if mode.startswith("arena"):  # the same buffers as views of one flat allocation, one collective
            arena = adist.GradArena(n, d, torch.device("cpu"))
            arena.grad_sh.copy_(t_sh)
            arena.grad_density.copy_(t_de)
            arena.grad_surface.copy_(t_su)
            arena.grad_fake_sample_std.copy_(t_std)
            arena.mask.copy_(t_m)
            base = arena.flat.data_ptr()
            assert all((v.data_ptr() - base) % 256 == 0 for v in (arena.grad_sh, arena.grad_density, arena.grad_surface))
            used = arena.all_reduce(mode={"arena": "dense", "arena_sparse": "sparse", "arena_auto": "auto"}[mode])
            t_sh, t_de, t_su, t_m, t_std = (arena.grad_sh, arena.grad_density, arena.grad_surface, arena.mask,
                                            arena.grad_fake_sample_std)
            used = "arena:" + used
        else:
            used = adist.all_reduce_grads(t_sh, t_de, t_su, t_m, t_std, mode=mode)
        exp = [_shard_grads(r, n, d, frac, 42) for r in range(world)]
        e_sh = sum(e[0] for e in exp)
        e_de = sum(e[1] for e in exp)
        e_su = sum(e[2] for e in exp)
        e_m = np.logical_or.reduce([e[3] for e in exp])
        ok = (np.allclose(t_sh.numpy(), e_sh, atol=1e-6) and np.allclose(t_de.numpy(), e_de, atol=1e-6)
              and np.allclose(t_su.numpy(), e_su, atol=1e-6) and (t_m.numpy() == e_m).all()
              and float(t_std.item()) == sum(range(1, world + 1)))
        out.put((rank, bool(ok), used))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode,frac,expect", [("dense", 0.3, "dense"), ("sparse", 0.05, "sparse"),
                                              ("auto", 0.03, "sparse"), ("auto", 0.5, "dense"),
                                              ("sparse", 0.0, "sparse"), ("arena", 0.3, "arena:dense"),
                                              ("arena_sparse", 0.2, "arena:sparse"), ("arena_auto", 0.1, "arena:sparse"),
                                              ("arena_auto", 0.6, "arena:dense"), ("pipelined", 0.1, "pipelined"),
                                              ("pipelined", 0.0, "pipelined")])
def test_gradient_exchange_world2(mode, frac, expect):
    world, n, d = 2, 4001, 27
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, frac, n, d, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, used in res:
        assert ok, f"rank {rank}: exchanged gradients differ from the sum of the shards"
        assert used == expect


def test_shard_rays_partition():
    from alphasurf_amd import dist as adist
    for n, w in [(640000, 8), (10, 4), (7, 8), (0, 2)]:
        got = []
        for r in range(w):
            lo, hi = adist.shard_rays(n, r, w)
            got.extend(range(lo, hi))
        assert got == list(range(n))
