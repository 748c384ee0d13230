#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of one fused forward+backward pass of the surf_trav hot
path (`volume_render_surf_trav_fused`) over an 800x800 view of a 256^3 sparse SDF grid,
SH degree 2 (BASELINE.json configs[1]), plus the HBM-roofline fraction of the dominant
kernel and the CPU oracle timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one fused call on one batch of synthetic rays (one 800x800 view per rank,
inputs resident in HBM).  With N > 1 (launched by torch.distributed.run, one rank per GPU)
every rank renders its own view of the replicated grid and the gradient buffers are
all-reduced over RCCL inside the timed step: weak scaling, value = total rays / time.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reso", type=int, default=256)
    ap.add_argument("--basis-dim", type=int, default=9)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--band", type=float, default=3.0, help="half-thickness (voxels) of the active shell")
    ap.add_argument("--shuffle", action="store_true", help="training-order (shuffled) rays")
    ap.add_argument("--sort", action="store_true",
                    help="order the batch for coherence inside the timed step (asurf_ray_order), scatter rgb back")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--links-epoch", action="store_true",
                    help="tell the renders that `links` is the same from step to step, as it is in the training loop "
                         "(asurf_grid::links_epoch): the occupancy pass then reads the non-empty brick rows only.  Off by "
                         "default: every timed step does all of its work from the inputs alone")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step in a HIP graph after the warm-up and replay it in the timed region")
    ap.add_argument("--cpu-rays", type=int, default=40000)
    ap.add_argument("--exchange", default="dense", choices=["dense", "sparse", "auto"],
                    help="gradient exchange for --gpus > 1 (alphasurf_amd/dist.py)")
    return ap.parse_args()


def algorithmic_bytes(n_rays, v_t, v_a, s, D, stat_loss, M):
    """compulsory-traffic model of the REFERENCE algorithm.
    Returns (forward bytes, backward bytes) for the whole batch."""
    b_f = n_rays * (24 + 12) + 32 * v_t + 64 * v_a + s * 8 * 4 * (1 + D)
    b_b = b_f + n_rays * 12 + s * 2 * 8 * 4 * (D + 2) + s * 16
    if stat_loss:
        b_b += 2 * 3 * 4 * min(s, n_rays * M)
    return float(b_f), float(b_b)


def algorithmic_bytes_per_kernel(n_rays, v_t, v_a, s, D, stat_loss, M):
    """The same model split over the kernels of the fused call; the
    shares add up to algorithmic_bytes().  The reference re-traverses in the backward pass:
    that share goes to the kernel that reloads the voxels (k_bwd_geo)."""
    cache = 2 * 3 * 4 * min(s, n_rays * M) if stat_loss else 0
    return {
        "k_build_masks": 0.0,                                   # no counterpart in the model
        "k_build_masks_live": 0.0,                              # (the same pass over the non-empty brick rows)
        "k_sweep": float(24 * n_rays + 32 * v_t),               # ray + 8 links per visited voxel: the search for the
        "k_walk": float(24 * n_rays + 32 * v_t),                # work voxels (k_sweep, or k_walk with ASURF_WALK=step)
        "k_geom": float(64 * v_a),                              # 8 density + 8 surface values
        "k_comp": float(12 * n_rays + s * 32 * (1 + D)),        # rgb + corner density/SH rows
        "k_forward_fallback": 0.0,
        "k_bwd_seq": float(24 * n_rays + cache),                # colour cache + rgb_gt + sample cache
        "k_bwd_sh": float(s * 32 * D + s * 64 * D),             # SH rows again + grad_sh read-modify-write
        "k_bwd_geo": float(24 * n_rays + 32 * v_t + 64 * v_a + s * 32 + s * 64 * 2 + s * 16),
        "k_backward_fallback": 0.0,
    }


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU as CHILD processes through
    torch.distributed.run (nothing in this process has touched the GPU yet), pass their output through
    and return their exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def main():
    args = parse()
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # ASURF_DIST_BACKEND=gloo: rehearsal of the N > 1 control flow on fewer GPUs than ranks
        # (ranks share devices, the exchange goes through host memory); never used for results
        backend = os.environ.get("ASURF_DIST_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        local_rank = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from alphasurf_amd import csrc_api as _C, dist as adist, lib as _lib, synth

    # ---- synthetic workload (seeded; identical grid on every rank) ----
    R, B = args.reso, args.basis_dim
    D = 3 * B
    on_device = R > 256 or args.band != 3.0   # big grids: build the scene with torch on the GPU
    if on_device:
        sc = synth.make_sparse_scene_torch(R, B, dev, band=args.band, trunc_a=5.0, **synth.BENCH_SCENE)
    else:
        sc = synth.make_sparse_scene(R, B, trunc_a=5.0, **synth.BENCH_SCENE)
    o_np, d_np = synth.pinhole_rays(synth.orbit_camera(rank % 8), args.width, args.width)
    if args.shuffle:
        perm = np.random.RandomState(11 + rank).permutation(o_np.shape[0])
        o_np, d_np = o_np[perm], d_np[perm]
    Q = o_np.shape[0]
    gt_np = np.random.RandomState(7 + rank).rand(Q, 3).astype(np.float32)

    g = _C.SparseGridSpec()
    def to_dev(a):
        return a if torch.is_tensor(a) else torch.from_numpy(a).to(dev)

    g.density_data = to_dev(sc["density"])
    g.surface_data = to_dev(sc["surface"])
    g.sh_data = to_dev(sc["sh"])
    g.links = to_dev(sc["links"])
    g.level_set_data = to_dev(sc["level_set"])
    g._offset = torch.from_numpy(sc["offset"])
    g._scaling = torch.from_numpy(sc["scaling"])
    g.basis_dim, g.basis_type, g.surface_type = B, 1, 102
    g.fake_sample_std, g.truncated_vol_render_a = sc["fake_sample_std"], sc["truncated_vol_render_a"]
    # --links-epoch: as in the training loop, `links` stays what it is from step to step (densities, surfaces and SH
    # do not): the renders are told so and keep their list of non-empty brick rows (asurf_grid::links_epoch,
    # include/asurf.h).  The default measures a step that owes nothing to the step before it.
    g.links_epoch = 1 if args.links_epoch else 0
    r = _C.RaysSpec()
    r.origins, r.dirs = torch.from_numpy(o_np).to(dev), torch.from_numpy(d_np).to(dev)
    r.masks = torch.ones(Q, dtype=torch.bool, device=dev)
    opt = _C.RenderOptions()
    for k, v in synth.SYN_OPTS.items():
        setattr(opt, k, v)
    gt = torch.from_numpy(gt_np).to(dev)
    rgb = torch.zeros_like(gt)
    gr = _C.GridOutputGrads()
    # one flat allocation: with more than one rank the dense exchange is a single all-reduce over it
    arena = adist.GradArena(g.density_data.shape[0], g.sh_data.shape[1], dev)
    gr.grad_density_out, gr.grad_sh_out, gr.grad_surface_out = arena.grad_density, arena.grad_sh, arena.grad_surface
    gr.mask_out = arena.mask
    F = synth.SYN_FUSED
    M = F["l_dist_max_sample"]
    Q_global = Q * world

    r_in, gt_in, rgb_user = r, gt, rgb
    if args.sort:
        r = _C.RaysSpec()
        r.masks = r_in.masks
        rgb_user = torch.empty_like(rgb)

    def step(counters=None, exchange=True):
        nonlocal gt
        if args.sort:  # order, gather, render, scatter back: all inside the timed step
            perm = _C.ray_order(r_in.origins, r_in.dirs).long()
            r.origins, r.dirs = r_in.origins[perm], r_in.dirs[perm]
            gt = gt_in[perm]
        _step(counters, exchange)
        if args.sort:
            rgb_user[perm] = rgb

    def _step(counters=None, exchange=True):
        if world > 1:
            # the exchange sums whatever the buffers hold, so they must hold this step only
            # (in training the optimizer kernels zero them, optim_kernel.cu:24)
            arena.zero_()
        else:
            arena.zero_mask()  # the reference allocates a fresh sparse_grad_indexer per call (svox2.py:3535)
        _C.volume_render_surf_trav_fused(
            g, r, opt, gt, 0.0, 0.0, 0.0, True, False, F["lambda_l2"], F["lambda_l1"], 0.0,
            F["lambda_l_entropy"], False, 0.0, 0.0, 0.0, 0.0, F["l_di_alpha_thresh"],
            F["surf_sparse_alpha_thresh"], 0.0, 0.0, F["lambda_conv_mode_samp"], M, rgb, gr,
            global_n_rays=Q_global, counters=counters)
        if world > 1 and exchange:
            # data-parallel exchange: sum of per-shard gradients == gradient of the global batch.
            # A view touches ~20 % of the rows here, so the dense ring all-reduce is chosen
            # (alphasurf_amd/dist.py switches to the sparse all-to-all below 1/8).
            if args.exchange == "dense":
                arena.all_reduce()
            else:
                adist.all_reduce_grads(gr.grad_sh_out, gr.grad_density_out, gr.grad_surface_out, gr.mask_out,
                                       mode=args.exchange)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    timed_step = step
    if args.graph:
        # the call is stream-ordered (one memset + ten launches, no host synchronisation): capturable as it stands
        assert world == 1 and not args.sort, "--graph: single GPU, unsorted rays"
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(graph):
            step()
        timed_step = graph.replay
        timed_step()
        barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        timed_step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        adist._reduce(tt, dist.ReduceOp.MAX, None)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = Q_global / (dt / args.steps) / 1e6

    out = None
    if rank == 0:
        # ---- counters (V_t, V_a, S) and per-kernel HIP-event times, outside the timed region ----
        cnt = torch.zeros(4, dtype=torch.int64, device=dev)
        step(cnt, exchange=False)   # rank 0 only: no collective here
        torch.cuda.synchronize()
        v_t, v_a, s_cnt, n_r = [int(x) for x in cnt.cpu().tolist()]
        _lib.profile_enable(True)
        t_mask, t_fwd, t_bwd = [], [], []
        spans = {}
        n_prof = max(5, min(args.steps, 20))
        for _ in range(n_prof):
            arena.zero_mask()
            _C.volume_render_surf_trav_fused(
                g, r, opt, gt, 0.0, 0.0, 0.0, True, False, F["lambda_l2"], F["lambda_l1"], 0.0,
                F["lambda_l_entropy"], False, 0.0, 0.0, 0.0, 0.0, F["l_di_alpha_thresh"],
                F["surf_sparse_alpha_thresh"], 0.0, 0.0, F["lambda_conv_mode_samp"], M, rgb, gr,
                global_n_rays=Q_global)
            a, b, c = _lib.profile_read()
            t_mask.append(a), t_fwd.append(b), t_bwd.append(c)
            for name, ms in _lib.profile_read_spans():
                spans[name] = spans.get(name, 0.0) + ms / n_prof
        _lib.profile_enable(False)
        t_mask, t_fwd, t_bwd = float(np.mean(t_mask)), float(np.mean(t_fwd)), float(np.mean(t_bwd))
        b_f, b_b = algorithmic_bytes(Q, v_t, v_a, s_cnt, D, True, M)
        shares = algorithmic_bytes_per_kernel(Q, v_t, v_a, s_cnt, D, True, M)
        shares.pop("k_walk" if "k_sweep" in spans else "k_sweep")   # one of the two runs
        assert abs(sum(shares.values()) - (b_f + b_b)) <= 1e-6 * (b_f + b_b)
        # the dominant kernel = the longest-running one of the call
        dom_name = max(spans, key=spans.get)
        dom_ms, dom_bytes = spans[dom_name], shares.get(dom_name, 0.0)
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        peak = 8000.0
        # real HBM traffic per launch from the PMC counters (profiles/traffic.json, collected with rocprofv3 --pmc in
        # separate passes): next to the model's bytes it shows what the kernels actually move
        traffic, traffic_call = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(dom_name, {}).get("hbm_bytes_per_launch")
                traffic_call = sum(tj[k]["hbm_bytes_per_launch"] for k in spans if k in tj)
            except Exception:
                traffic, traffic_call = None, None
        roofline = dict(bound="hbm", kernel=dom_name, achieved=round(achieved, 1), peak=peak, unit="GB/s",
                        frac=round(achieved / peak, 4), traffic=traffic,
                        traffic_GBps=(round(traffic / (dom_ms * 1e-3) / 1e9, 1) if traffic else None),
                        note="achieved = bytes of SURVEY.md 8(d)'s compulsory-traffic model of the REFERENCE algorithm "
                             "(8 link loads per traversed voxel, a second traversal in the backward pass) / time; "
                             "the kernels find the work voxels by bit tests and replay samples, so they move far "
                             "fewer real bytes (traffic, traffic_GBps) and frac exceeds 1",
                        algorithmic_bytes_per_launch=dom_bytes, avg_launch_ms=round(dom_ms, 4),
                        per_kernel_ms={k: round(v, 4) for k, v in spans.items()},
                        per_kernel_algorithmic_GBps={k: round(shares.get(k, 0.0) / (v * 1e-3) / 1e9, 1)
                                                     for k, v in spans.items() if v > 0},
                        groups_ms=dict(k_build_masks=round(t_mask, 4), forward=round(t_fwd, 4),
                                       backward=round(t_bwd, 4)),
                        whole_call=dict(algorithmic_bytes=b_f + b_b,
                                        achieved=round((b_f + b_b) / ((t_mask + t_fwd + t_bwd) * 1e-3) / 1e9, 1),
                                        frac=round((b_f + b_b) / ((t_mask + t_fwd + t_bwd) * 1e-3) / 1e9 / peak, 4),
                                        traffic=traffic_call,
                                        traffic_GBps=(round(traffic_call / ((t_mask + t_fwd + t_bwd) * 1e-3) / 1e9, 1)
                                                      if traffic_call else None),
                                        traffic_frac=(round(traffic_call / ((t_mask + t_fwd + t_bwd) * 1e-3) / 1e9 / peak, 4)
                                                      if traffic_call else None)))

        cpu = None
        if not args.no_cpu_baseline:
            # the oracle (CPU restatement) on a bounded, strided sample of the same rays
            from oracle import oracle as O
            n_cpu = min(args.cpu_rays, Q)
            idx = np.linspace(0, Q - 1, n_cpu).astype(np.int64)
            if on_device:
                sc = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in sc.items()}
            osc = O.make_scene(sc["links"], sc["density"], sc["surface"], sc["sh"], sc["level_set"],
                               sc["offset"], sc["scaling"], B, sc["fake_sample_std"],
                               sc["truncated_vol_render_a"])
            fused = dict(lambda_l2=F["lambda_l2"], lambda_l1=F["lambda_l1"],
                         lambda_l_entropy=F["lambda_l_entropy"],
                         lambda_conv_mode_samp=F["lambda_conv_mode_samp"], l_dist_max_sample=M,
                         l_di_alpha_thresh=F["l_di_alpha_thresh"],
                         surf_sparse_alpha_thresh=F["surf_sparse_alpha_thresh"], global_n_rays=Q_global)
            O.render_fused(osc, o_np[idx[:2000]], d_np[idx[:2000]], synth.SYN_OPTS, gt_np[idx[:2000]], fused)
            reps, tc = 0, time.perf_counter()
            while True:
                O.render_fused(osc, o_np[idx], d_np[idx], synth.SYN_OPTS, gt_np[idx], fused)
                reps += 1
                if time.perf_counter() - tc > 10.0 or reps >= 50:
                    break
            tcpu = (time.perf_counter() - tc) / reps
            cpu = dict(value=round(n_cpu / tcpu / 1e6, 4), unit="Mrays/s", cores=O.num_threads(), kind="port",
                       sample=f"{n_cpu} rays strided over the same {args.width}x{args.width} view, fused fwd+bwd, "
                              f"{reps} repetitions, OpenMP over rays")

        out = {
            "metric": "Mrays/s fwd+bwd, 800x800 on 256^3 SDF grid; achieved HBM GB/s vs peak",
            "value": round(value, 3), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{R}^3 sparse SDF grid ({g.density_data.shape[0]} active voxels, union of 16 "
                                   f"spheres), SH deg {int(round(B ** 0.5)) - 1} (basis_dim {B}), "
                                   f"{args.width}x{args.width} pinhole view per GPU "
                                   f"({'shuffled' if args.shuffle else 'image-order'} rays"
                                   f"{', ordered per step' if args.sort else ''}), "
                                   "volume_render_surf_trav_fused fwd+bwd, surface_cuda_syn.yaml options"
                                   + ("; links unchanged from step to step (links_epoch)" if args.links_epoch else ""),
                       "rays_per_step_per_gpu": Q, "active_voxels": int(g.density_data.shape[0]),
                       "avg_per_ray": {"V_t": round(v_t / Q, 2), "V_a": round(v_a / Q, 2), "S": round(s_cnt / Q, 3)},
                       "parallelism": f"ray-sharded dp{world}, {args.exchange} RCCL exchange of grads" if world > 1 else "single GPU"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
