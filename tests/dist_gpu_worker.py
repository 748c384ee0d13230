"""One rank of the two-rank rehearsal on ONE device (tests/test_dist_gpu.py): both ranks use cuda:0, the exchange goes over
gloo (device tensors staged through host memory by alphasurf_amd/dist.py) -- the control flow and the device-side kernels
of the exchange (asurf_rows_list / _pack / _unpack) are the ones an RCCL run uses.  Launched by tests/conftest.py through
torch.distributed.run BEFORE the pytest process touches the GPU; writes one JSON line per rank into the directory given
as argv[1]."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    out_dir = sys.argv[1]
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
    res = dict(rank=rank, world=world, cases={})
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        import hip_util as H
        from alphasurf_amd import csrc_api as _C, dist as adist, synth
        sc = synth.make_sparse_scene(64, 9, seed=3, trunc_a=4.0)
        o, d = synth.pinhole_rays(synth.orbit_camera(1), 96, 96)
        Q = o.shape[0]
        gt = np.random.RandomState(2).rand(Q, 3).astype(np.float32)
        opts = dict(sigma_thresh=-1e4, stop_thresh=-1e4, only_outward_intersect=True, truncated_vol_render=True,
                    trunc_vol_weight_min=1e-10, limited_fake_sample=True, fake_sample_l_dist=False)
        fused = dict(lambda_l2=1.0, lambda_l_entropy=1e-4, lambda_conv_mode_samp=1e-6)
        args = [fused.get(k, dflt) for k, dflt in H.FUSED_ORDER]
        g, op = H.grid_spec(sc), H.opt_spec(opts)
        n, D = g.density_data.shape[0], g.sh_data.shape[1]

        def render(lo, hi, arena):
            r = H.rays_spec(o[lo:hi], d[lo:hi])
            gr = _C.GridOutputGrads()
            gr.grad_density_out, gr.grad_sh_out, gr.grad_surface_out = arena.grad_density, arena.grad_sh, arena.grad_surface
            gr.grad_fake_sample_std_out = arena.grad_fake_sample_std
            gr.mask_out = arena.mask
            rgb = torch.zeros(hi - lo, 3, device=dev)
            _C.volume_render_surf_trav_fused(g, r, op, torch.from_numpy(gt[lo:hi]).to(dev), *args, rgb, gr, global_n_rays=Q)
            return rgb

        # the single-batch answer (every rank computes it for itself)
        ref = adist.GradArena(n, D, dev)
        rgb_ref = render(0, Q, ref)
        lo, hi = adist.shard_rays(Q, rank, world)
        for mode in ("dense", "sparse", "auto"):
            arena = adist.GradArena(n, D, dev)
            rgb = render(lo, hi, arena)
            used = arena.all_reduce(mode=mode)
            torch.cuda.synchronize()
            case = dict(used=used, rgb_equal=bool(torch.equal(rgb, rgb_ref[lo:hi])),
                        masks_equal=bool(torch.equal(arena.mask, ref.mask)), touched=float(ref.mask.float().mean()))
            for name in ("grad_sh", "grad_density", "grad_surface", "grad_fake_sample_std"):
                a, b = getattr(arena, name), getattr(ref, name)
                scale = max(float(b.abs().max()), 1e-30)
                case[name] = float((a - b).abs().max()) / scale
            res["cases"][mode] = case
        # the step taken in two parts, the first part's rows exchanged "under" the second part's render
        # (dist.PipelinedReduce; over gloo the collectives are synchronous, the arithmetic is the same)
        mid = (lo + hi) // 2
        px = adist.PipelinedReduce(dev)
        a0, a1 = adist.GradArena(n, D, dev), adist.GradArena(n, D, dev)
        rgb0 = render(lo, mid, a0)
        px.submit(a0)
        rgb1 = render(mid, hi, a1)
        px.progress()
        px.submit(a1)
        used = px.finish(a0)
        torch.cuda.synchronize()
        rgb = torch.cat([rgb0, rgb1])
        case = dict(used=used, rgb_equal=bool(torch.equal(rgb, rgb_ref[lo:hi])),
                    masks_equal=bool(torch.equal(a0.mask, ref.mask)), touched=float(ref.mask.float().mean()))
        for name in ("grad_sh", "grad_density", "grad_surface", "grad_fake_sample_std"):
            a, b = getattr(a0, name), getattr(ref, name)
            scale = max(float(b.abs().max()), 1e-30)
            case[name] = float((a - b).abs().max()) / scale
        res["cases"]["pipelined"] = case
        res["ok"] = True
    except Exception as e:  # noqa: BLE001 -- reported to the test
        import traceback
        res["ok"] = False
        res["error"] = f"{e!r}\n{traceback.format_exc()}"
    finally:
        with open(os.path.join(out_dir, f"rank{rank}.json"), "w") as f:
            json.dump(res, f)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
