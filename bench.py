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
import itertools
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
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--reso", type=int, default=256)
    ap.add_argument("--basis-dim", type=int, default=9)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--band", type=float, default=3.0, help="half-thickness (voxels) of the active shell")
    ap.add_argument("--shuffle", action="store_true", help="training-order (shuffled) rays")
    ap.add_argument("--sort", action="store_true",
                    help="order the batch for coherence inside the timed step (asurf_ray_order), scatter rgb back")
    ap.add_argument("--order-rays", type=int, default=0, choices=[0, 1, 2],
                    help="asurf_fused_params::order_rays: 0 = the library decides (orders a batch of >= 65 536 rays inside "
                         "the call when the last batch arrived shuffled), 1 = never, 2 = always")
    ap.add_argument("--eikonal", action="store_true",
                    help="BASELINE config 3 as stated: the step also takes the eikonal loss's gradient over all non-empty "
                         "cells (surface_eikonal_grad_sparse; opt/opt.py:1002-1009)")
    ap.add_argument("--no-also", action="store_true",
                    help="the headline run (default flags, one GPU) also measures the training script's batch size, a whole "
                         "training iteration and the two heavy scenes as child processes and reports them in `also`; "
                         "this switches that off")
    ap.add_argument("--no-image-width", action="store_true",
                    help="do not tell the call that the batch is an image (asurf_fused_params::image_width = 0)")
    ap.add_argument("--heavy", nargs="?", const="levels", default=None, choices=["levels", "fake"],
                    help="a heavier variant of the workload (never the driver's default line).  levels: three level sets "
                         "(-1, 0, 1 voxel), both crossings of each -- about 4 samples a ray instead of 0.7; fake: three level "
                         "sets, outward crossings, limited fake samples on -- about 11")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the self-check (a strided sample of the timed batch against the CPU oracle; a mismatch "
                         "makes bench.py exit non-zero)")
    ap.add_argument("--verify-rays", type=int, default=2000)
    ap.add_argument("--counters", default=None,
                    help="PMC counters file (default profiles/counters.json; variants/pmc_profile.sh regenerates it)")
    ap.add_argument("--links-epoch", action="store_true",
                    help="tell the renders that `links` is the same from step to step, as it is in the training loop "
                         "(asurf_grid::links_epoch): the occupancy pass then reads the non-empty brick rows only.  Off by "
                         "default: every timed step does all of its work from the inputs alone")
    ap.add_argument("--graph", action="store_true",
                    help="capture one step in a HIP graph after the warm-up and replay it in the timed region")
    ap.add_argument("--cpu-rays", type=int, default=40000)
    ap.add_argument("--exchange", default=os.environ.get("ASURF_EXCHANGE", "pipelined"),
                    choices=["dense", "sparse", "auto", "pipelined"],
                    help="gradient exchange for --gpus > 1 (alphasurf_amd/dist.py).  pipelined: the rank's view is rendered as "
                         "two half batches, the first half's touched rows are exchanged while the second half renders "
                         "(dist.PipelinedReduce); the others exchange after the whole view: one ring all-reduce of all "
                         "buffers (dense), of the union of the touched rows (sparse), or whichever moves fewer bytes (auto)")
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


HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec; 6.3 TB/s is what a streaming copy reaches)
ATOMIC_PEAK_GBPS = 1300.0       # ... "Global float atomics": ~1.3 TB/s of added bytes chip-wide (aligned 64-byte pieces)
ATOMIC_PEAK_GREQ = 21.0         # G atomic requests/s at the L2, one per 64-byte line an instruction touches, whatever the lanes
                                # in it (variants/micro/atomics.hip, profiles/r4_micro_atomics.txt: scattered floats, rows of
                                # 8 / 16 / 27 / 32 / 64 floats all run at 20.7-21.2 G lines/s; 21 G x 64 B = the 1.3 TB/s above)
N_SIMD, CLOCK_HZ, VALU_CYCLES = 1024, 2.4e9, 2   # 256 CUs x 4 SIMD-32; a wave64 VALU instruction holds its SIMD 2 cycles
ATOMIC_KERNELS = ("k_bwd_sh", "k_bwd_geo", "k_backward_fallback")
# What a wave64 vector instruction of each kernel costs its SIMD ON AVERAGE, by the kernel's static instruction mix
# (variants/isa_mix.py over the library's ISA) priced with the per-class costs variants/micro/issue_mix.hip measured
# (profiles/r4_micro_issue_mix.txt: f32 fma/add/mul, integer add/sub, two-operand bit-wise logic and moves 2.5-3.0
# cycles; shifts, bit fields, compares, selects, min/max, three-operand integer forms, integer multiplies, conversions,
# readlane 4.2-4.7; f64 4.9-5.0; rcp/sqrt 8.3-8.6).  `valu_mix_frac` prices a
# kernel's instructions with ITS figure; `valu_issue_frac` -- the one `frac` and `bound` use -- with the 2 cycles of
# the fma rate, which no kernel of this path can reach.
VALU_MIX_CYCLES = {"k_build_masks2": 4.04, "k_sweep": 3.87, "k_geom": 4.23, "k_comp": 3.87, "k_bwd_seq": 3.95,
                   "k_bwd_sh": 3.94, "k_bwd_geo": 4.47}


def config_key(a):
    """What identifies the workload of a run (the counters of profiles/counters.json are collected per workload)."""
    image_width = 0 if (a.shuffle or a.no_image_width) else a.width
    return dict(reso=a.reso, basis_dim=a.basis_dim, width=a.width, band=a.band, shuffle=bool(a.shuffle),
                sort=bool(a.sort), links_epoch=bool(a.links_epoch), rays=a.width * a.width, heavy=a.heavy,
                order_rays=a.order_rays, image_width=image_width, eikonal=bool(getattr(a, "eikonal", False)))


def load_counters(path, cfg_key):
    """profiles/counters.json: per-kernel PMC counters of ONE configuration (collected by variants/pmc_profile.sh:
    rocprofv3 --pmc passes over this very script).  Counters of another configuration say nothing about this run:
    then None."""
    try:
        cj = json.load(open(path))
    except Exception:
        return None
    for one in ([cj] + list(cj.get("sets", []))):     # (the headline's set, then the sets of other workloads)
        if one.get("config") == cfg_key:
            return one.get("kernels")
    return None


def kernel_ceilings(ms, c, name=None):
    """Utilisation of the three ceilings this path can meet, each a fraction <= 1 of a hardware rate, from the
    kernel's measured time and its PMC counters: HBM (FETCH_SIZE doubled + WRITE_SIZE, the guide's gfx950
    correction), VALU issue (SQ_INSTS_VALU x 2 cycles over 1024 SIMDs at 2.4 GHz) and, for the scatter kernels, float
    atomics (WRITE_SIZE bytes against ~1.3 TB/s of added bytes)."""
    t = ms * 1e-3
    out = dict(ms=round(ms, 4), hbm_bytes=None, hbm_GBps=None, hbm_frac=None, valu_issue_frac=None, atomic_frac=None,
               wait_frac=None, bound=None)
    if not c or t <= 0:
        return out
    if c.get("hbm_bytes") is not None:
        out["hbm_bytes"] = int(c["hbm_bytes"])
        out["hbm_GBps"] = round(c["hbm_bytes"] / t / 1e9, 1)
        out["hbm_frac"] = round(c["hbm_bytes"] / t / 1e9 / HBM_PEAK_GBPS, 4)
    if c.get("valu_insts") is not None:
        out["valu_issue_frac"] = round(c["valu_insts"] * VALU_CYCLES / (N_SIMD * CLOCK_HZ * t), 4)
        if name in VALU_MIX_CYCLES:
            out["valu_mix_frac"] = round(c["valu_insts"] * VALU_MIX_CYCLES[name] / (N_SIMD * CLOCK_HZ * t), 4)
    if c.get("atomic_requests") is not None and c.get("atomic_bytes") is not None:   # (the scatter kernels only)
        out["atomic_Greq_s"] = round(c["atomic_requests"] / t / 1e9, 2)
        out["atomic_frac"] = round(c["atomic_requests"] / t / 1e9 / ATOMIC_PEAK_GREQ, 4)
    elif c.get("atomic_bytes") is not None:   # (counters collected before round 4: bytes written against 1.3 TB/s -- an
        out["atomic_frac"] = round(c["atomic_bytes"] / t / 1e9 / ATOMIC_PEAK_GBPS, 4)   # underestimate, L2 merges writes)
    if c.get("wave_cycles"):
        out["wait_frac"] = round(c.get("wait_any", 0.0) / c["wave_cycles"], 4)
    fr = {k: out[k + "_frac"] for k in ("hbm", "valu_issue", "atomic") if out[k + "_frac"] is not None}
    if fr:
        out["bound"] = max(fr, key=fr.get)
    return out


def build_roofline(spans, shares, dom_name, call_ms, model_bytes, cfg_key, counters_path):
    """The `roofline` object of the JSON line.  `frac` is a UTILISATION of the ceiling that binds the dominant kernel
    (per-kernel table in `ceilings`); SURVEY.md 8(d)'s compulsory-traffic model of the reference algorithm is kept
    beside it as `reference_model` -- a throughput equivalent, not a utilisation (it exceeds 1: the kernels find
    their work voxels by bit tests and replay samples instead of walking twice)."""
    counters = load_counters(counters_path, cfg_key)
    ceil = {}
    for name, ms in spans.items():
        # (the fallback kernels -- 1 % of the call -- share their names with bench.py's own counter pass over all rays:
        # their averaged counters describe neither)
        c = None if name.endswith("_fallback") else (counters or {}).get(name)
        if c is not None:
            c = dict(c)
            c["atomic_bytes"] = c.get("write_bytes") if name in ATOMIC_KERNELS else None
        ceil[name] = kernel_ceilings(ms, c, name)
    dom = ceil[dom_name]
    dom_ms = spans[dom_name]
    model_dom = shares.get(dom_name, 0.0)
    ref_model = dict(
        note="SURVEY.md 8(d): compulsory bytes of the REFERENCE algorithm (8 link loads per traversed voxel, a second "
             "traversal in the backward pass) / measured time; a throughput equivalent, not a utilisation",
        kernel=dom_name, algorithmic_bytes_per_launch=model_dom,
        achieved=round(model_dom / (dom_ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
        frac=round(model_dom / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
        per_kernel_GBps={k: round(shares.get(k, 0.0) / (v * 1e-3) / 1e9, 1) for k, v in spans.items() if v > 0},
        whole_call=dict(algorithmic_bytes=model_bytes, achieved=round(model_bytes / (call_ms * 1e-3) / 1e9, 1),
                        frac=round(model_bytes / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)))
    if counters is None or dom["bound"] is None:
        # no counters for this configuration (or none for its dominant kernel): nothing measured to price a ceiling with
        return dict(bound="hbm", kernel=dom_name, achieved=None, peak=HBM_PEAK_GBPS, unit="GB/s", frac=None, traffic=None,
                    avg_launch_ms=round(dom_ms, 4), per_kernel_ms={k: round(v, 4) for k, v in spans.items()},
                    note="no PMC counters on file for this configuration (variants/pmc_profile.sh collects them)",
                    reference_model=ref_model)
    whole = dict(ms=round(call_ms, 4))
    tot_hbm = sum(v["hbm_bytes"] or 0 for v in ceil.values())
    tot_valu = sum(((counters.get(k) or {}).get("valu_insts") or 0.0) for k in spans)
    whole["hbm_bytes"] = int(tot_hbm)
    whole["hbm_GBps"] = round(tot_hbm / (call_ms * 1e-3) / 1e9, 1)
    whole["hbm_frac"] = round(tot_hbm / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)
    whole["valu_issue_frac"] = round(tot_valu * VALU_CYCLES / (N_SIMD * CLOCK_HZ * call_ms * 1e-3), 4)
    bound = dom["bound"] or "hbm"
    if bound == "hbm":
        achieved, peak, unit = dom["hbm_GBps"], HBM_PEAK_GBPS, "GB/s"
    elif bound == "atomic":
        if dom.get("atomic_Greq_s") is not None:
            achieved, peak, unit = dom["atomic_Greq_s"], ATOMIC_PEAK_GREQ, "G atomic line requests/s (TCC_ATOMIC)"
        else:
            achieved, peak, unit = round(dom["atomic_frac"] * ATOMIC_PEAK_GBPS, 1), ATOMIC_PEAK_GBPS, "GB/s (float atomics)"
    else:
        peak = N_SIMD * CLOCK_HZ / VALU_CYCLES / 1e9
        achieved, unit = round(dom["valu_issue_frac"] * peak, 1), "G wave-instructions/s (VALU issue)"
    return dict(bound=bound, kernel=dom_name, achieved=achieved, peak=round(peak, 1), unit=unit,
                frac=round(achieved / peak, 4), traffic=dom["hbm_bytes"], hbm_frac=dom["hbm_frac"],
                avg_launch_ms=round(dom_ms, 4), per_kernel_ms={k: round(v, 4) for k, v in spans.items()},
                ceilings=ceil, whole_call=whole, counters_file=os.path.relpath(counters_path, ROOT),
                note="frac = utilisation of the ceiling that binds the dominant kernel; ceilings: every kernel against "
                     "HBM bytes (PMC FETCH_SIZE x 2 + WRITE_SIZE), VALU issue (SQ_INSTS_VALU x 2 cycles / 1024 SIMDs / "
                     "2.4 GHz -- the fma rate; valu_mix_frac prices the same count with the kernel's own mix, variants/isa_mix.py) and, for the scatter kernels, atomic requests at the L2 (TCC_ATOMIC_sum against the 21 G 64-byte "
                     "line requests/s variants/micro/atomics.hip reaches); wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES",
                reference_model=ref_model)


def also_lines(args):
    """The configurations VERDICT r3 asked to see beside the headline, measured by THIS run so that the driver times
    them: each is a child process (this script again, or variants/bench_iteration.py) that checks its own batch against
    the oracle; a failed check or a crashed child is reported as such (`ok: false`), never dropped."""
    import subprocess
    here = os.path.abspath(__file__)
    it_py = os.path.join(ROOT, "variants", "bench_iteration.py")
    common = ["--no-cpu-baseline", "--no-also"]
    jobs = [
        ("rays_5184_shuffled", "opt/opt.py's batch (config_util.py:340: 5000 rays; 72 x 72 here), shuffled, links_epoch = 0",
         [here, "--width", "72", "--shuffle", "--steps", "200", "--warmup", "20"] + common),
        ("rays_5184_shuffled_links_epoch", "the same with the caller's promise that `links` is unchanged (what the SparseGrid mirror passes)",
         [here, "--width", "72", "--shuffle", "--links-epoch", "--steps", "200", "--warmup", "20"] + common),
        ("training_iteration_5184", "one iteration of surface_cuda_syn.yaml through the SparseGrid mirror at that batch size: fused "
         "render, TV on alpha, surface TV, normal loss, alpha sparsify, three RMSprop steps (variants/bench_iteration.py)",
         [it_py, "--rays", "5184", "--verify"]),
        ("heavy_levels", "config 2 with three level sets (-1, 0, 1), both crossings: 4.2 samples a ray",
         [here, "--heavy", "levels", "--steps", "10", "--warmup", "3"] + common),
        ("heavy_fake", "config 2 with three level sets and limited fake samples: 11 samples a ray",
         [here, "--heavy", "fake", "--steps", "10", "--warmup", "3"] + common),
    ]
    res = {}
    for name, what, cmd in jobs:
        t0 = time.perf_counter()
        entry = dict(what=what)
        try:
            p = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=240, cwd=ROOT)
            line = [ln for ln in p.stdout.strip().splitlines() if ln.startswith("{")]
            d = json.loads(line[-1]) if line else None
            if p.returncode != 0 or d is None:
                entry.update(ok=False, rc=p.returncode, stderr_tail=p.stderr.strip().splitlines()[-3:])
            elif "ms_per_iteration" in d:
                entry.update(ok=bool((d.get("verify") or {}).get("ok", False)), rays=d["rays"],
                             ms_per_iteration=d["ms_per_iteration"], wall_ms_per_iteration=d["wall_ms_per_iteration"],
                             Mrays_per_s=d["Mrays_per_s"], verify=d.get("verify"))
            else:
                entry.update(ok=bool((d.get("verify") or {}).get("ok", False)), value=d["value"], unit=d["unit"],
                             ms_per_step=d["ms_per_step"], rays_per_step=d["config"]["rays_per_step_per_gpu"],
                             avg_per_ray=d["config"]["avg_per_ray"], per_kernel_ms=d["roofline"]["per_kernel_ms"],
                             verify=d.get("verify"))
        except Exception as e:  # noqa: BLE001
            entry.update(ok=False, error=repr(e))
        entry["child_s"] = round(time.perf_counter() - t0, 1)
        res[name] = entry
    return res


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
        import datetime
        dist.init_process_group(backend, timeout=datetime.timedelta(seconds=int(os.environ.get("ASURF_DIST_TIMEOUT", "300"))))
    dev = torch.device("cuda", local_rank if world > 1 else 0)
    torch.cuda.set_device(dev)

    from alphasurf_amd import csrc_api as _C, dist as adist, lib as _lib, synth

    # ---- synthetic workload (seeded; identical grid on every rank) ----
    R, B = args.reso, args.basis_dim
    D = 3 * B
    on_device = R > 256 or args.band != 3.0   # big grids: build the scene with torch on the GPU
    scene_kw = dict(synth.BENCH_SCENE)
    render_opts = dict(synth.SYN_OPTS)
    if args.heavy:
        scene_kw["level_set"] = (-1.0, 0.0, 1.0)
        if args.heavy == "levels":
            render_opts.update(only_outward_intersect=False)
        else:
            render_opts.update(only_outward_intersect=True, surf_fake_sample=True, limited_fake_sample=True,
                               surf_fake_sample_min_vox_len=0.1)
    if on_device:
        sc = synth.make_sparse_scene_torch(R, B, dev, band=args.band, trunc_a=5.0, **scene_kw)
    else:
        sc = synth.make_sparse_scene(R, B, trunc_a=5.0, **scene_kw)
    o_np, d_np = synth.pinhole_rays(synth.orbit_camera(rank % 8), args.width, args.width)
    if args.shuffle:
        perm = np.random.RandomState(11 + rank).permutation(o_np.shape[0])
        o_np, d_np = o_np[perm], d_np[perm]
    Q = o_np.shape[0]
    gt_np = np.random.RandomState(7 + rank).rand(Q, 3).astype(np.float32)
    # an image in image order is handed over as one (asurf_fused_params::image_width: large batches are then rendered in
    # the order of their 8 x 8 pixel tiles inside the call); a shuffled batch is not an image
    image_width = 0 if (args.shuffle or args.no_image_width) else args.width

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
    for k, v in render_opts.items():
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

    eik_cells = torch.nonzero(g.links.view(-1) >= 0).view(-1).to(torch.int32).contiguous() if args.eikonal else None
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

    # --exchange pipelined: the two half batches, an arena of its own for the second one (the first half's rows are being
    # packed while the second half's atomics run), a value of links_epoch that changes every step -- the first half's
    # occupancy pass is the full one, the second half's is told that `links` has not changed SINCE THE FIRST HALF
    pipelined = world > 1 and args.exchange == "pipelined" and not args.sort
    if pipelined:
        h = Q // 2
        halves = []
        for lo, hi in ((0, h), (h, Q)):
            rh = _C.RaysSpec()
            rh.origins, rh.dirs, rh.masks = r.origins[lo:hi], r.dirs[lo:hi], r.masks[lo:hi]
            halves.append((rh, gt[lo:hi], rgb[lo:hi]))
        arena_b = adist.GradArena(g.density_data.shape[0], g.sh_data.shape[1], dev)
        gr_b = _C.GridOutputGrads()
        gr_b.grad_density_out, gr_b.grad_sh_out, gr_b.grad_surface_out = arena_b.grad_density, arena_b.grad_sh, arena_b.grad_surface
        gr_b.mask_out = arena_b.mask
        px = adist.PipelinedReduce(dev, timed=True)
        step_epochs = itertools.count(1000)

    def _render(rays, target, out, grads, counters=None):
        _C.volume_render_surf_trav_fused(
            g, rays, opt, target, 0.0, 0.0, 0.0, True, False, F["lambda_l2"], F["lambda_l1"], 0.0,
            F["lambda_l_entropy"], False, 0.0, 0.0, 0.0, 0.0, F["l_di_alpha_thresh"],
            F["surf_sparse_alpha_thresh"], 0.0, 0.0, F["lambda_conv_mode_samp"], M, out, grads,
            global_n_rays=Q_global, counters=counters, order_rays=args.order_rays, image_width=image_width)

    def _step_pipelined():
        arena.zero_()
        arena_b.zero_()
        if not args.links_epoch:
            g.links_epoch = next(step_epochs)
        _render(halves[0][0], halves[0][1], halves[0][2], gr)
        px.submit(arena)
        _render(halves[1][0], halves[1][1], halves[1][2], gr_b)
        px.progress()          # (the second half is queued: the host may wait for the first half's row count now)
        px.submit(arena_b)
        px.finish(arena)
        if not args.links_epoch:
            g.links_epoch = 0

    def _step(counters=None, exchange=True):
        if pipelined and exchange and counters is None:
            return _step_pipelined()
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
            global_n_rays=Q_global, counters=counters, order_rays=args.order_rays, image_width=image_width)
        if args.eikonal:
            _C.surface_eikonal_grad_sparse(g.links, g.surface_data, eik_cells, None, 1e-4, 2.0 / R, arena.grad_surface)
        if world > 1 and exchange:
            # data-parallel exchange: sum of per-shard gradients == gradient of the global batch (alphasurf_amd/dist.py:
            # "dense" = one ring all-reduce of the whole arena, "sparse" = the rows of the union of the ranks' masks
            # only, "auto" = sparse unless the union exceeds half of the rows)
            xev[0].record()
            xmode[0] = arena.all_reduce(mode="auto" if args.exchange == "pipelined" else args.exchange)
            xev[1].record()

    xev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
    xmode = [None]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # (VERDICT r3 #7c: which exchange runs is decided HERE, by --exchange / ASURF_EXCHANGE, identically on every rank and
    # before the first collective; a collective that fails raises on its rank, the process group's timeout ends the
    # others, and the job exits non-zero -- nothing retries on a communicator that has failed)
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

    exchange = None
    if world > 1:
        # ---- one more exchanged step, outside the timed region: what the exchange costs, and whether it is right ----
        step()
        barrier()
        if pipelined:
            x_ms, x_bytes, x_mode = px.exchange_ms(), px.last_bytes, "pipelined"
        else:
            x_ms = xev[0].elapsed_time(xev[1])
            x_mode = xmode[0]
            n_rows_, W_ = g.density_data.shape[0], g.sh_data.shape[1] + 2
            x_bytes = arena.flat.numel() * 4 if x_mode == "dense" else int(arena.mask.sum().item()) * W_ * 4 + n_rows_
        tx = torch.tensor([x_ms], device=dev, dtype=torch.float64)
        adist._reduce(tx, dist.ReduceOp.MAX, None)
        exchange = dict(mode=x_mode, exchange_ms=round(float(tx.item()), 4), bytes_per_rank=int(x_bytes),
                        note="device time of the step's collectives and their list / pack kernels (max over ranks; pipelined: "
                             "on the side stream, partly hidden behind the second half's render); bytes a rank hands to them")
        if not args.no_verify:
            # the arena now holds the sum over the ranks.  Rank 0 renders EVERY rank's rays alone, with the same global
            # normaliser, into a fresh arena: sum of the shards' gradients == gradient of the whole batch
            if rank == 0:
                ref = adist.GradArena(g.density_data.shape[0], g.sh_data.shape[1], dev)
                gref = _C.GridOutputGrads()
                gref.grad_density_out, gref.grad_sh_out, gref.grad_surface_out = ref.grad_density, ref.grad_sh, ref.grad_surface
                gref.mask_out = ref.mask
                for rk in range(world):
                    o_k, d_k = synth.pinhole_rays(synth.orbit_camera(rk % 8), args.width, args.width)
                    if args.shuffle:
                        pk = np.random.RandomState(11 + rk).permutation(o_k.shape[0])
                        o_k, d_k = o_k[pk], d_k[pk]
                    gt_k = torch.from_numpy(np.random.RandomState(7 + rk).rand(Q, 3).astype(np.float32)).to(dev)
                    r_k = _C.RaysSpec()
                    r_k.origins, r_k.dirs = torch.from_numpy(o_k).to(dev), torch.from_numpy(d_k).to(dev)
                    r_k.masks = torch.ones(Q, dtype=torch.bool, device=dev)
                    _render(r_k, gt_k, torch.zeros_like(gt_k), gref)
                torch.cuda.synchronize()
                e_x = {}
                for k, a_, b_ in (("sh", arena.grad_sh, ref.grad_sh), ("density", arena.grad_density, ref.grad_density),
                                  ("surface", arena.grad_surface, ref.grad_surface)):
                    e_x[k] = float((a_ - b_).abs().max().item()) / max(float(b_.abs().max().item()), 1e-30)
                m_ok = bool((arena.mask == ref.mask).all().item())
                x_ok = max(e_x.values()) <= 2e-4 and m_ok and bool(ref.mask.any().item())
                exchange["verify"] = dict(ok=x_ok, grad_rel={k: round(v, 8) for k, v in e_x.items()}, masks_equal=m_ok,
                                          rows_touched=int(ref.mask.sum().item()), bound=2e-4,
                                          checker=f"rank 0 alone: the {world} ranks' views rendered one after the other "
                                                  "with the global normaliser, against the exchanged arena")
                del ref, gref
        barrier()

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
                global_n_rays=Q_global, order_rays=args.order_rays, image_width=image_width)
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
        call_ms = t_mask + t_fwd + t_bwd
        cfg_key = config_key(args)
        assert cfg_key["rays"] == Q
        roofline = build_roofline(spans, shares, dom_name, call_ms, b_f + b_b, cfg_key,
                                  args.counters or os.path.join(ROOT, "profiles", "counters.json"))
        roofline["groups_ms"] = dict(k_build_masks=round(t_mask, 4), forward=round(t_fwd, 4), backward=round(t_bwd, 4))

        cpu, verify = None, None
        need_oracle = not (args.no_cpu_baseline and args.no_verify)
        if need_oracle:
            # the oracle (CPU restatement, test infrastructure): the checker of the self-check and the CPU baseline
            from oracle import oracle as O
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
        if not args.no_verify:
            # ---- self-check, outside the timed region: a strided sample of the TIMED batch against the oracle ----
            # (1) the colours the timed steps left for those rays; (2) the same rays rendered on their own through
            # the same entry point, so that their gradients and masks can be compared row by row
            n_v = min(args.verify_rays, Q)
            vidx = np.linspace(0, Q - 1, n_v).astype(np.int64)
            rgb_o, go, st_o = O.render_fused(osc, o_np[vidx], d_np[vidx], render_opts, gt_np[vidx], fused)
            rgb_timed = (rgb_user if args.sort else rgb)[torch.from_numpy(vidx).to(dev)].cpu().numpy()
            e_timed = float(np.abs(rgb_timed - rgb_o).max())
            rv = _C.RaysSpec()
            vt = torch.from_numpy(vidx).to(dev)
            rv.origins, rv.dirs = r_in.origins[vt].contiguous(), r_in.dirs[vt].contiguous()
            rv.masks = torch.ones(n_v, dtype=torch.bool, device=dev)
            gv = _C.GridOutputGrads()
            gv.grad_density_out = torch.zeros_like(g.density_data)
            gv.grad_sh_out = torch.zeros_like(g.sh_data)
            gv.grad_surface_out = torch.zeros_like(g.surface_data)
            gv.mask_out = torch.zeros(g.density_data.shape[0], dtype=torch.bool, device=dev)
            rgb_v = torch.zeros(n_v, 3, device=dev)
            _C.volume_render_surf_trav_fused(
                g, rv, opt, gt_in[vt].contiguous(), 0.0, 0.0, 0.0, True, False, F["lambda_l2"], F["lambda_l1"], 0.0,
                F["lambda_l_entropy"], False, 0.0, 0.0, 0.0, 0.0, F["l_di_alpha_thresh"],
                F["surf_sparse_alpha_thresh"], 0.0, 0.0, F["lambda_conv_mode_samp"], M, rgb_v, gv,
                global_n_rays=Q_global)
            torch.cuda.synchronize()
            e_rgb = float(np.abs(rgb_v.cpu().numpy() - rgb_o).max())
            e_grad = {}
            for k, t in (("sh", gv.grad_sh_out), ("density", gv.grad_density_out), ("surface", gv.grad_surface_out)):
                ref = go[k]
                e_grad[k] = float(np.abs(t.cpu().numpy().reshape(ref.shape) - ref).max()) / max(float(np.abs(ref).max()), 1e-30)
            mask_ok = bool((gv.mask_out.cpu().numpy().astype(bool) == go["mask"].astype(bool)).all())
            ok = (e_timed <= 2e-5 and e_rgb <= 2e-5 and max(e_grad.values()) <= 2e-4 and mask_ok and
                  int(st_o["n_samples"].sum()) > 0)
            verify = dict(ok=ok, rays=n_v, samples=int(st_o["n_samples"].sum()), rgb_linf_timed_batch=e_timed,
                          rgb_linf=e_rgb, grad_rel={k: round(v, 8) for k, v in e_grad.items()}, masks_equal=mask_ok,
                          bounds=dict(rgb=2e-5, grad_rel=2e-4), checker="oracle (CPU restatement), outside the timed region")
        if not args.no_cpu_baseline:
            # the oracle on a bounded, strided sample of the same rays
            n_cpu = min(args.cpu_rays, Q)
            idx = np.linspace(0, Q - 1, n_cpu).astype(np.int64)
            O.render_fused(osc, o_np[idx[:2000]], d_np[idx[:2000]], render_opts, gt_np[idx[:2000]], fused)
            reps, tc = 0, time.perf_counter()
            while True:
                O.render_fused(osc, o_np[idx], d_np[idx], render_opts, gt_np[idx], fused)
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
                                   + ({"levels": " -- HEAVY variant: level sets (-1, 0, 1), both crossings of each",
                                       "fake": " -- HEAVY variant: level sets (-1, 0, 1), limited fake samples on"}[args.heavy]
                                      if args.heavy else "")
                                   + ("; links unchanged from step to step (links_epoch)" if args.links_epoch else "")
                                   + ("; + eikonal loss gradient over all non-empty cells in every step" if args.eikonal else ""),
                       "rays_per_step_per_gpu": Q, "active_voxels": int(g.density_data.shape[0]),
                       "avg_per_ray": {"V_t": round(v_t / Q, 2), "V_a": round(v_a / Q, 2), "S": round(s_cnt / Q, 3),
                                       "rays_with_a_sample": round(float(((rgb_user if args.sort else rgb) != 1.0)
                                                                         .any(dim=1).float().mean().item()), 4)},
                       "parallelism": (f"ray-sharded dp{world}, {args.exchange} exchange of grads over "
                                       f"{'RCCL' if os.environ.get('ASURF_DIST_BACKEND', 'nccl') == 'nccl' else 'gloo (rehearsal: ranks share devices)'}")
                       if world > 1 else "single GPU"},
            "roofline": roofline, "cpu_baseline": cpu, "verify": verify,
        }
        if exchange is not None:
            out["exchange"] = exchange
        headline = (world == 1 and R == 256 and B == 9 and args.width == 800 and args.band == 3.0 and not args.shuffle and
                    not args.sort and not args.heavy and not args.graph and not args.links_epoch and args.order_rays == 0)
        if headline and not args.no_also:
            out["also"] = also_lines(args)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))
        if out.get("verify") is not None and not out["verify"]["ok"]:
            sys.exit("bench.py: the timed batch does NOT match the oracle: " + json.dumps(out["verify"]))
        xv = (out.get("exchange") or {}).get("verify")
        if xv is not None and not xv["ok"]:
            sys.exit("bench.py: the EXCHANGED gradients do not match the single-rank render of all shards: " + json.dumps(xv))


if __name__ == "__main__":
    main()
