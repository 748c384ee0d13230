"""The randomised option sweep's draw stream, shared by tests/test_hip_parity.py::test_random_option_fuzz, its
regression case and tests/tools/fuzz_draw.py: draw `it` of seed `seed` is the same scene, rays, options and loss weights
wherever it is replayed."""
import numpy as np


def draws(seed, n_draws, only=None):
    """Yields (it, scene, origins, dirs, opts, fused, rgb_gt, description).  With `only` set, the draws before it are
    skipped without building their scenes (their random numbers are consumed all the same)."""
    from alphasurf_amd import synth
    rs = np.random.RandomState(seed)
    for it in range(n_draws):
        reso = int(rs.choice([24, 40, 57, 72]))
        B = int(rs.choice([1, 4, 9, 16]))
        lv = tuple(sorted(rs.uniform(-0.6, 0.6, size=int(rs.choice([1, 1, 2, 3]))).tolist()))
        sc_args = dict(seed=int(rs.randint(1000)), band=float(rs.uniform(1.5, 4.0)), dilate=int(rs.randint(0, 3)),
                       level_set=lv, trunc_a=float(rs.uniform(1.0, 6.0)), fake_sample_std=float(rs.uniform(0.3, 1.5)))
        ray_seed = int(rs.randint(1000))
        opts = dict(
            sigma_thresh=float(rs.choice([-1e4, -3.0, 0.3, 0.5])), stop_thresh=float(rs.choice([-1e4, 0.0, 1e-7, 1e-2])),
            step_size=float(rs.choice([0.01, 0.5, 2.0])), near_clip=float(rs.choice([0.0, 0.0, 0.2])),
            use_spheric_clip=bool(rs.rand() < 0.2), alpha_activation_type=int(rs.randint(0, 2)),
            only_outward_intersect=bool(rs.rand() < 0.5), truncated_vol_render=bool(rs.rand() < 0.6),
            trunc_vol_weight_min=float(rs.choice([0.0, 1e-10, 0.05])), surf_fake_sample=bool(rs.rand() < 0.5),
            surf_fake_sample_min_vox_len=float(rs.choice([0.0, 0.05, 0.3])), limited_fake_sample=bool(rs.rand() < 0.5),
            fake_sample_l_dist=bool(rs.rand() < 0.5), fake_sample_normalize_surf=bool(rs.rand() < 0.5),
            no_surf_grad_from_sh=bool(rs.rand() < 0.2), background_brightness=float(rs.choice([0.0, 0.5, 1.0])))
        fused = dict(lambda_l2=float(rs.rand()), lambda_l1=float(rs.rand()))
        for k, hi in (("lambda_l_entropy", 0.05), ("lambda_l_dist", 0.05), ("lambda_l_dist_a", 0.05),
                      ("lambda_l_entropy_a", 0.02), ("lambda_l_samp_dist", 0.05), ("lambda_conv_mode_samp", 1e-3),
                      ("sparsity_loss", 0.01), ("lambda_inwards_norm_loss", 0.01), ("lambda_l_di", 0.01),
                      ("lambda_inplace_surf_sparse", 1e-3)):
            if rs.rand() < 0.3:
                fused[k] = float(rs.uniform(0.1, 1.0) * hi)
        fused["l_di_alpha_thresh"] = 0.5
        fused["surf_sparse_alpha_thresh"] = 0.5
        fused["no_norm_weight_l_entropy"] = bool(rs.rand() < 0.5)
        fused["l_dist_max_sample"] = int(rs.choice([2, 8, 64]))
        gt = rs.rand(1500, 3).astype(np.float32)
        if only is not None and it != only:
            continue
        sc = synth.make_sparse_scene(reso, B, **sc_args)
        o, d = synth.random_rays(1500, ray_seed)
        tag = f"draw {it}: reso {reso} B {B} lv {lv} opts {opts} fused {fused}"
        yield it, sc, o, d, opts, fused, gt, tag
