"""Two alternative forms of the fused call that are OFF by default (measured slower on MI355X, kept under test):
ASURF_BWD_SH=tab -- the SH gradient rows of a whole chunk of samples merged through an LDS hash table and a counting
sort before they leave (k_bwd_sh_tab, asurf_bwd_pipeline.h) -- and ASURF_TILES=1 -- an image batch
(asurf_fused_params::image_width) rendered in the order of its 8 x 8 pixel tiles.  Both must be transparent: same
colours, same masks, gradients up to the order of float atomics.  `pytest -m gpu`."""
import numpy as np
import pytest
from test_hip_parity import OPTION_SETS, hip, sparse_case  # noqa: F401

pytestmark = pytest.mark.gpu


def _close(ref, got, what=""):
    assert float(np.abs(got[0] - ref[0]).max()) <= 2.5e-7, what
    assert (got[1]["mask"] == ref[1]["mask"]).all(), what
    for k in ("sh", "density", "surface"):
        scale = max(1.0, float(np.abs(ref[1][k]).max()))
        np.testing.assert_allclose(got[1][k], ref[1][k], rtol=2e-4, atol=2e-6 * scale, err_msg=f"{what} {k}")


@pytest.mark.parametrize("basis_dim", [1, 4, 9, 16])
@pytest.mark.parametrize("optname", ["syn_yaml", "fake_limited", "no_sh_dpos"])
def test_chunk_wide_merge_of_sh_rows(hip, oracle, basis_dim, optname):
    from alphasurf_amd import lib
    lv = (0.0,) if optname != "fake_limited" else (-0.3, 0.0, 0.4)
    sc, o, d = sparse_case(basis_dim, level_set=lv, trunc_a=3.0, fake_sample_std=0.8)
    gt = np.random.RandomState(3).rand(o.shape[0], 3).astype(np.float32)
    fused = dict(lambda_l2=0.7, lambda_l1=0.3, lambda_l_entropy=1e-4)
    ref = hip.render_fused(sc, o, d, OPTION_SETS[optname], gt, fused)
    lib.config_set("ASURF_BWD_SH", "tab")
    got = hip.render_fused(sc, o, d, OPTION_SETS[optname], gt, fused)
    _close(ref, got, optname)
    _, go, _ = oracle.render_fused(sc, o, d, OPTION_SETS[optname], gt, fused)
    for k in ("sh", "surface"):
        scale = max(float(np.abs(go[k]).max()), 1e-12)
        assert float(np.abs(got[1][k] - go[k]).max()) / scale <= 2e-4, k


@pytest.mark.parametrize("basis_dim", [1, 9, 16])
@pytest.mark.parametrize("optname", ["syn_yaml", "fake_limited", "no_sh_dpos"])
def test_chunks_sorted_by_voxel(hip, oracle, basis_dim, optname):
    """k_bwd_sh / k_bwd_geo take the chunks of the record list sorted by voxel (wave_sort64; the default) or as they lie
    (ASURF_BWD_SORT=0, round 3): the same sums in another order."""
    from alphasurf_amd import lib
    lv = (0.0,) if optname != "fake_limited" else (-0.3, 0.0, 0.4)
    sc, o, d = sparse_case(basis_dim, level_set=lv, trunc_a=3.0, fake_sample_std=0.8)
    gt = np.random.RandomState(3).rand(o.shape[0], 3).astype(np.float32)
    fused = dict(lambda_l2=0.7, lambda_l1=0.3, lambda_l_entropy=1e-4)
    got = hip.render_fused(sc, o, d, OPTION_SETS[optname], gt, fused)
    lib.config_set("ASURF_BWD_SORT", "0")
    ref = hip.render_fused(sc, o, d, OPTION_SETS[optname], gt, fused)
    _close(ref, got, optname)
    lib.config_set("ASURF_BWD_SORT", "wide")   # two chunks sorted together (long lists take this form by themselves)
    _close(ref, hip.render_fused(sc, o, d, OPTION_SETS[optname], gt, fused), optname + " wide")
    _, go, _ = oracle.render_fused(sc, o, d, OPTION_SETS[optname], gt, fused)
    for k in ("sh", "density", "surface"):
        scale = max(float(np.abs(go[k]).max()), 1e-12)
        assert float(np.abs(got[1][k] - go[k]).max()) / scale <= 2e-4, k


@pytest.mark.parametrize("optname", ["syn_yaml", "fake_limited"])
def test_kernels_without_an_occupancy_bound(hip, optname):
    """Small batches take k_geom / k_bwd_geo compiled without an occupancy bound (more registers, no scratch;
    ASURF_RELAXED=0: never): the same arithmetic, the same colours bit for bit."""
    from alphasurf_amd import lib
    lv = (0.0,) if optname != "fake_limited" else (-0.3, 0.0, 0.4)
    sc, o, d = sparse_case(9, level_set=lv, trunc_a=3.0, fake_sample_std=0.8)
    gt = np.random.RandomState(3).rand(o.shape[0], 3).astype(np.float32)
    fused = dict(lambda_l2=0.7, lambda_l1=0.3, lambda_l_entropy=1e-4)
    got = hip.render_fused(sc, o, d, OPTION_SETS[optname], gt, fused)
    lib.config_set("ASURF_RELAXED", "0")
    ref = hip.render_fused(sc, o, d, OPTION_SETS[optname], gt, fused)
    assert (got[0] == ref[0]).all()
    _close(ref, got, optname)


def test_image_batch_in_tile_order(hip):
    """A 320 x 264 image (edge tiles: 264 = 33 x 8, the width 320 = 40 x 8; and a second image 300 wide whose last tile
    column is 4 pixels) rendered as given and in tile order, with both backward forms."""
    import torch
    from alphasurf_amd import csrc_api as _C, lib, synth
    H = hip
    sc = synth.make_sparse_scene(64, 9, seed=3, trunc_a=3.0)
    for width, height in ((320, 264), (300, 250)):
        o, d = synth.pinhole_rays(synth.orbit_camera(1), width, height)
        assert o.shape[0] >= 65536
        gt = np.random.RandomState(0).rand(o.shape[0], 3).astype(np.float32)
        g, r, op = H.grid_spec(sc), H.rays_spec(o, d), H.opt_spec(synth.SYN_OPTS)
        gt_t = torch.from_numpy(gt).to(H.dev())

        def call(image_width):
            gr = H.zero_grads(g)
            args = [dflt for _, dflt in H.FUSED_ORDER]
            rgb = torch.full_like(gt_t, float("nan"))   # (every ray is owed its colour)
            _C.volume_render_surf_trav_fused(g, r, op, gt_t, *args, rgb, gr, image_width=image_width, order_rays=0)
            torch.cuda.synchronize()
            return rgb.cpu().numpy(), H.grads_np(gr)

        ref = call(0)
        # the backward's record list alone in tiles (k_bwd_seq; the default for an image batch): every tile shape, the
        # chunks sorted by voxel and as they lie, both forms of k_bwd_sh; shapes that do not partition the image fall
        # back to rows
        for tile in ("8", "16", "4", "32", "0"):
            lib.config_set("ASURF_BWD_TILE", tile)
            for srt in (None, "0", "wide"):
                lib.config_set("ASURF_BWD_SORT", srt)
                for form in (None, "tab"):
                    lib.config_set("ASURF_BWD_SH", form)
                    _close(ref, call(width), f"{width} list tiles {tile} sort {srt} {form}")
        lib.config_set("ASURF_BWD_TILE", None)
        lib.config_set("ASURF_BWD_SORT", None)
        lib.config_set("ASURF_BWD_SH", None)
        lib.config_set("ASURF_TILES", "1")
        for form in (None, "tab"):
            lib.config_set("ASURF_BWD_SH", form)
            _close(ref, call(width), f"{width} {form}")
        lib.config_set("ASURF_TILES", None)
        lib.config_set("ASURF_BWD_SH", None)
