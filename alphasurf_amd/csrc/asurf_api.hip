// asurf_api.hip -- C-ABI entry points of libasurf_hip.so (include/asurf.h).
//
// Host launchers that replace the reference's volume_render_surf_trav{,_backward,_fused}
// (svox2/csrc/render_lerp_kernel_surf_trav.cu:3596-3942).  Validation mirrors the
// reference's TORCH_CHECK/CHECK_INPUT set as far as raw pointers allow; failures return a
// negative code + message instead of throwing.
#include "asurf.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "asurf_kernels.h"
#include "asurf_optim.h"
#include "asurf_pipeline.h"
#include "asurf_sweep.h"
#include "asurf_masks2.h"
#include "asurf_bwd_pipeline.h"
#include "asurf_legacy.h"
#include "asurf_reg.h"
#include "asurf_eval.h"
#include "asurf_misc.h"
#include "asurf_cuvol.h"
#include "asurf_order.h"
#include "asurf_exchange.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

using namespace asurf;

// --- internal switches ---------------------------------------------------------------------------------------------
// Every switch selects between internal forms that compute the SAME results (the tests compare them with each other);
// none is read on the call path.  The ASURF_* environment variables are read once, when the first entry point runs;
// after that only asurf_config_set() (include/asurf.h) changes a switch.  The measurement aids that make results
// invalid (ASURF_DEBUG bits 1, 2, 4 and 256 * n) exist in the instrumented build only (`make timeline`,
// -DASURF_INSTRUMENT): the shipped library masks them off here and compiles their code out (asurf_device.h).
struct Config {
    int debug = 0;         // ASURF_DEBUG: 8 per-voxel walk (k_walk, no brick jumps), 16 print the pipeline's counters,
                           // 32 k_geom without its no-root filter
    int walk_seg = 0;      // ASURF_WALK_SEG = 1, 2, 4, 8: lanes per ray of the stepping walk (0: by batch size)
    int walk_step = 0;     // ASURF_WALK=step: the stepping walk instead of k_sweep
    int rpw = 0;           // ASURF_RPW = 4 .. 64: rays per wave of k_comp / k_bwd_seq (0: by batch size)
    int no_pipeline = 0;   // ASURF_NO_PIPELINE: the one-kernel forward
    int cuvol_rpw = 0;     // ASURF_CUVOL_RPW = 1 .. 64
    int no_replay = 0;     // ASURF_NO_REPLAY: the walking backward
    int no_split_bwd = 0;  // ASURF_NO_SPLIT_BWD: the one-kernel replay
    int eval_walk = 0;     // ASURF_EVAL=walk: the stepping evaluation renders
    int timeline = 0;      // ASURF_TIMELINE = kernel id (instrumented build)
    int masks_single = 0;  // ASURF_MASKS=single / split: the full occupancy pass as one kernel / in three steps whatever
    int masks_split = 0;   // ... the grid's size (tests; by default grids of kMaskSplitRows rows and more are split)
    int side_stream = 0;   // ASURF_SIDE_STREAM: the fused call's fallback kernels on a stream of their own (measured: slower)
    int order_radix = 0;   // ASURF_ORDER=radix: the ray ordering through rocPRIM's radix sort instead of the strip sort
    int masks_row = 0;     // ASURF_MASKS=row: the one-pass occupancy kernel (k_build_masks) instead of the two stages
    int sweep_block = 0;   // ASURF_SWEEP=block: the block-synchronous form of k_sweep (round 2)
    int sweep_wpb = 0;     // ASURF_SWEEP_WPB = 1, 2, 4: waves per block of k_sweep_wave (measurement)
    int sweep_rpw = 0;     // ASURF_SWEEP_RPW = 2, 4, 8, 16: rays per wave of k_sweep_wave (0: by batch size)
    int bwd_sh_tab = 0;    // ASURF_BWD_SH=tab: the chunk-wide merge of the SH gradient rows (k_bwd_sh_tab) instead of k_bwd_sh's
                           // sums along runs of one voxel (measured slower: asurf_bwd_pipeline.h)
    int bwd_sort = 1;      // ASURF_BWD_SORT=0: k_bwd_sh / k_bwd_geo take the chunks of the record list as they lie (round 3) instead
                           // of sorted by voxel; =wide: two chunks sorted together whatever the list's length (tests)
    int bwd_tile = 8;      // ASURF_BWD_TILE = 0 | 4 | 8 | 16 | 32: width of the pixel tile a wave of k_bwd_seq takes of an image batch
                           // (0: a row of 64 pixels, as for any other batch; 8 x 8 measured best)
    int optim_two_step = 0;   // ASURF_OPTIM=list: the masked optimizer steps as list kernel + step kernel (k_compact_segments)
    int relaxed_rays = -1;    // ASURF_RELAXED = n: batches of up to n rays take k_geom / k_bwd_geo without an occupancy bound
    int sum_list = -1;     // ASURF_SUM_LIST=0: the default call (links_epoch = 0) takes the full occupancy pass every time;
                           // =1: the checksum-validated list whatever the grid's size (tests; by default from 128^3 up)
    int cuvol_tile = 8;    // ASURF_CUVOL_TILE = 0 | 4 | 8 | 16: width of the pixel tile a wave of the cuvol kernels takes of an image
                           // batch (asurf_rays::image_width; 0: rows)
    int tiles = 0;         // ASURF_TILES=1: image batches (asurf_fused_params::image_width) are rendered in tile order
                           // (measured slower with the kernels as they are: include/asurf.h)
    int lazy_rays = -1;    // ASURF_LAZY = n: batches of up to n rays run over planes of `links` alone (asurf_masks2.h);
                           // 0 never, -1 the default (kLazyMaxRays: off)
};
#ifdef ASURF_INSTRUMENT
constexpr int kDebugMask = 1 | 2 | 4 | 8 | 16 | 32 | (7 << 8);
#else
constexpr int kDebugMask = 8 | 16 | 32;
#endif
std::mutex g_cfg_mu;
Config g_cfg;
std::once_flag g_cfg_once;

bool config_assign(Config& c, const char* name, const char* value) {
    const int n = value ? atoi(value) : 0;
    if (!strcmp(name, "ASURF_DEBUG")) c.debug = n & kDebugMask;
    else if (!strcmp(name, "ASURF_WALK_SEG")) c.walk_seg = (n == 1 || n == 2 || n == 4 || n == 8) ? n : 0;
    else if (!strcmp(name, "ASURF_WALK")) c.walk_step = (value && !strcmp(value, "step")) ? 1 : 0;
    else if (!strcmp(name, "ASURF_RPW")) c.rpw = (n == 4 || n == 8 || n == 16 || n == 32 || n == 64) ? n : 0;
    else if (!strcmp(name, "ASURF_NO_PIPELINE")) c.no_pipeline = n ? 1 : 0;
    else if (!strcmp(name, "ASURF_CUVOL_RPW")) c.cuvol_rpw = (n >= 1 && n <= 64 && (n & (n - 1)) == 0) ? n : 0;
    else if (!strcmp(name, "ASURF_NO_REPLAY")) c.no_replay = n ? 1 : 0;
    else if (!strcmp(name, "ASURF_NO_SPLIT_BWD")) c.no_split_bwd = n ? 1 : 0;
    else if (!strcmp(name, "ASURF_EVAL")) c.eval_walk = (value && !strcmp(value, "walk")) ? 1 : 0;
    else if (!strcmp(name, "ASURF_TIMELINE")) c.timeline = n;
    else if (!strcmp(name, "ASURF_SIDE_STREAM")) c.side_stream = n ? 1 : 0;
    else if (!strcmp(name, "ASURF_ORDER")) c.order_radix = (value && !strcmp(value, "radix")) ? 1 : 0;
    else if (!strcmp(name, "ASURF_MASKS")) {
        c.masks_row = (value && !strcmp(value, "row")) ? 1 : 0;
        c.masks_single = (value && !strcmp(value, "single")) ? 1 : 0;
        c.masks_split = (value && !strcmp(value, "split")) ? 1 : 0;
    }
    else if (!strcmp(name, "ASURF_SWEEP")) c.sweep_block = (value && !strcmp(value, "block")) ? 1 : 0;
    else if (!strcmp(name, "ASURF_SWEEP_WPB")) c.sweep_wpb = (n == 1 || n == 2 || n == 4) ? n : 0;
    else if (!strcmp(name, "ASURF_SWEEP_RPW")) c.sweep_rpw = (n == 1 || n == 2 || n == 4 || n == 8 || n == 16) ? n : 0;
    else if (!strcmp(name, "ASURF_BWD_SH")) c.bwd_sh_tab = (value && !strcmp(value, "tab")) ? 1 : 0;
    else if (!strcmp(name, "ASURF_BWD_SORT")) c.bwd_sort = (value && !strcmp(value, "0")) ? 0 : (value && !strcmp(value, "wide")) ? 2 : 1;
    else if (!strcmp(name, "ASURF_BWD_TILE")) c.bwd_tile = (!value || !*value) ? 8 : (n == 4 || n == 8 || n == 16 || n == 32) ? n : 0;
    else if (!strcmp(name, "ASURF_OPTIM")) c.optim_two_step = (value && !strcmp(value, "list")) ? 1 : 0;
    else if (!strcmp(name, "ASURF_RELAXED")) c.relaxed_rays = (value && *value) ? std::max(n, 0) : -1;
    else if (!strcmp(name, "ASURF_SUM_LIST")) c.sum_list = (value && !strcmp(value, "0")) ? 0 : (value && !strcmp(value, "1")) ? 1 : -1;
    else if (!strcmp(name, "ASURF_CUVOL_TILE")) c.cuvol_tile = (!value || !*value) ? 8 : (n == 4 || n == 8 || n == 16) ? n : 0;
    else if (!strcmp(name, "ASURF_TILES")) c.tiles = (value && !strcmp(value, "1")) ? 1 : 0;
    else if (!strcmp(name, "ASURF_LAZY")) c.lazy_rays = (value && *value) ? std::max(n, 0) : -1;
    else return false;
    return true;
}

Config config() {  // a copy: a call works with one consistent set
    std::call_once(g_cfg_once, [] {
        static const char* const names[] = {"ASURF_DEBUG", "ASURF_WALK_SEG", "ASURF_WALK", "ASURF_RPW", "ASURF_NO_PIPELINE",
                                            "ASURF_CUVOL_RPW", "ASURF_NO_REPLAY", "ASURF_NO_SPLIT_BWD", "ASURF_EVAL",
                                            "ASURF_TIMELINE", "ASURF_SWEEP", "ASURF_SWEEP_WPB", "ASURF_SWEEP_RPW", "ASURF_MASKS", "ASURF_SIDE_STREAM", "ASURF_ORDER",
                                            "ASURF_LAZY", "ASURF_BWD_SH", "ASURF_TILES", "ASURF_BWD_SORT", "ASURF_BWD_TILE", "ASURF_OPTIM", "ASURF_RELAXED", "ASURF_SUM_LIST", "ASURF_CUVOL_TILE"};
        std::lock_guard<std::mutex> lock(g_cfg_mu);
        for (const char* nm : names)
            if (const char* v = getenv(nm)) config_assign(g_cfg, nm, v);
    });
    std::lock_guard<std::mutex> lock(g_cfg_mu);
    return g_cfg;
}

struct MaskLayout {
    int nb[3];
    int nwz;
    size_t n_bricks;
    size_t coarse_words;
    size_t off_fine_work, off_fine_live, off_coarse_work, off_coarse_live, off_fine_y, off_fine_z;
    size_t off_mask_hdr, off_live_rows;  // MaskHdr and the list of rows of bricks that are not empty space
    int n_rows;                          // rows of 8 bricks along z
    size_t off_cache_a, off_cache_w, off_cache_t, off_cache_n;
    size_t off_rec_v, off_rec_m, off_rec_n, off_rec_neff, off_rec_sws, off_rec_f;
    size_t off_p_items, off_p_citem, off_p_nitems, off_p_cut, off_p_flagged, off_p_counters, off_p_pool;
    size_t off_p_cand;  // cand_mask [Q]
    size_t off_p_spbase, off_p_spill, off_p_list, off_rec_list;
    int rec_depth;   // records a ray (layout_for)
    // ray ordering inside the fused call (batches of kOrderMinRays rays and more): permutation, the rays / targets /
    // colours in the new order, sort scratch, the coherence probe's two counts
    size_t off_ord_perm, off_ord_o, off_ord_d, off_ord_gt, off_ord_rgb, off_ord_tmp, off_ord_probe, ord_tmp_bytes;
    int pool_shard_cap, sp_cap, list_cap, band_rays, rec_list_cap;
    size_t total;
};

// Lanes per ray in the walk.  k_walk's duration is that of its longest serial chains until the batch
// fills the chip several times over: measured on the bench scene (ms per fused step, 1 / 4 / 8 lanes):
// 1.3 k rays 0.64 / 0.48 / 0.39, 20 k 0.77 / 0.56 / 0.55, 80 k 0.86 / 0.64 / 0.67, 160 k 0.95 / 0.79 / 0.85,
// 640 k 1.52 / 1.65 / 1.79.  ASURF_WALK_SEG = 1, 2, 4 or 8 overrides (tests, measurement).
static int walk_segments(long long n_rays) {
    if (const int n = config().walk_seg) return n;
    if (n_rays <= 8192) return 8;
    if (n_rays <= 262144) return 4;
    return 1;
}

// Batches this large are put in a coherent order inside the fused call when they arrive shuffled
// (asurf_fused_params::order_rays): below, the ordering's fixed costs exceed what it returns
constexpr long long kOrderMinRays = 65536;

// per-ray capacity of the sample records the fused forward leaves for the replaying backward; rays with more samples fall
// back to the walking backward kernel.  kRecDepth is what asurf_workspace_bytes() sizes the workspace for; a caller that
// expects more samples a ray (fake samples in every voxel: 11 a ray on the bench scene, a fifth of the rays beyond 32)
// asks asurf_workspace_bytes_rec() for a deeper one, and every call takes the deepest of kRecDepths that the workspace
// it is handed holds -- a function of the buffer's size alone, so a buffer has one layout.
constexpr int kRecDepth = 32;
// Batches up to this many rays test planes of `links` alone and apply the value gates to the voxels they meet
// (asurf_masks2.h, k_links_planes; KOpt::lazy_gates).  Built for the training script's 5 000-ray batches, where the
// per-call pass over the volume is 0.063 of the call's 0.158 ms -- measured (MI355X, 5 184 shuffled rays of the bench
// scene) and left OFF: the checksum of `links` takes 0.017 ms and the decision 0.006, but a ray meets 34 voxels of the
// shell where it meets 6 work voxels, and in k_sweep's exact test each turn of 64 of them is two dependent gathers
// (links, then values): k_sweep 0.024 -> 0.053-0.065 ms (1 to 4 rays a wave; larger task and key buffers change
// nothing), the call 0.1617 -> 0.1519 ms without a promise and 0.1337 -> 0.148 with links_epoch.  ASURF_LAZY = n turns
// it on for batches of up to n rays (tests/test_links_planes_gpu.py keeps the form alive).
constexpr long long kLazyMaxRays = 0;
// blocks of k_links_planes: as a rule they read the header, find the planes vouched for and leave -- few blocks, few
// tickets; when the planes must be rebuilt (once per change of `links`) each takes its share of the rows
constexpr int kLinksPlanesBlocks = 64;
constexpr int kMaskSplitRows = 16384;   // rows of 8 bricks from which the full occupancy pass is split (512^3: 32 768)
constexpr int kRecDepths[] = {128, 64, 32};
static_assert(kRecDepths[0] <= (1 << (32 - kRecRayBits)), "a record's number fits above the ray's in a list cell");

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

MaskLayout layout_for(const int32_t size[3], int64_t n_rays, int32_t M, int rec_depth = kRecDepth) {
    MaskLayout L;
    L.rec_depth = rec_depth;
    for (int i = 0; i < 3; ++i) L.nb[i] = (size[i] + 7) / 8;
    L.nwz = (L.nb[2] + 31) / 32;
    L.n_bricks = (size_t)L.nb[0] * L.nb[1] * L.nb[2];
    L.coarse_words = (size_t)L.nb[0] * L.nb[1] * L.nwz;
    size_t off = 0;
    L.off_fine_work = off;
    off = align_up(off + L.n_bricks * 64, 256);
    L.off_fine_live = off;
    off = align_up(off + L.n_bricks * 64, 256);
    L.off_coarse_work = off;
    off = align_up(off + L.coarse_words * 4, 256);
    L.off_coarse_live = off;
    off = align_up(off + L.coarse_words * 4, 256);
    L.off_fine_y = off;  // the plane the walk tests, cut along y and along z (k_sweep)
    off = align_up(off + L.n_bricks * 64, 256);
    L.off_fine_z = off;
    off = align_up(off + L.n_bricks * 64, 256);
    L.n_rows = (int)((size_t)L.nb[0] * L.nb[1] * ((L.nb[2] + 7) / 8));
    L.off_mask_hdr = off;
    off = align_up(off + sizeof(MaskHdr), 256);
    // the pipeline's counters lie right behind the header: a pass that does not vouch for the header clears both with
    // one memset (n_flagged, spill cursor, kPoolShards pool cursors, the lists' cursors)
    L.off_p_counters = off;
    off = align_up(off + kCtrTotal * 4, 256);
    L.off_live_rows = off;
    off = align_up(off + (size_t)L.n_rows * 4, 256);
    const size_t plane = (size_t)(M > 0 ? M : 0) * (size_t)(n_rays > 0 ? n_rays : 0) * 4;
    L.off_cache_a = off;
    off = align_up(off + plane, 256);
    L.off_cache_w = off;
    off = align_up(off + plane, 256);
    L.off_cache_t = off;
    off = align_up(off + plane, 256);
    L.off_cache_n = off;
    off = align_up(off + (size_t)(n_rays > 0 ? n_rays : 0) * 4, 256);
    // sample records exist only for fused calls (M > 0 is not required for them, but a
    // forward-only / backward-only call never passes through the fused path)
    const size_t rplane = (size_t)rec_depth * (size_t)(n_rays > 0 ? n_rays : 0) * 4;
    L.off_rec_v = off;
    off = align_up(off + rplane, 256);
    L.off_rec_m = off;
    off = align_up(off + rplane, 256);
    L.off_rec_n = off;
    off = align_up(off + (size_t)(n_rays > 0 ? n_rays : 0) * 4, 256);
    L.off_rec_neff = off;
    off = align_up(off + (size_t)(n_rays > 0 ? n_rays : 0) * 4, 256);
    L.off_rec_sws = off;
    off = align_up(off + (size_t)(n_rays > 0 ? n_rays : 0) * 4, 256);
    L.off_rec_f = off;  // 8 forward + 9 backward float planes [rec_depth][Q]
    off = align_up(off + 17 * rplane, 256);
    // three-kernel forward pipeline (asurf_pipeline.h)
    const size_t nr = (size_t)(n_rays > 0 ? n_rays : 0);
    const size_t iplane = (size_t)kPipeC1 * nr * 4;
    L.off_p_items = off;
    off = align_up(off + iplane, 256);
    L.off_p_citem = off;
    off = align_up(off + iplane, 256);
    L.off_p_nitems = off;
    off = align_up(off + nr * 4, 256);
    L.off_p_cut = off;
    off = align_up(off + nr * 4, 256);
    L.off_p_flagged = off;
    off = align_up(off + nr * 4, 256);
    L.off_p_spbase = off;
    off = align_up(off + nr * 4, 256);
    L.off_p_cand = off;
    off = align_up(off + nr * 8, 256);
    L.sp_cap = (int)(8 * nr) + 256;  // spilled work voxels (beyond kPipeC1 per ray), all rays together (dense, noisy
                                     // scenes reach 6 a ray; beyond the capacity rays go to the one-kernel path)
    L.off_p_spill = off;             // sp_items | sp_citem | sp_ray
    off = align_up(off + (size_t)L.sp_cap * 3 * 4, 256);
    L.off_p_list = off;      // dense list of occupied table cells (k_walk / k_sweep -> k_geom), kListSegs segments: the
                             // producer blocks of the bands of group g (list_segment, asurf_pipeline.h) cover at most
                             // nr / kListGroups + band_rays rays, a kListShards-th of them per segment (a band is a whole
                             // number of rounds of 16 producer blocks)
    L.band_rays = kBandRaysMax;
    while (L.band_rays > kBandRaysMin && (long long)L.band_rays * 64 > nr) L.band_rays >>= 1;
    L.list_cap = kPipeC1 * (int)(((nr + kListGroups - 1) / kListGroups + L.band_rays) / kListShards + 256);
    off = align_up(off + (size_t)L.list_cap * kListSegs * 4, 256);
    L.off_rec_list = off;    // dense list of visited records (k_bwd_seq -> k_bwd_sh, k_bwd_geo), kRecShards segments:
                             // block b of k_bwd_seq (256 rays, rec_depth records each at most) fills segment b % 8
    L.rec_list_cap = (int)(((nr + kBlock - 1) / kBlock + kRecShards - 1) / kRecShards) * kBlock * rec_depth;
    off = align_up(off + (size_t)L.rec_list_cap * kRecShards * 4, 256);
    // 16 candidates a ray on average (the live config yields 0.7; three level sets with fake samples 11: with 8 a fifth of
    // the rays of such a scene went to the one-kernel path, whose walking backward then took 30 of the step's 39 ms)
    L.pool_shard_cap = (int)((16 * nr + kPoolShards - 1) / kPoolShards) + 256;  // (a wave of k_geom may ask for 3 x 64 at once)
    L.off_p_pool = off;
    off = align_up(off + (size_t)L.pool_shard_cap * kPoolShards * sizeof(CandRec), 256);
    const size_t onr = (n_rays >= kOrderMinRays) ? nr : 0;
    L.off_ord_perm = off;
    off = align_up(off + onr * 4, 256);
    L.off_ord_o = off;
    off = align_up(off + onr * 12, 256);
    L.off_ord_d = off;
    off = align_up(off + onr * 12, 256);
    L.off_ord_gt = off;
    off = align_up(off + onr * 12, 256);
    L.off_ord_rgb = off;
    off = align_up(off + onr * 12, 256);
    L.ord_tmp_bytes = onr ? std::max(asurf_order::tmp_bytes32((long long)onr), asurf_order::tmp_bytes_strips((long long)onr)) : 0;
    L.off_ord_tmp = off;
    off = align_up(off + L.ord_tmp_bytes, 256);
    L.off_ord_probe = off;
    off = align_up(off + (onr ? 3 * asurf_order::kProbeBlocks * sizeof(int) : 0), 256);
    L.total = off;
    return L;
}

// the layout of a workspace of `bytes` bytes: the deepest records that fit (kRecDepth when nothing does: the caller's size
// check then fails as it always did)
MaskLayout layout_fitting_uncached(const int32_t size[3], int64_t n_rays, int32_t M, size_t bytes);

// (a call of a training loop asks for the layout it asked for the last time: up to three layouts, each with two rocPRIM
// temp-size queries for batches that may be ordered, were recomputed on every call -- ADVICE r3)
MaskLayout layout_fitting(const int32_t size[3], int64_t n_rays, int32_t M, size_t bytes) {
    struct Key {
        int32_t size[3];
        int64_t n_rays;
        int32_t M;
        size_t bytes;
    };
    thread_local Key k_last = {{0, 0, 0}, -1, 0, 0};
    thread_local MaskLayout l_last;
    if (k_last.n_rays == n_rays && k_last.M == M && k_last.bytes == bytes && k_last.size[0] == size[0] &&
        k_last.size[1] == size[1] && k_last.size[2] == size[2])
        return l_last;
    l_last = layout_fitting_uncached(size, n_rays, M, bytes);
    k_last = Key{{size[0], size[1], size[2]}, n_rays, M, bytes};
    return l_last;
}

MaskLayout layout_fitting_uncached(const int32_t size[3], int64_t n_rays, int32_t M, size_t bytes) {
    if (n_rays > 0)
        for (int d : kRecDepths) {
            if (d == kRecDepth) break;
            const MaskLayout L = layout_for(size, n_rays, M, d);
            if (L.total <= bytes) return L;
        }
    return layout_for(size, n_rays, M);
}

int check_grid(const asurf_grid* g, bool need_surface = true) {
    if (!g) return fail(ASURF_E_INVALID, "grid is NULL");
    if (!g->links || !g->density_data || !g->sh_data)
        return fail(ASURF_E_INVALID, "links, density_data and sh_data must be device tensors");
    if (need_surface && (g->surface_type == ASURF_SURFACE_TYPE_NONE || !g->surface_data))
        return fail(ASURF_E_INVALID, "surf_trav needs surface_data (surface_type != NONE)");
    if (g->level_set_num < 0 || (g->level_set_num > 0 && !g->level_set_data))
        return fail(ASURF_E_INVALID, "level_set_data missing");
    if (g->basis_type != ASURF_BASIS_TYPE_SH)
        return fail(ASURF_E_UNSUPPORTED, "basis_type %d: only BASIS_TYPE_SH is built (SURVEY.md 8f)",
                    g->basis_type);
    if (!(g->basis_dim == 1 || g->basis_dim == 4 || g->basis_dim == 9 || g->basis_dim == 16))
        return fail(ASURF_E_INVALID, "basis_dim must be 1, 4, 9 or 16 (got %d)", g->basis_dim);
    if (g->background_nlayers != 0)
        return fail(ASURF_E_UNSUPPORTED, "MSI background layers are outside this build's scope");
    for (int i = 0; i < 3; ++i)
        if (g->size[i] < 2 || g->size[i] > 1024)
            return fail(ASURF_E_INVALID, "links.shape[%d] = %d outside [2, 1024] (voxel coordinates are "
                        "packed 10 bits per axis)", i, g->size[i]);
    if ((int64_t)g->size[0] * g->size[1] * g->size[2] >= (1ll << 31))
        return fail(ASURF_E_INVALID, "links has >= 2^31 elements");
    if (g->capacity <= 0) return fail(ASURF_E_INVALID, "capacity must be > 0");
    return 0;
}

int check_rays(const asurf_rays* r) {
    if (!r) return fail(ASURF_E_INVALID, "rays is NULL");
    if (r->n_rays < 0) return fail(ASURF_E_INVALID, "n_rays < 0");
    if (r->n_rays > 0 && (!r->origins || !r->dirs))
        return fail(ASURF_E_INVALID, "rays.origins / rays.dirs must be device tensors");
    if (r->n_rays * 3 >= (1ll << 31)) return fail(ASURF_E_INVALID, "too many rays for 32-bit indexing");
    return 0;
}

KGrid make_kgrid(const asurf_grid* g, const MaskLayout& ML, char* ws, bool use_live_plane) {
    KGrid k;
    k.links = g->links;
    k.density = g->density_data;
    k.surface = g->surface_data;
    k.sh = g->sh_data;
    k.level_set = g->level_set_data;
    for (int i = 0; i < 3; ++i) {
        k.size[i] = g->size[i];
        k.offset[i] = g->offset[i];
        k.scaling[i] = g->scaling[i];
        k.nb[i] = ML.nb[i];
    }
    k.offx = g->size[1] * g->size[2];
    k.offy = g->size[2];
    k.level_set_num = g->level_set_num;
    k.fake_sample_std = g->fake_sample_std;
    k.trunc_a = g->truncated_vol_render_a;
    auto* fw = reinterpret_cast<unsigned long long*>(ws + ML.off_fine_work);
    auto* fl = reinterpret_cast<unsigned long long*>(ws + ML.off_fine_live);
    auto* cw = reinterpret_cast<unsigned int*>(ws + ML.off_coarse_work);
    auto* cl = reinterpret_cast<unsigned int*>(ws + ML.off_coarse_live);
    k.fine = use_live_plane ? fl : fw;
    k.fine_live = fl;
    k.fine_y = reinterpret_cast<unsigned long long*>(ws + ML.off_fine_y);
    k.fine_z = reinterpret_cast<unsigned long long*>(ws + ML.off_fine_z);
    k.coarse = use_live_plane ? cl : cw;
    k.nwz = ML.nwz;
    k.coarse_words = (int)ML.coarse_words;
    return k;
}

KOpt make_kopt(const asurf_render_options* o) {
    KOpt k;
    k.background_brightness = o->background_brightness;
    k.step_size = o->step_size;
    k.sigma_thresh = o->sigma_thresh;
    k.stop_thresh = o->stop_thresh;
    k.near_clip = o->near_clip;
    k.use_spheric_clip = o->use_spheric_clip;
    k.surf_fake_sample = o->surf_fake_sample;
    k.surf_fake_sample_min_vox_len = o->surf_fake_sample_min_vox_len;
    k.limited_fake_sample = o->limited_fake_sample;
    k.no_surf_grad_from_sh = o->no_surf_grad_from_sh;
    k.alpha_activation_type = o->alpha_activation_type;
    k.fake_sample_l_dist = o->fake_sample_l_dist;
    k.fake_sample_normalize_surf = o->fake_sample_normalize_surf;
    k.only_outward_intersect = o->only_outward_intersect;
    k.truncated_vol_render = o->truncated_vol_render;
    k.trunc_vol_weight_min = o->trunc_vol_weight_min;
    const Config c = config();
    k.debug = c.debug;
    k.lazy_gates = 0;
#ifdef ASURF_TIMELINE
    static std::once_flag tl_once;  // ASURF_TIMELINE=<kernel id>: that kernel's waves leave their times (asurf_device.h)
    std::call_once(tl_once, [&c] {
        const int id = c.timeline;
        if (id > 0) (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_kernel), &id, sizeof(int));
    });
#endif
    return k;
}

constexpr size_t kMaxCoarseLds = 32 * 1024;

struct Launch {
    KGrid g;
    KOpt o;
    MaskLayout ML;
    char* ws;
    hipStream_t stream;
    int coarse_in_lds;
    size_t dyn_lds;
    dim3 ray_grid;
    bool sweep;  // the forward pipeline finds its work voxels with k_sweep instead of k_walk
    bool lazy;   // the planes hold the links' test only (small batches): KOpt::lazy_gates is set
};

// k_sweep serves the calls whose walk tests the sparse `work` plane (on the `live` plane nearly every voxel
// of the shell is flagged and listing them per slab has no advantage over stepping); ASURF_WALK=step and
// ASURF_DEBUG & 8 (the per-voxel reference walk of the tests) keep k_walk.
bool want_sweep(bool use_live, const KOpt& o) {
    if (use_live || (o.debug & 8)) return false;
    return !config().walk_step;
}

// --- optional per-kernel timing (asurf_profile_enable): one hipEvent after every launch ---
// The state is per host thread (like g_err): a thread that enables profiling times its own calls and reads
// its own spans; calls made by other threads (e.g. an autograd backward thread) are not timed.
constexpr int kMaxSpans = 24;
thread_local bool g_prof_on = false;
thread_local hipEvent_t g_ev[kMaxSpans + 1] = {};
thread_local const char* g_span_name[kMaxSpans] = {};
thread_local int g_span_group[kMaxSpans] = {};  // 0 occupancy pass, 1 forward, 2 backward
thread_local int g_spans = -1;                  // spans recorded by the last call (-1: none)

void prof_begin(hipStream_t s) {
    g_spans = -1;
    if (!g_prof_on) return;
    if (!g_ev[0] && hipEventCreate(&g_ev[0]) != hipSuccess) return;
    if (hipEventRecord(g_ev[0], s) == hipSuccess) g_spans = 0;
}

// closes the span of the kernel(s) launched since the previous mark
void prof_after(const char* name, int group, hipStream_t s) {
    if (!g_prof_on || g_spans < 0 || g_spans >= kMaxSpans) return;
    const int i = g_spans + 1;
    if (!g_ev[i] && hipEventCreate(&g_ev[i]) != hipSuccess) return;
    if (hipEventRecord(g_ev[i], s) != hipSuccess) return;
    g_span_name[g_spans] = name;
    g_span_group[g_spans] = group;
    g_spans = i;
}

int launch_error(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ASURF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

// --- asurf_grid::links_epoch: which workspaces hold a list of live rows, and for what ---
// Host-side table (the device-side header is what k_build_masks_live checks before it reads anything): a few
// entries, least recently used replaced.  The length of the list is read back once, asynchronously, into pinned
// memory; until it has arrived the short pass is launched with one block per row and the blocks beyond the list
// leave at once.
struct MaskCacheEntry {
    void* ws = nullptr;
    unsigned long long epoch = 0;
    const int32_t* links = nullptr;
    int32_t size[3] = {0, 0, 0};
    int* h_count = nullptr;   // pinned
    hipEvent_t ev = nullptr;  // the copy into h_count
    bool pending = false;
    int count = -1;
    unsigned long long used = 0;
    int kind = 0;   // 0: the list of live rows beside per-call planes; 1: planes of `links` alone (k_links_planes);
                    // 2: the list of live rows vouched for by a checksum of `links` the call takes itself
    int parity = 0;           // (2) the set of checksum slots the NEXT call adds into
    int* h_invalid = nullptr; // (2) pinned: set by the device when a call's checksum did not match the list's
};
constexpr int kMaskCacheEntries = 8;
std::mutex g_mask_cache_mu;
MaskCacheEntry g_mask_cache[kMaskCacheEntries];
unsigned long long g_mask_cache_clock = 0;

// 1: the workspace holds the list for this epoch (live_blocks = blocks to launch); 0: it does not;
// -1: epoch 0 (whatever the table says about this workspace is dropped: its planes are about to be rewritten in full,
// which keeps the list true, but nothing promises `links` is what it was)
int mask_cache_lookup(void* ws, unsigned long long epoch, const asurf_grid* grid, int n_rows, int* live_blocks,
                      bool capturing) {
    std::lock_guard<std::mutex> lock(g_mask_cache_mu);
    for (MaskCacheEntry& e : g_mask_cache) {
        if (e.ws != ws) continue;
        if (epoch == 0ull || e.kind != 0) {   // (kind 1: the planes of links alone are about to be overwritten, their
            e.ws = nullptr;                   //  header cleared by the full pass's memset)
            return epoch == 0ull ? -1 : 0;
        }
        if (e.epoch != epoch || e.links != grid->links || e.size[0] != grid->size[0] || e.size[1] != grid->size[1] ||
            e.size[2] != grid->size[2]) {
            // another grid's pass is about to overwrite the planes.  Outside a capture the caller stores the new value
            // over this entry; a capturing call stores nothing, so the entry goes now (the replays clear the
            // device-side header themselves)
            if (capturing) e.ws = nullptr;
            return 0;
        }
        if (e.pending && !capturing) {  // (no query while a graph is being captured)
            if (hipEventQuery(e.ev) == hipSuccess) {
                e.pending = false;
                e.count = *e.h_count;
            } else {
                (void)hipGetLastError();  // "not ready" is no error of this call: keep it out of launch_error()
            }
        }
        e.used = ++g_mask_cache_clock;
        *live_blocks = (e.pending || e.count < 0) ? n_rows : std::max(1, std::min(e.count, n_rows));
        return 1;
    }
    return epoch == 0ull ? -1 : 0;
}

void mask_cache_store(void* ws, unsigned long long epoch, const asurf_grid* grid, const int* d_count, hipStream_t stream,
                      int kind = 0) {
    std::lock_guard<std::mutex> lock(g_mask_cache_mu);
    MaskCacheEntry* slot = nullptr;
    for (MaskCacheEntry& e : g_mask_cache)
        if (e.ws == ws) slot = &e;
    if (!slot) {
        slot = &g_mask_cache[0];
        for (MaskCacheEntry& e : g_mask_cache)
            if (e.ws == nullptr || e.used < slot->used) {
                slot = &e;
                if (e.ws == nullptr) break;
            }
    }
    if (slot->pending) (void)hipEventSynchronize(slot->ev);  // (its old copy still writes h_count)
    slot->ws = ws;
    slot->epoch = epoch;
    slot->links = grid->links;
    for (int i = 0; i < 3; ++i) slot->size[i] = grid->size[i];
    slot->count = -1;
    slot->pending = false;
    slot->kind = kind;
    slot->parity = 1;   // (the building call used set 0)
    if (kind == 2) {
        if (!slot->h_invalid && hipHostMalloc(reinterpret_cast<void**>(&slot->h_invalid), sizeof(int)) != hipSuccess)
            slot->h_invalid = nullptr;   // (without the word a broken list costs every call the long way; still correct)
        if (slot->h_invalid) *slot->h_invalid = 0;
    }
    slot->used = ++g_mask_cache_clock;
    if (!slot->h_count && hipHostMalloc(reinterpret_cast<void**>(&slot->h_count), sizeof(int)) != hipSuccess) {
        slot->h_count = nullptr;
        return;  // the list is used all the same, with the long grid
    }
    if (!slot->ev && hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming) != hipSuccess) {
        slot->ev = nullptr;
        return;
    }
    if (hipMemcpyAsync(slot->h_count, d_count, sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess &&
        hipEventRecord(slot->ev, stream) == hipSuccess)
        slot->pending = true;
}

// The default call (links_epoch = 0): does this workspace hold a list of live rows built under a checksum of these
// `links`?  1: yes (live_blocks, the parity of this call's checksum slots and the device's "did not match" word are
// returned; the entry's parity flips); 0: no, or the device has reported a mismatch since: the caller rebuilds.
int sum_cache_lookup(void* ws, const asurf_grid* grid, int n_rows, int* live_blocks, int* parity, int** h_invalid) {
    std::lock_guard<std::mutex> lock(g_mask_cache_mu);
    for (MaskCacheEntry& e : g_mask_cache) {
        if (e.ws != ws) continue;
        if (e.kind != 2 || e.links != grid->links || e.size[0] != grid->size[0] || e.size[1] != grid->size[1] ||
            e.size[2] != grid->size[2] || (e.h_invalid != nullptr && *e.h_invalid != 0))
            return 0;
        if (e.pending) {
            if (hipEventQuery(e.ev) == hipSuccess) {
                e.pending = false;
                e.count = *e.h_count;
            } else {
                (void)hipGetLastError();
            }
        }
        e.used = ++g_mask_cache_clock;
        *live_blocks = (e.pending || e.count < 0) ? n_rows : std::max(1, std::min(e.count, n_rows));
        *parity = e.parity;
        e.parity ^= 1;
        *h_invalid = e.h_invalid;
        return 1;
    }
    return 0;
}

// Planes of `links` alone (small batches): has this workspace's header been initialised for these links?  The header on
// the device decides whether the planes are trusted (k_links_planes); this table only spares the memset that makes a
// header of unknown content safe to read.  1: yes; 0: no -- the caller clears the header, and the entry is made;
// -1: no, and nothing may be cleared or stored now (a graph is being captured): the call takes the per-call pass.
int lazy_cache_lookup(void* ws, const asurf_grid* grid, bool capturing) {
    std::lock_guard<std::mutex> lock(g_mask_cache_mu);
    MaskCacheEntry* slot = nullptr;
    for (MaskCacheEntry& e : g_mask_cache)
        if (e.ws == ws) slot = &e;
    if (slot != nullptr && slot->kind == 1 && slot->links == grid->links && slot->size[0] == grid->size[0] &&
        slot->size[1] == grid->size[1] && slot->size[2] == grid->size[2]) {
        slot->used = ++g_mask_cache_clock;
        return 1;
    }
    if (capturing) return -1;
    if (!slot) {
        slot = &g_mask_cache[0];
        for (MaskCacheEntry& e : g_mask_cache)
            if (e.ws == nullptr || e.used < slot->used) {
                slot = &e;
                if (e.ws == nullptr) break;
            }
    }
    if (slot->pending) (void)hipEventSynchronize(slot->ev);  // (its old copy still writes h_count)
    slot->ws = ws;
    slot->epoch = 0;
    slot->links = grid->links;
    for (int i = 0; i < 3; ++i) slot->size[i] = grid->size[i];
    slot->count = -1;
    slot->pending = false;
    slot->kind = 1;
    slot->used = ++g_mask_cache_clock;
    return 0;
}

// --- asurf_fused_params::order_rays = 0 (auto): was the last batch rendered through this workspace coherent? ---
// Every fused call over kOrderMinRays rays leaves the answer of a probe (asurf_order.h: a sample of consecutive ray
// pairs) in pinned memory, asynchronously; the next call reads it if it has arrived.  Unknown means "as given".
struct OrderState {
    void* ws = nullptr;
    int* h_probe = nullptr;   // pinned: coherent pairs, pairs looked at, pairs with different origins, per probe block
    hipEvent_t ev = nullptr;
    bool pending = false;
    int coherent = -1;        // -1 unknown, 0 shuffled, 1 coherent
    int one_origin = 0;       // 1: no sampled pair of the last batch had two origins (a batch of one camera)
    unsigned long long used = 0;
};
constexpr int kOrderStates = 8;
std::mutex g_order_mu;
OrderState g_order[kOrderStates];
unsigned long long g_order_clock = 0;

// returns what is known about the batches of `ws` (-1 / 0 / 1) and, unless a graph is being captured, queues the probe of
// THIS batch for the next call
// launch = false: what is known (called before the render); true: queue the probe of this batch (called after the
// render's last kernel, so that neither the probe nor its copy to the host stands in front of anything)
int order_probe(void* ws, const float* origins, const float* dirs, long long n, int* d_probe, hipStream_t stream,
                bool capturing, int* one_origin, bool launch) {
    std::lock_guard<std::mutex> lock(g_order_mu);
    OrderState* slot = nullptr;
    for (OrderState& e : g_order)
        if (e.ws == ws) slot = &e;
    if (!slot) {
        // (a workspace the table has not met, during a capture: nothing is known, and no slot is evicted -- waiting for
        // its probe's copy would be a synchronisation inside the capture)
        if (capturing) {
            *one_origin = 0;
            return -1;
        }
        slot = &g_order[0];
        for (OrderState& e : g_order)
            if (e.ws == nullptr || e.used < slot->used) {
                slot = &e;
                if (e.ws == nullptr) break;
            }
        if (slot->pending && slot->ev) (void)hipEventSynchronize(slot->ev);
        slot->ws = ws;
        slot->pending = false;
        slot->coherent = -1;
        slot->one_origin = 0;
    }
    slot->used = ++g_order_clock;
    // (one probe in flight at a time; none while a graph is being captured)
    bool may_probe = !capturing;
    if (!launch && !(may_probe && slot->pending)) may_probe = false;   // (before the render: only collect an answer)
    if (may_probe && slot->pending) {
        if (hipEventQuery(slot->ev) == hipSuccess) {
            slot->pending = false;
            int ok = 0, seen = 0, diff = 0;
            for (int b = 0; b < asurf_order::kProbeBlocks; ++b) {
                ok += slot->h_probe[3 * b];
                seen += slot->h_probe[3 * b + 1];
                diff += slot->h_probe[3 * b + 2];
            }
            if (seen > 0) {
                slot->coherent = (2 * ok >= seen) ? 1 : 0;
                slot->one_origin = diff == 0 ? 1 : 0;
            }
        } else {
            (void)hipGetLastError();  // "not ready" is no error of this call
            may_probe = false;
        }
    }
    if (may_probe && !slot->h_probe && hipHostMalloc(reinterpret_cast<void**>(&slot->h_probe), 3 * asurf_order::kProbeBlocks * sizeof(int)) != hipSuccess) {
        slot->h_probe = nullptr;
        may_probe = false;
    }
    if (may_probe && !slot->ev && hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming) != hipSuccess) {
        slot->ev = nullptr;
        may_probe = false;
    }
    if (may_probe && launch) {
        asurf_order::probe(origins, dirs, n, d_probe, stream);
        prof_after("k_order_probe", 2, stream);
        if (hipMemcpyAsync(slot->h_probe, d_probe, 3 * asurf_order::kProbeBlocks * sizeof(int), hipMemcpyDeviceToHost, stream) == hipSuccess &&
            hipEventRecord(slot->ev, stream) == hipSuccess)
            slot->pending = true;
    }
    *one_origin = slot->one_origin;
    return slot->coherent;
}

// Common prologue: validation, workspace carve-up and the occupancy-pyramid pass.
int prepare(const asurf_grid* grid, const asurf_rays* rays, const asurf_render_options* opt,
            int32_t M, void* workspace, size_t workspace_bytes, void* stream, Launch& L,
            bool masks_only = false, bool cuvol = false, bool allow_lazy = false) {
    if (int e = check_grid(grid, !cuvol)) return e;
    if (int e = check_rays(rays)) return e;
    if (!opt) return fail(ASURF_E_INVALID, "render options are NULL");
    if (opt->alpha_activation_type != ASURF_SIGMOID_FN && opt->alpha_activation_type != ASURF_EXP_FN)
        return fail(ASURF_E_INVALID, "alpha_activation_type must be 0 (sigmoid) or 1 (exp)");
    L.ML = layout_fitting(grid->size, masks_only ? 0 : rays->n_rays, M, workspace ? workspace_bytes : 0);  // masks_only: no per-ray tables
    if (!workspace || workspace_bytes < L.ML.total)
        return fail(ASURF_E_WORKSPACE, "workspace of %zu bytes needed, got %zu", L.ML.total,
                    workspace ? workspace_bytes : (size_t)0);
    L.ws = static_cast<char*>(workspace);
    L.stream = static_cast<hipStream_t>(stream);
    // A voxel that holds no level set contributes nothing unless unrestricted fake sampling
    // is on (surf_trav.cu:423); and with stop_thresh > 1 the reference stops at the first
    // voxel that passes the link/density gates (:544), so the walk must test that plane.
    const bool need_surf = !(opt->surf_fake_sample && !opt->limited_fake_sample);
    const bool use_live = !cuvol && (!need_surf || (opt->stop_thresh > 1.f));
    L.g = make_kgrid(grid, L.ML, L.ws, use_live);
    L.o = make_kopt(opt);
    L.coarse_in_lds = (L.ML.coarse_words * 4 <= kMaxCoarseLds) ? 1 : 0;
    L.dyn_lds = L.coarse_in_lds ? L.ML.coarse_words * 4 : 0;
    L.ray_grid = dim3((unsigned)((rays->n_rays + kBlock - 1) / kBlock));
    L.sweep = !cuvol && !masks_only && want_sweep(use_live, L.o);
    L.lazy = false;
    prof_begin(L.stream);
    const int nzg = (L.ML.nb[2] + 7) / 8;  // groups of 8 bricks along z
    unsigned long long* fw = reinterpret_cast<unsigned long long*>(L.ws + L.ML.off_fine_work);
    unsigned long long* fl = reinterpret_cast<unsigned long long*>(L.ws + L.ML.off_fine_live);
    unsigned int* cw = reinterpret_cast<unsigned int*>(L.ws + L.ML.off_coarse_work);
    unsigned int* cl = reinterpret_cast<unsigned int*>(L.ws + L.ML.off_coarse_live);
    unsigned long long* fy = reinterpret_cast<unsigned long long*>(L.ws + L.ML.off_fine_y);
    unsigned long long* fz = reinterpret_cast<unsigned long long*>(L.ws + L.ML.off_fine_z);
    MaskHdr* hdr = reinterpret_cast<MaskHdr*>(L.ws + L.ML.off_mask_hdr);
    int* live_rows = reinterpret_cast<int*>(L.ws + L.ML.off_live_rows);
    const int mode = cuvol ? 2 : (need_surf ? 1 : 0);  // 2: cuvol cells (asurf_cuvol.h)
    // links_epoch (include/asurf.h): the rows of bricks that are not empty space are remembered in the workspace
    const unsigned long long epoch = cuvol ? 0ull : (unsigned long long)grid->links_epoch;
    // the pipeline's counters: zeroed by block 0 of the pass over the live rows, otherwise by the memset that also
    // clears the header in front of them
    int* zero_ptr = reinterpret_cast<int*>(L.ws + L.ML.off_p_counters);
    const int zero_n = kCtrTotal;
    const size_t hdr_and_counters = (L.ML.off_p_counters - L.ML.off_mask_hdr) + (size_t)kCtrTotal * sizeof(int);
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(L.stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    // ---- small batches: planes of `links` alone, kept from call to call (asurf_masks2.h) ----
    {
        const int cfg_lazy = config().lazy_rays;
        const long long lazy_max = cfg_lazy < 0 ? kLazyMaxRays : (long long)cfg_lazy;
        if (allow_lazy && L.sweep && !config().sweep_block && rays->n_rays > 0 && rays->n_rays <= lazy_max) {
            const int have = lazy_cache_lookup(workspace, grid, capturing);
            if (have >= 0) {
                if (have == 0 && hipMemsetAsync(hdr, 0, sizeof(MaskHdr), L.stream) != hipSuccess)
                    return fail(ASURF_E_LAUNCH, "hipMemsetAsync(mask header) failed");
                const long long n_links = (long long)grid->size[0] * grid->size[1] * grid->size[2];
                const int use_sum = epoch == 0ull ? 1 : 0;
                if (use_sum) {
                    // enough 16-byte pieces in flight to stream: 8 blocks of 256 a CU at most, 8 pieces a thread at least
                    const long long pieces = (n_links + 3) / 4;
                    const unsigned blocks = (unsigned)std::min<long long>(std::max<long long>(pieces / (256 * 8), 1), 2048);
                    hipLaunchKernelGGL(k_links_checksum, dim3(blocks), dim3(256), 0, L.stream, grid->links, n_links, hdr, 0);
                    prof_after("k_links_checksum", 0, L.stream);
                }
                hipLaunchKernelGGL(k_links_planes, dim3((unsigned)std::min(L.ML.n_rows, kLinksPlanesBlocks)), dim3(512), 0, L.stream, L.g, fw, fl,
                                   cw, cl, fy, fz, nzg, L.ML.n_rows, hdr, epoch, use_sum, zero_ptr, zero_n);
                L.lazy = true;
                L.o.lazy_gates = 1;
                prof_after("k_links_planes", 0, L.stream);
                return launch_error("k_links_planes");
            }
        }
    }
    int live_blocks = 0;
    // ---- the default call, nothing promised about `links`: a list of live rows the call vouches for ITSELF -- it sums
    // all of `links` (k_links_checksum: 64 MiB in 0.017 ms) and the pass over the listed rows (0.034 ms) checks on the
    // device that this is the sum the list was built under; 0.052 ms instead of the full pass's 0.063.  A mismatch (links
    // rewritten in place) makes that launch take every row and tells the host, which rebuilds list and sum with the
    // next call.  Grids the full pass takes in three steps (512^3 and up: their checksum alone is what the split pass
    // costs), captured calls (a replay cannot flip the parity of the slots) and the per-voxel forms keep the full pass.
    {
        const long long n_links = (long long)grid->size[0] * grid->size[1] * grid->size[2];
        const bool fast_sum = !cuvol && !(need_surf && grid->level_set_num > kMaskFastSets) && !config().masks_row;
        const int want_sum = config().sum_list;
        if (epoch == 0ull && fast_sum && !capturing && want_sum != 0 && L.ML.n_rows < kMaskSplitRows &&
            !config().masks_split && !config().masks_single && (want_sum == 1 || n_links >= (1ll << 21))) {
            const int n_sets_s = need_surf ? grid->level_set_num : 0;
            const long long pieces = (n_links + 3) / 4;
            const unsigned cblocks = (unsigned)std::min<long long>(std::max<long long>(pieces / (256 * 8), 1), 2048);
            int parity = 0;
            int* h_invalid = nullptr;
            if (sum_cache_lookup(workspace, grid, L.ML.n_rows, &live_blocks, &parity, &h_invalid) == 1) {
                hipLaunchKernelGGL(k_links_checksum, dim3(cblocks), dim3(256), 0, L.stream, grid->links, n_links, hdr, parity);
                prof_after("k_links_checksum", 0, L.stream);
                MaskRows R{hdr, live_rows, 0ull, L.ML.n_rows, 3, parity, h_invalid};
                hipLaunchKernelGGL(k_build_masks2, dim3((unsigned)live_blocks), dim3(512), 0, L.stream, L.g, opt->sigma_thresh,
                                   need_surf ? 1 : 0, n_sets_s, fw, fl, cw, cl, L.sweep ? fy : nullptr, L.sweep ? fz : nullptr,
                                   use_live ? 1 : 0, nzg, R, static_cast<MaskHdr*>(nullptr), static_cast<int*>(nullptr), zero_ptr,
                                   zero_n);
            } else {
                {   // (whatever the table held for this workspace is void: the planes are rewritten in full)
                    std::lock_guard<std::mutex> lock(g_mask_cache_mu);
                    for (MaskCacheEntry& e : g_mask_cache)
                        if (e.ws == workspace) e.ws = nullptr;
                }
                if (hipMemsetAsync(hdr, 0, hdr_and_counters, L.stream) != hipSuccess)
                    return fail(ASURF_E_LAUNCH, "hipMemsetAsync(mask header, pipeline counters) failed");
                hipLaunchKernelGGL(k_links_checksum, dim3(cblocks), dim3(256), 0, L.stream, grid->links, n_links, hdr, 0);
                const MaskRows R{hdr, live_rows, 0ull, L.ML.n_rows, 0};
                hipLaunchKernelGGL(k_build_masks2, dim3((unsigned)L.ML.n_rows), dim3(512), 0, L.stream, L.g, opt->sigma_thresh,
                                   need_surf ? 1 : 0, n_sets_s, fw, fl, cw, cl, fy, fz, use_live ? 1 : 0, nzg, R, hdr, live_rows,
                                   static_cast<int*>(nullptr), 0);
                hipLaunchKernelGGL(k_mask_seal_sum, dim3(1), dim3(64), 0, L.stream, hdr, 0);
                mask_cache_store(workspace, 0ull, grid, &hdr->n_live, L.stream, 2);
            }
            prof_after("k_build_masks2", 0, L.stream);
            return launch_error("k_build_masks");
        }
    }
    int cached = mask_cache_lookup(workspace, epoch, grid, L.ML.n_rows, &live_blocks, capturing);
    // building the list allocates pinned memory and reads a counter back: not while a graph is being captured
    if (cached == 0 && capturing) cached = -1;
    // From vertex bit planes (asurf_masks2.h) for the surf_trav planes with a short level-set list; the per-voxel kernel
    // for the cuvol plane, longer lists and ASURF_MASKS=row
    const bool fast = !cuvol && !(need_surf && grid->level_set_num > kMaskFastSets) && !config().masks_row;
    const int n_sets = need_surf ? grid->level_set_num : 0;
    if (cached == 1) {
        // (a header that is not what the host table says -- a captured full pass of another grid replayed over this
        // workspace since -- makes the kernels take every row)
        if (fast) {
            const MaskRows R{hdr, live_rows, epoch, L.ML.n_rows, 1};
            hipLaunchKernelGGL(k_build_masks2, dim3((unsigned)live_blocks), dim3(512), 0, L.stream, L.g, opt->sigma_thresh,
                               need_surf ? 1 : 0, n_sets, fw, fl, cw, cl, L.sweep ? fy : nullptr, L.sweep ? fz : nullptr,
                               use_live ? 1 : 0, nzg, R, static_cast<MaskHdr*>(nullptr), static_cast<int*>(nullptr), zero_ptr,
                               zero_n);
        } else {
            hipLaunchKernelGGL(k_build_masks_live, dim3((unsigned)live_blocks), dim3(512), 0, L.stream, L.g, opt->sigma_thresh,
                               mode, fw, fl, cw, cl, L.sweep ? fy : nullptr, L.sweep ? fz : nullptr, use_live ? 1 : 0, nzg, hdr,
                               live_rows, epoch, L.ML.n_rows, zero_ptr, zero_n);
        }
    } else {
        // a full pass rewrites every plane: whatever the header said is void from here on, on this call and on every
        // replay of a graph this call is captured into (the memset is a node of that graph)
        // Large grids take the pass in three steps (asurf_masks2.h, k_rows_alive): the planes, the header and the
        // counters lie one after the other and are cleared together; the rows that hold a vertex are listed; the
        // vertex-plane kernel takes the listed rows.
        const bool split = fast && (L.ML.n_rows >= kMaskSplitRows || config().masks_split) && (grid->size[2] & 3) == 0 &&
                           (reinterpret_cast<uintptr_t>(grid->links) & 15) == 0 && !config().masks_single;
        if (split) {
            const size_t all = (L.ML.off_mask_hdr - L.ML.off_fine_work) + hdr_and_counters;
            if (hipMemsetAsync(L.ws + L.ML.off_fine_work, 0, all, L.stream) != hipSuccess)
                return fail(ASURF_E_LAUNCH, "hipMemsetAsync(occupancy planes) failed");
            hipLaunchKernelGGL(k_rows_alive, dim3((unsigned)((L.ML.n_rows + 3) / 4)), dim3(256), 0, L.stream, L.g, nzg,
                               L.ML.n_rows, hdr, live_rows);
            const bool leave = cached == 0 && epoch != 0ull;
            const MaskRows R{hdr, live_rows, epoch, L.ML.n_rows, 2};
            hipLaunchKernelGGL(k_build_masks2, dim3((unsigned)std::min(L.ML.n_rows, 8192)), dim3(512), 0, L.stream, L.g,
                               opt->sigma_thresh, need_surf ? 1 : 0, n_sets, fw, fl, cw, cl,
                               (leave || L.sweep) ? fy : nullptr, (leave || L.sweep) ? fz : nullptr, use_live ? 1 : 0, nzg, R,
                               static_cast<MaskHdr*>(nullptr), static_cast<int*>(nullptr), static_cast<int*>(nullptr), 0);
            if (leave) {
                hipLaunchKernelGGL(k_mask_seal, dim3(1), dim3(1), 0, L.stream, hdr, epoch);
                mask_cache_store(workspace, epoch, grid, &hdr->n_live, L.stream);
            }
            prof_after("k_build_masks2", 0, L.stream);
            return launch_error("k_build_masks");
        }
        if (hipMemsetAsync(hdr, 0, hdr_and_counters, L.stream) != hipSuccess)
            return fail(ASURF_E_LAUNCH, "hipMemsetAsync(mask header, pipeline counters) failed");
        // first call with this value: the full pass leaves the list (and zeros in EVERY plane of the empty rows: the
        // y / z cuts are written whatever this call needs, the list outlives it)
        const bool leave_list = cached == 0 && epoch != 0ull;
        unsigned long long* fy_w = (leave_list || L.sweep) ? fy : nullptr;
        unsigned long long* fz_w = (leave_list || L.sweep) ? fz : nullptr;
        MaskHdr* hdr_w = leave_list ? hdr : nullptr;
        int* rows_w = leave_list ? live_rows : nullptr;
        if (fast) {
            const MaskRows R{hdr, live_rows, epoch, L.ML.n_rows, 0};
            hipLaunchKernelGGL(k_build_masks2, dim3((unsigned)L.ML.n_rows), dim3(512), 0, L.stream, L.g, opt->sigma_thresh,
                               need_surf ? 1 : 0, n_sets, fw, fl, cw, cl, fy_w, fz_w, use_live ? 1 : 0, nzg, R, hdr_w, rows_w,
                               static_cast<int*>(nullptr), 0);
        } else {
            hipLaunchKernelGGL(k_build_masks, dim3((unsigned)L.ML.n_rows), dim3(512), 0, L.stream, L.g, opt->sigma_thresh,
                               mode, fw, fl, cw, cl, fy_w, fz_w, use_live ? 1 : 0, nzg, hdr_w, rows_w);
        }
        if (leave_list) {
            hipLaunchKernelGGL(k_mask_seal, dim3(1), dim3(1), 0, L.stream, hdr, epoch);
            mask_cache_store(workspace, epoch, grid, &hdr->n_live, L.stream);
        }
    }
    prof_after(fast ? "k_build_masks2" : (cached == 1 ? "k_build_masks_live" : "k_build_masks"), 0, L.stream);
    return launch_error("k_build_masks");
}

// Blocks of the kernels that stride over a dense list whose length only the device knows: enough for the list a
// batch of n_rays can produce to be spread over the chip (a ray rarely holds more than 16 work voxels, the table
// keeps 32), but no more than kListBlocks -- 2048 blocks that find an empty list still cost their dispatch, which
// is most of these kernels' time at the training script's batch size.  (Grids of exactly the blocks the device keeps
// resident -- every wave starts at once and strides the same number of turns -- change nothing: these kernels run at
// the rate of their memory requests, not of their wave slots; variants/timeline.py.)
// Batches up to this size take the forms of k_geom / k_bwd_geo that are compiled without an occupancy bound (more
// registers, no scratch): below it the chip holds a wave or two a SIMD and a wave's own chain is what a kernel lasts.
constexpr long long kRelaxedMaxRays = 196608;   // (measured: a gain up to 160 k rays, even at 262 k, a loss at 640 k)
long long relaxed_max_rays() {
    const int n = config().relaxed_rays;   // ASURF_RELAXED = n (measurement; 0: never)
    return n < 0 ? kRelaxedMaxRays : (long long)n;
}

unsigned list_blocks(long long n_rays, int per_ray) {
    const long long want = (n_rays * per_ray + kBlock - 1) / kBlock;
    return (unsigned)std::min<long long>((std::max<long long>(want, 64) + 7) & ~7ll, kListBlocks);  // (k_geom: 8 groups)
}

// Rays a wave of the ray-per-lane kernels of the pipeline takes (k_comp, k_bwd_seq): 64 when the batch fills the chip
// several times over, fewer below -- a wave works its rays' samples off in turns, and with few waves the kernel is as
// slow as one of them.  ASURF_RPW = 4 ... 64 overrides (measurement).
int rays_per_wave(long long n_rays) {
    if (const int n = config().rpw) return n;
    // fused step, shuffled rays of the bench scene, 64 / 32 / 16 / 8 rays a wave: 5 k rays 0.176 / 0.165 / 0.160 / 0.159 ms,
    // 40 k 0.259 / 0.251 / 0.247 / 0.249, 160 k 0.541 / 0.541 / 0.557
    if (n_rays <= 8192) return 8;
    if (n_rays <= 65536) return 16;
    if (n_rays <= 262144) return 32;
    return 64;
}

// The three-kernel forward is used unless the call needs what only the one-kernel path keeps:
// exact V_t/V_a/S counters (a ray stopped by stop_thresh must not be walked further),
// stop_thresh > 1 (the reference then stops at the first voxel that passes the gates even if
// it holds no sample), or more level sets than a record encodes.
bool use_pipeline(const asurf_grid* grid, const asurf_rays* rays, const asurf_render_options* opt,
                  const uint64_t* counters) {
    if (config().no_pipeline) return false;
    return counters == nullptr && !(opt->stop_thresh > 1.f) && grid->level_set_num <= 32 &&
           rays->n_rays < (1ll << kListRayBits) &&
           grid->size[0] <= 1024 && grid->size[1] <= 1024 && grid->size[2] <= 1024;
}

// The first two stages of the forward pipeline: the work voxels of every ray (k_sweep, or the stepping walk) and
// their sample candidates (k_geom), left in the per-ray tables and the candidate pool that `P` names.
int find_candidates(const Launch& L, const asurf_rays* rays, bool record, PipeBuf& P) {
    const long long Q = rays->n_rays;
    {
        P.items = reinterpret_cast<int*>(L.ws + L.ML.off_p_items);
        P.citem = reinterpret_cast<int*>(L.ws + L.ML.off_p_citem);
        P.n_items = reinterpret_cast<int*>(L.ws + L.ML.off_p_nitems);
        P.cut = reinterpret_cast<int*>(L.ws + L.ML.off_p_cut);
        P.flagged = reinterpret_cast<int*>(L.ws + L.ML.off_p_flagged);
        P.counters = reinterpret_cast<int*>(L.ws + L.ML.off_p_counters);
        P.sp_base = reinterpret_cast<int*>(L.ws + L.ML.off_p_spbase);
        P.sp_items = reinterpret_cast<int*>(L.ws + L.ML.off_p_spill);
        P.sp_citem = P.sp_items + L.ML.sp_cap;
        P.sp_ray = P.sp_citem + L.ML.sp_cap;
        P.sp_cap = L.ML.sp_cap;
        P.pool = reinterpret_cast<CandRec*>(L.ws + L.ML.off_p_pool);
        P.cand_mask = reinterpret_cast<unsigned long long*>(L.ws + L.ML.off_p_cand);
        P.shard_cap = L.ML.pool_shard_cap;
        P.C1 = kPipeC1;
        P.list = reinterpret_cast<unsigned*>(L.ws + L.ML.off_p_list);
        P.list_cap = L.ML.list_cap;
        P.band_rays = L.ML.band_rays;
        // (P.counters: zeroed in prepare(), with the occupancy pass of this call)
        const int seg = walk_segments(Q);
        const Config cfgc = config();
        if (L.sweep) {
            int nbmax = 1;
            for (int k = 0; k < 3; ++k) nbmax = std::max(nbmax, ((L.g.size[k] - 2) >> 3) + 1);
            int lpr_log2 = 0;
            while ((1 << lpr_log2) < nbmax) ++lpr_log2;
            // rays per block from the (ray, brick layer) pairs a block takes; threads from the measured optimum (the
            // pairs are strided over them): 256 threads x 16 rays at 256^3 -- with 512 the phases that occupy one
            // wave leave seven waiting (0.50 ms against 0.35), with 128 a block takes too many rounds (0.39)
            // 16 rays per block whatever the grid: the block's fixed costs -- barriers, set-up, allocation -- are per
            // block (8 rays at 512^3 and 768^3, as the 512 pairs of round 2's first half allowed: k_sweep 0.62 against
            // 0.51 ms at 512^3, 3.39 against 3.05 at 768^3)
            if (cfgc.sweep_block) {
            const int rpb = 16;
            const int sweep_threads = std::min(kSweepThreads, std::max(64, rpb << lpr_log2));
            // at least 128 keys per ray: a ray that finds more is handed to the one-kernel path, whose latency
            // (that of one wave walking a whole ray) is then the call's
            const int keys = std::max(kSweepKeys, rpb * (lpr_log2 >= 7 ? 256 : 128));  // (longer rays on larger grids)
            const int K = keys / rpb;
            const int task_cap = std::min((rpb << lpr_log2) * kSweepTasksPerLane2 / 2, 2560);  // (20 KB; k_sweep takes fewer rays a pass beyond)
            const size_t lds = (size_t)rpb * sizeof(SweepRay) + (size_t)task_cap * 8 + (size_t)keys * 4;
            const dim3 sgrid((unsigned)((Q + rpb - 1) / rpb));
            hipLaunchKernelGGL(k_sweep, sgrid, dim3(sweep_threads), lds, L.stream, L.g, L.o, rays->origins,
                               rays->dirs, Q, P, lpr_log2, rpb, K, record ? 1 : 0, task_cap, nbmax);
            } else {
                // the wave-autonomous form: a wave owns 16 rays; its LDS slice holds their set-up, a task list and a
                // key buffer shared by the rays (asurf_sweep.h)
                // rays per wave: 16 when the batch fills the chip several times over; fewer below, where the kernel
                // lasts as long as ONE wave's chain of phases (k_sweep at 5 184 rays, 16 / 8 / 4 / 2 rays a wave: 0.035 / 0.024 / 0.021 / 0.021 ms;
                // at 65 536 rays 0.052 / 0.054 / 0.064 / 0.091)
                const int rpw = cfgc.sweep_rpw ? cfgc.sweep_rpw : (Q <= 8192 ? 4 : (Q <= 32768 ? 8 : kSweepWaveRays));
                // (the slice decides how many waves share a CU: LDS is handed out in 1 280-byte pieces, and at 256^3 a
                // slice of 7 504 bytes lets the 20 waves the registers allow in, 8 016 -- 384 tasks -- only 18:
                // 0.279 against 0.296 ms; a wave with more tasks than fit takes its rays four at a time)
                // (over planes of links alone a ray meets every voxel of the shell it crosses, five times the work voxels
                // on the bench scene: four times the keys, twice the tasks)
                const int lazy_tm = 2, lazy_km = 4;
                const int task_cap = (std::min(std::max(rpw * nbmax / 2, 128) * (L.lazy ? lazy_tm : 1), 4096) + 3) & ~3;
                const int key_cap = (std::max((nbmax > 64 ? 2048 : (nbmax > 32 ? 1280 : 768)) * rpw / kSweepWaveRays, 256) * (L.lazy ? lazy_km : 1) + 15) & ~15;
                const int wave_bytes = (int)((size_t)rpw * sizeof(SweepRay) + (size_t)task_cap * 4 + (size_t)key_cap * 5 +
                                             sizeof(SweepWaveMisc));   // (keys + a byte each for the ray they belong to)
                const long long waves = (Q + rpw - 1) / rpw;
                const int wpb = cfgc.sweep_wpb ? cfgc.sweep_wpb : kSweepWavesPerBlock;
                const dim3 sgrid((unsigned)((waves + wpb - 1) / wpb));
                if (L.lazy)
                    hipLaunchKernelGGL(k_sweep_wave_lazy, sgrid, dim3(64 * wpb), (size_t)wave_bytes * wpb,
                                       L.stream, L.g, L.o, rays->origins, rays->dirs, Q, P, nbmax, record ? 1 : 0, task_cap,
                                       key_cap, wave_bytes, rpw);
                else
                hipLaunchKernelGGL(k_sweep_wave, sgrid, dim3(64 * wpb), (size_t)wave_bytes * wpb,
                                   L.stream, L.g, L.o, rays->origins, rays->dirs, Q, P, nbmax, record ? 1 : 0, task_cap,
                                   key_cap, wave_bytes, rpw);
            }
        } else if (seg > 1) {
            const dim3 seg_grid((unsigned)((Q * seg + kBlock - 1) / kBlock));
#define ASURF_WALK_SEG(S_)                                                                                   \
    if (L.coarse_in_lds)                                                                                     \
        hipLaunchKernelGGL((k_walk_seg<true, S_>), seg_grid, dim3(kBlock), L.dyn_lds, L.stream, L.g, L.o,    \
                           rays->origins, rays->dirs, Q, P, record ? 1 : 0);                                 \
    else                                                                                                     \
        hipLaunchKernelGGL((k_walk_seg<false, S_>), seg_grid, dim3(kBlock), 0, L.stream, L.g, L.o,           \
                           rays->origins, rays->dirs, Q, P, record ? 1 : 0)
            if (seg == 2) { ASURF_WALK_SEG(2); }
            else if (seg == 4) { ASURF_WALK_SEG(4); }
            else { ASURF_WALK_SEG(8); }
#undef ASURF_WALK_SEG
        } else if (L.coarse_in_lds)
            hipLaunchKernelGGL((k_walk<true>), L.ray_grid, dim3(kBlock), L.dyn_lds, L.stream, L.g, L.o,
                               rays->origins, rays->dirs, Q, P, record ? 1 : 0);
        else
            hipLaunchKernelGGL((k_walk<false>), L.ray_grid, dim3(kBlock), 0, L.stream, L.g, L.o,
                               rays->origins, rays->dirs, Q, P, record ? 1 : 0);
        prof_after(L.sweep ? "k_sweep" : "k_walk", 1, L.stream);
        // (with fake samples every work voxel passes to the solver stage unseen: the register-rich form wins at any batch
        // size -- 0.872 -> 0.787 ms on the heavy fake scene --, unless the switch says never)
        if (Q <= relaxed_max_rays() || (L.o.surf_fake_sample && relaxed_max_rays() > 0))
            hipLaunchKernelGGL(k_geom<true>, dim3(list_blocks(Q, 16)), dim3(kBlock), 0, L.stream, L.g, L.o, rays->origins,
                               rays->dirs, Q, P);
        else
            hipLaunchKernelGGL(k_geom<false>, dim3(list_blocks(Q, 16)), dim3(kBlock), 0, L.stream, L.g, L.o, rays->origins,
                               rays->dirs, Q, P);
        prof_after("k_geom", 1, L.stream);
    }
    return launch_error("forward pipeline");
}

// The one-kernel forward.  `pipeline` = the three-kernel pipeline runs first and this kernel
// only renders the rays it handed back (k_forward's ray_filter).
// A stream of the library's own, per host thread, for work of a call that nothing else in the call waits for (the fused
// call's one-kernel fallbacks); forked from and joined to the caller's stream with events, so the call stays stream-ordered
// for its caller and capturable into a graph.
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int device = -1;
};
SideStream* side_stream() {
    thread_local SideStream tl[8];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    SideStream& s = tl[dev & 7];
    if (s.stream == nullptr || s.device != dev) {
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&s.ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s.ev_join, hipEventDisableTiming) != hipSuccess) {
            s.stream = nullptr;
            (void)hipGetLastError();
            return nullptr;
        }
        s.device = dev;
    }
    return &s;
}

// `fb`: where the one-kernel forward of the handed-back rays is launched when the pipeline ran first -- nullptr: on the
// call's stream, leaving records like the pipeline; else on that launch context's stream (the fused call's side stream)
// and WITHOUT records, so that nothing on the call's own stream waits for it.
template <int B>
int run_forward(const Launch& L, const asurf_rays* rays, float* rgb_out, int M, bool cache,
                bool record, uint64_t* counters, bool pipeline, const Launch* fb = nullptr, SideStream* side = nullptr) {
    float* ca = reinterpret_cast<float*>(L.ws + L.ML.off_cache_a);
    float* cw = reinterpret_cast<float*>(L.ws + L.ML.off_cache_w);
    float* ct = reinterpret_cast<float*>(L.ws + L.ML.off_cache_t);
    int* cn = reinterpret_cast<int*>(L.ws + L.ML.off_cache_n);
    int* rv = record ? reinterpret_cast<int*>(L.ws + L.ML.off_rec_v) : nullptr;
    int* rm = record ? reinterpret_cast<int*>(L.ws + L.ML.off_rec_m) : nullptr;
    int* rn = record ? reinterpret_cast<int*>(L.ws + L.ML.off_rec_n) : nullptr;
    float* rf = record ? reinterpret_cast<float*>(L.ws + L.ML.off_rec_f) : nullptr;
    auto* cnt = reinterpret_cast<unsigned long long*>(counters);
    const long long Q = rays->n_rays;
    const int* ray_list = nullptr;
    const int* n_list = nullptr;
    if (pipeline) {
        PipeBuf P;
        if (int e = find_candidates(L, rays, record, P)) return e;
        const int rpw = rays_per_wave(Q);
        const dim3 comp_grid((unsigned)((Q + (kBlock / 64) * rpw - 1) / ((kBlock / 64) * rpw)));
        if (cache)
            hipLaunchKernelGGL((k_comp<B, true>), comp_grid, dim3(kBlock), 0, L.stream, L.g, L.o,
                               rays->dirs, Q, P, rgb_out, M, ca, cw, ct, cn, rv, rm, rn, L.ML.rec_depth, rf, rpw);
        else
            hipLaunchKernelGGL((k_comp<B, false>), comp_grid, dim3(kBlock), 0, L.stream, L.g, L.o,
                               rays->dirs, Q, P, rgb_out, M, ca, cw, ct, cn, rv, rm, rn, L.ML.rec_depth, rf, rpw);
        prof_after("k_comp", 1, L.stream);
        if (int e = launch_error("forward pipeline")) return e;
        if (L.o.debug & 16) {  // diagnostics: what the pipeline's counters hold after this call
            int h[kCtrTotal];
            if (hipStreamSynchronize(L.stream) == hipSuccess &&
                hipMemcpy(h, P.counters, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
                long long pool = 0;
                for (int i = 0; i < kPoolShards; ++i) pool += h[2 + i];
                fprintf(stderr, "[asurf] rays %lld: handed back %d, spilled voxels %d, table cells %d, candidates %lld\n",
                        Q, h[0], h[1], [&] { int t = 0; for (int i = 0; i < kListSegs; ++i) t += h[kCtrListShard0 + i * kCtrShardStride]; return t; }(), pool);
            }
        }
        ray_list = P.flagged;
        n_list = P.counters;
    }
    hipStream_t fstream = L.stream;
    if (pipeline && fb != nullptr && side != nullptr) {
        // fork: the side stream continues from here (k_comp and everything before it are queued on the call's stream)
        if (hipEventRecord(side->ev_fork, L.stream) != hipSuccess ||
            hipStreamWaitEvent(side->stream, side->ev_fork, 0) != hipSuccess)
            return fail(ASURF_E_LAUNCH, "side stream fork failed");
        fstream = fb->stream;
        rv = nullptr;  // (k_forward records iff rec_v != nullptr)
    }
#define ASURF_FWD(CACHE, COUNT, CLDS)                                                            \
    hipLaunchKernelGGL((k_forward<B, CACHE, COUNT, CLDS>), L.ray_grid, dim3(kBlock), L.dyn_lds,  \
                       fstream, L.g, L.o, rays->origins, rays->dirs, (long long)rays->n_rays,    \
                       rgb_out, M, ca, cw, ct, cn, rv, rm, rn, L.ML.rec_depth, rf, ray_list, n_list, cnt)
#define ASURF_FWD2(CACHE, COUNT)             \
    do {                                     \
        if (L.coarse_in_lds) ASURF_FWD(CACHE, COUNT, true); \
        else ASURF_FWD(CACHE, COUNT, false); \
    } while (0)
    if (cache) {
        if (counters) ASURF_FWD2(true, true);
        else ASURF_FWD2(true, false);
    } else {
        if (counters) ASURF_FWD2(false, true);
        else ASURF_FWD2(false, false);
    }
#undef ASURF_FWD2
#undef ASURF_FWD
    if (fstream == L.stream) prof_after(pipeline ? "k_forward_fallback" : "k_forward", 1, L.stream);
    return launch_error("k_forward");
}

// mode 0: walking backward over all rays; 1: record replay in one kernel (fused only);
// 2: walking backward restricted to the rays whose record overflowed (fused only);
// 3: record replay as three kernels (asurf_bwd_pipeline.h)
template <int B>
int run_backward(const Launch& L, const asurf_rays* rays, const KLoss& loss, const KGrads& G,
                 const float* grad_in, const float* color_cache, int M, bool fused, int mode,
                 bool counters_zeroed = false, int image_width = 0, bool pipeline_ran = false) {
    const float* ca = reinterpret_cast<const float*>(L.ws + L.ML.off_cache_a);
    const float* cw = reinterpret_cast<const float*>(L.ws + L.ML.off_cache_w);
    const float* ct = reinterpret_cast<const float*>(L.ws + L.ML.off_cache_t);
    const int* cn = reinterpret_cast<const int*>(L.ws + L.ML.off_cache_n);
    const int* rv = reinterpret_cast<const int*>(L.ws + L.ML.off_rec_v);
    const int* rm = reinterpret_cast<const int*>(L.ws + L.ML.off_rec_m);
    const int* rn = reinterpret_cast<const int*>(L.ws + L.ML.off_rec_n);
    const int* filter = (mode == 2) ? rn : nullptr;
    // (mode 2 behind the pipeline: the counters that say whether any ray was left to this launch)
    const int* ctr = reinterpret_cast<const int*>(L.ws + L.ML.off_p_counters);
    const int* any_a = (mode == 2 && pipeline_ran) ? ctr : nullptr;
    const int* any_b = (mode == 2 && pipeline_ran) ? ctr + kCtrRecOver : nullptr;
#define ASURF_BWD(FUSED, CLDS, REC)                                                               \
    hipLaunchKernelGGL((k_backward<B, FUSED, CLDS, REC>), L.ray_grid, dim3(kBlock),               \
                       (REC) ? 0 : L.dyn_lds, L.stream, L.g, L.o, loss, G, rays->origins,          \
                       rays->dirs, (long long)rays->n_rays, grad_in, color_cache, M, ca, cw, ct,  \
                       cn, rv, rm, rn, filter, any_a, any_b)
    if (mode == 3) {
        // split replay: sequential scalars per ray, then one sample per wave step (SH) and
        // one sample per lane (geometry)
        SampleRec S;
        const size_t plane = (size_t)L.ML.rec_depth * (size_t)rays->n_rays;
        float* f = reinterpret_cast<float*>(L.ws + L.ML.off_rec_f);
        S.v = const_cast<int*>(rv);
        S.m = const_cast<int*>(rm);
        S.n = const_cast<int*>(rn);
        S.raw = f;
        S.a = f + plane;
        S.c0 = f + 2 * plane;
        S.c1 = f + 3 * plane;
        S.c2 = f + 4 * plane;
        S.px = f + 5 * plane;
        S.py = f + 6 * plane;
        S.pz = f + 7 * plane;
        S.w = f + 8 * plane;
        S.g0 = f + 9 * plane;
        S.g1 = f + 10 * plane;
        S.g2 = f + 11 * plane;
        S.cgr = f + 12 * plane;
        S.gst = f + 13 * plane;
        S.gx = f + 14 * plane;
        S.gy = f + 15 * plane;
        S.gz = f + 16 * plane;
        S.list = reinterpret_cast<unsigned*>(L.ws + L.ML.off_rec_list);
        S.n_list = reinterpret_cast<int*>(L.ws + L.ML.off_p_counters) + kCtrRecShard0;
        S.list_cap = L.ML.rec_list_cap;
        S.n_eff = reinterpret_cast<int*>(L.ws + L.ML.off_rec_neff);
        S.sws = reinterpret_cast<float*>(L.ws + L.ML.off_rec_sws);
        S.R = L.ML.rec_depth;
        const long long Q = rays->n_rays;
        // (the list's cursor lies among the pipeline's counters: the forward of this call has zeroed it if it ran the
        // pipeline -- a memset node of its own is 4 us on the stream)
        if (!counters_zeroed &&
            hipMemsetAsync(S.n_list, 0, kRecShards * kCtrShardStride * sizeof(int), L.stream) != hipSuccess)
            return fail(ASURF_E_LAUNCH, "hipMemsetAsync failed");
        const int rpw = rays_per_wave(Q);
        // an image batch in row order: the record list in 2-D tiles (k_bwd_seq)
        const int ttw = config().bwd_tile;
        const bool tiled = image_width > 0 && ttw > 0 && rpw == 64 && image_width % ttw == 0 && Q % image_width == 0 &&
                           (Q / image_width) % (64 / ttw) == 0;
        hipLaunchKernelGGL(k_bwd_seq, dim3((unsigned)((Q + (kBlock / 64) * rpw - 1) / ((kBlock / 64) * rpw))), dim3(kBlock), 0,
                           L.stream, L.o, loss, L.g.trunc_a, Q, grad_in, color_cache, M, ca, cw, ct, cn, S, rpw,
                           tiled ? image_width : 0, ttw);
        prof_after("k_bwd_seq", 2, L.stream);
        if (!config().bwd_sh_tab)
        hipLaunchKernelGGL((k_bwd_sh<B>), dim3(list_blocks(Q, 8)), dim3(kBlock), 0, L.stream, L.g, G,
                           L.o.no_surf_grad_from_sh ? 0 : 1, rays->dirs, Q, S, config().bwd_sort);
        else   // (the same number of waves: its blocks are half as large)
            hipLaunchKernelGGL((k_bwd_sh_tab<B>), dim3(list_blocks(Q, 8) * (kBlock / BwdShTab<B>::kThreads)),
                               dim3(BwdShTab<B>::kThreads), 0, L.stream, L.g, G, L.o.no_surf_grad_from_sh ? 0 : 1,
                               rays->dirs, Q, S);
        prof_after("k_bwd_sh", 2, L.stream);
        if (Q <= relaxed_max_rays())
            hipLaunchKernelGGL(k_bwd_geo<true>, dim3(list_blocks(Q, 2)), dim3(kBlock), 0, L.stream, L.g, L.o, loss, G,
                               rays->origins, rays->dirs, Q, S, config().bwd_sort);
        else
            hipLaunchKernelGGL(k_bwd_geo<false>, dim3(list_blocks(Q, 2)), dim3(kBlock), 0, L.stream, L.g, L.o, loss, G,
                               rays->origins, rays->dirs, Q, S, config().bwd_sort);
        prof_after("k_bwd_geo", 2, L.stream);
    } else if (mode == 1) {
        ASURF_BWD(true, false, true);
        prof_after("k_backward_replay", 2, L.stream);
    } else if (fused) {
        if (L.coarse_in_lds) ASURF_BWD(true, true, false);
        else ASURF_BWD(true, false, false);
    } else {
        if (L.coarse_in_lds) ASURF_BWD(false, true, false);
        else ASURF_BWD(false, false, false);
    }
#undef ASURF_BWD
    if (mode != 1 && mode != 3) prof_after(mode == 2 ? "k_backward_fallback" : "k_backward", 2, L.stream);
    return launch_error("k_backward");
}

#define ASURF_DISPATCH_B(basis_dim, CALL)               \
    switch (basis_dim) {                                \
        case 1: { constexpr int B_ = 1; rc = CALL; } break;   \
        case 4: { constexpr int B_ = 4; rc = CALL; } break;   \
        case 9: { constexpr int B_ = 9; rc = CALL; } break;   \
        default: { constexpr int B_ = 16; rc = CALL; } break; \
    }

KGrads make_grads(const asurf_grads* g) {
    KGrads k;
    k.grad_density = g->grad_density_out;
    k.grad_sh = g->grad_sh_out;
    k.grad_surface = g->grad_surface_out;
    k.grad_fake_sample_std = g->grad_fake_sample_std_out;
    k.mask = g->mask_out;
    k.mask2 = g->mask_sh_out;
    return k;
}

// the legacy backend walks caller-supplied voxel lists: no occupancy pyramid
KGrid make_kgrid_plain(const asurf_grid* g) {
    KGrid k;
    k.links = g->links;
    k.density = g->density_data;
    k.surface = g->surface_data;
    k.sh = g->sh_data;
    k.level_set = g->level_set_data;
    for (int i = 0; i < 3; ++i) {
        k.size[i] = g->size[i];
        k.offset[i] = g->offset[i];
        k.scaling[i] = g->scaling[i];
        k.nb[i] = 0;
    }
    k.offx = g->size[1] * g->size[2];
    k.offy = g->size[2];
    k.level_set_num = g->level_set_num;
    k.fake_sample_std = g->fake_sample_std;
    k.trunc_a = g->truncated_vol_render_a;
    k.fine = nullptr;
    k.fine_live = nullptr;
    k.coarse = nullptr;
    k.nwz = 0;
    k.coarse_words = 0;
    return k;
}

int check_legacy(const asurf_grid* grid, const asurf_rays* rays, const asurf_ray_vox* rv,
                 const asurf_render_options* opt) {
    if (int e = check_grid(grid)) return e;
    if (int e = check_rays(rays)) return e;
    if (!opt) return fail(ASURF_E_INVALID, "opt is NULL");
    if (!rv) return fail(ASURF_E_INVALID, "ray_vox is NULL");
    if (rv->n_vox < 0) return fail(ASURF_E_INVALID, "ray_vox.n_vox < 0");
    if (rays->n_rays > 0 && (!rv->vox_start_i || !rv->vox_num))
        return fail(ASURF_E_INVALID, "ray_vox.vox_start_i / vox_num is NULL");
    if (rv->n_vox > 0 && !rv->voxel_ls) return fail(ASURF_E_INVALID, "ray_vox.voxel_ls is NULL");
    if (grid->level_set_num < 1 || !grid->level_set_data)
        return fail(ASURF_E_INVALID, "the surface backend needs level_set_data[0]");
    return 0;
}

template <int B>
int leg_fwd(dim3 gd, hipStream_t s, const KGrid& g, const KOpt& o, const asurf_rays* rays,
            const asurf_ray_vox* rv, float* rgb_out) {
    hipLaunchKernelGGL((k_surface_fwd<B>), gd, dim3(kBlock), 0, s, g, o, rays->origins, rays->dirs,
                       (long long)rays->n_rays, rv->voxel_ls, rv->vox_start_i, rv->vox_num,
                       (long long)rv->n_vox, rgb_out);
    return 0;
}

template <int B>
int leg_bwd(dim3 gd, hipStream_t s, const KGrid& g, const KOpt& o, const KGrads& G, const asurf_rays* rays,
            const asurf_ray_vox* rv, const float* grad_in, const float* color_cache, int grad_in_is_rgb,
            float sparsity_loss) {
    hipLaunchKernelGGL((k_surface_bwd<B>), gd, dim3(kBlock), 0, s, g, o, G, rays->origins, rays->dirs,
                       (long long)rays->n_rays, rv->voxel_ls, rv->vox_start_i, rv->vox_num,
                       (long long)rv->n_vox, grad_in, color_cache, grad_in_is_rgb, sparsity_loss);
    return 0;
}

template <int MODE>
int launch_eval(const Launch& L, const asurf_rays* rays, float param, int max_sample, float* out0, float* out1,
                bool pipeline) {
    const long long Q = rays->n_rays;
    const int* ray_list = nullptr;
    const int* n_list = nullptr;
    if (pipeline) {
        PipeBuf P;
        if (int e = find_candidates(L, rays, false, P)) return e;
        hipLaunchKernelGGL((k_eval_comp<MODE>), L.ray_grid, dim3(kBlock), 0, L.stream, L.g, L.o, rays->dirs, Q, P, param,
                           max_sample, out0, out1);
        prof_after("k_eval_comp", 1, L.stream);
        ray_list = P.flagged;
        n_list = P.counters;
    }
    if (L.coarse_in_lds)
        hipLaunchKernelGGL((k_eval<MODE, true>), L.ray_grid, dim3(kBlock), L.dyn_lds, L.stream, L.g, L.o,
                           rays->origins, rays->dirs, Q, param, max_sample, out0, out1, ray_list, n_list);
    else
        hipLaunchKernelGGL((k_eval<MODE, false>), L.ray_grid, dim3(kBlock), 0, L.stream, L.g, L.o,
                           rays->origins, rays->dirs, Q, param, max_sample, out0, out1, ray_list, n_list);
    prof_after(pipeline ? "k_eval_fallback" : "k_eval", 1, L.stream);
    return launch_error("k_eval");
}

// Rays per wave of the cuvol kernels (asurf_cuvol.h).  Fused call on the bench scene, ms: 5 k rays 1.40 / 1.50 / 2.10 /
// 2.89 / 8.07 with 1 / 2 / 4 / 8 / 64 rays per wave; 20 k rays 3.75 / 2.92 / 2.89 / 3.15 / 9.27 (1 / 2 / 4 / 8 / 64);
// 80 k 6.26 / 6.48 / 7.06 / 9.86 (8 / 16 / 32 / 64); 160 k 10.8 / 10.3 / 11.0 / 11.7; 320 k 17.6 / 18.7 (32 / 64);
// 640 k 32.4 / 31.9.  ASURF_CUVOL_RPW = 1 .. 64 (a power of two) overrides.
static int cuvol_rays_per_wave(long long n_rays) {
    if (const int n = config().cuvol_rpw) return n;
    if (n_rays <= 6000) return 1;
    if (n_rays <= 12000) return 2;
    if (n_rays <= 30000) return 4;
    if (n_rays <= 100000) return 8;
    if (n_rays <= 200000) return 16;
    if (n_rays <= 400000) return 32;
    return 64;
}

// asurf_rays::image_width: 8 x 8 pixel tiles a wave instead of 64 pixels of a row (asurf_cuvol.h), where they partition
// the image and the batch fills the chip
static int cuvol_tile_width(const asurf_rays* rays, int rpw) {
    const long long W = rays->image_width, Q = rays->n_rays;
    const int tw = config().cuvol_tile;   // 0: rows; 4 / 8 / 16: the tile's width
    if (W <= 0 || W >= (1 << 24) || rpw != 64 || tw == 0 || W % tw != 0 || Q % W != 0 || (Q / W) % (64 / tw) != 0) return 0;
    return (int)W | (tw << 24);
}

static dim3 cuvol_grid(long long n_rays, int rpw) {
    const long long per_block = (long long)rpw * (kBlock / 64);
    return dim3((unsigned)((n_rays + per_block - 1) / per_block));
}

template <int B>
int cuvol_fwd(const Launch& L, const asurf_rays* rays, int last_opaque, float* rgb_out, float* log_t) {
    const int rpw = cuvol_rays_per_wave(rays->n_rays);
    const dim3 grid = cuvol_grid(rays->n_rays, rpw);
    if (L.coarse_in_lds)
        hipLaunchKernelGGL((k_cuvol_fwd<B, true>), grid, dim3(kBlock), L.dyn_lds, L.stream, L.g, L.o,
                           last_opaque, rpw, rays->origins, rays->dirs, (long long)rays->n_rays, rgb_out, log_t,
                           cuvol_tile_width(rays, rpw));
    else
        hipLaunchKernelGGL((k_cuvol_fwd<B, false>), grid, dim3(kBlock), 0, L.stream, L.g, L.o,
                           last_opaque, rpw, rays->origins, rays->dirs, (long long)rays->n_rays, rgb_out, log_t,
                           cuvol_tile_width(rays, rpw));
    prof_after("k_cuvol_fwd", 1, L.stream);
    return launch_error("k_cuvol_fwd");
}

template <int B>
int cuvol_bwd(const Launch& L, const asurf_rays* rays, int last_opaque, const KGrads& G, const float* grad_in,
              const float* color_cache, int is_rgb, const float* log_t, float beta_loss, float sparsity_loss) {
    const int rpw = cuvol_rays_per_wave(rays->n_rays);
    const dim3 grid = cuvol_grid(rays->n_rays, rpw);
    if (L.coarse_in_lds)
        hipLaunchKernelGGL((k_cuvol_bwd<B, true>), grid, dim3(kBlock), L.dyn_lds, L.stream, L.g, L.o,
                           last_opaque, rpw, G, rays->origins, rays->dirs, (long long)rays->n_rays, grad_in,
                           color_cache, is_rgb, log_t, beta_loss, sparsity_loss, cuvol_tile_width(rays, rpw));
    else
        hipLaunchKernelGGL((k_cuvol_bwd<B, false>), grid, dim3(kBlock), 0, L.stream, L.g, L.o,
                           last_opaque, rpw, G, rays->origins, rays->dirs, (long long)rays->n_rays, grad_in,
                           color_cache, is_rgb, log_t, beta_loss, sparsity_loss, cuvol_tile_width(rays, rpw));
    prof_after("k_cuvol_bwd", 2, L.stream);
    return launch_error("k_cuvol_bwd");
}

template <int MODE>
int cuvol_eval(const Launch& L, const asurf_rays* rays, float param, int max_sample, float* out0, float* out1) {
    if (L.coarse_in_lds)
        hipLaunchKernelGGL((k_cuvol_eval<MODE, true>), L.ray_grid, dim3(kBlock), L.dyn_lds, L.stream, L.g, L.o,
                           rays->origins, rays->dirs, (long long)rays->n_rays, param, max_sample, out0, out1);
    else
        hipLaunchKernelGGL((k_cuvol_eval<MODE, false>), L.ray_grid, dim3(kBlock), 0, L.stream, L.g, L.o,
                           rays->origins, rays->dirs, (long long)rays->n_rays, param, max_sample, out0, out1);
    prof_after("k_cuvol_eval", 1, L.stream);
    return launch_error("k_cuvol_eval");
}

// test hook: walk_hier's quotient() next to IEEE division, element by element
__global__ void __launch_bounds__(256)
k_test_quotient(const float* __restrict__ x, const float* __restrict__ d, long long n, float* __restrict__ fast,
                float* __restrict__ ieee) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float rd = 1.f / d[i];
    fast[i] = quotient(x[i], d[i], rd);
    ieee[i] = x[i] / d[i];
}

}  // namespace

extern "C" {

int asurf_abi_version(void) { return ASURF_ABI_VERSION; }
const char* asurf_last_error(void) { return g_err; }

size_t asurf_workspace_bytes(const int32_t size[3], int64_t n_rays, int32_t l_dist_max_sample) {
    if (!size) return 0;
    return layout_for(size, n_rays, l_dist_max_sample).total;
}

size_t asurf_workspace_bytes_rec(const int32_t size[3], int64_t n_rays, int32_t l_dist_max_sample, int32_t rec_depth) {
    if (!size) return 0;
    int depth = kRecDepths[0];
    for (int d : kRecDepths)
        if (d >= rec_depth) depth = d;   // the smallest supported depth that holds rec_depth (the deepest beyond)
    return layout_for(size, n_rays, l_dist_max_sample, depth).total;
}

int asurf_volume_render_surf_trav(const asurf_grid* grid, const asurf_rays* rays,
                                  const asurf_render_options* opt, float* rgb_out, void* workspace,
                                  size_t workspace_bytes, uint64_t* counters, void* stream) {
    Launch L;
    if (int e = check_grid(grid)) return e;
    if (int e = check_rays(rays)) return e;
    if (!opt) return fail(ASURF_E_INVALID, "render options are NULL");
    const bool pipe = use_pipeline(grid, rays, opt, counters);
    if (int e = prepare(grid, rays, opt, 0, workspace, workspace_bytes, stream, L, false, false, pipe)) return e;
    if (rays->n_rays == 0) return 0;
    if (!rgb_out) return fail(ASURF_E_INVALID, "rgb_out is NULL");
    int rc = 0;
    ASURF_DISPATCH_B(grid->basis_dim, run_forward<B_>(L, rays, rgb_out, 0, false, false, counters, pipe));
    return rc;
}

int asurf_volume_render_surf_trav_backward(const asurf_grid* grid, const asurf_rays* rays,
                                           const asurf_render_options* opt, const float* grad_out,
                                           const float* color_cache, const asurf_grads* grads,
                                           void* workspace, size_t workspace_bytes,
                                           uint64_t* counters, void* stream) {
    (void)counters;
    Launch L;
    if (int e = prepare(grid, rays, opt, 0, workspace, workspace_bytes, stream, L)) return e;
    if (!grads) return fail(ASURF_E_INVALID, "grads is NULL");
    if (rays->n_rays == 0) return 0;
    if (!grad_out || !color_cache) return fail(ASURF_E_INVALID, "grad_out / color_cache is NULL");
    KLoss loss;
    memset(&loss, 0, sizeof(loss));  // all fused-loss lambdas are 0 here (surf_trav.cu:3752-3773)
    const KGrads G = make_grads(grads);
    int rc = 0;
    ASURF_DISPATCH_B(grid->basis_dim, run_backward<B_>(L, rays, loss, G, grad_out, color_cache, 0, false, 0));
    return rc;
}

int asurf_volume_render_surf_trav_fused(const asurf_grid* grid, const asurf_rays* rays,
                                        const asurf_render_options* opt,
                                        const asurf_fused_params* fused, float* rgb_out,
                                        const asurf_grads* grads, void* workspace,
                                        size_t workspace_bytes, uint64_t* counters, void* stream) {
    if (!fused) return fail(ASURF_E_INVALID, "fused params are NULL");
    if (fused->fused_surf_norm_reg_scale > 0.f)
        return fail(ASURF_E_UNSUPPORTED,
                    "fused_surf_norm_reg_scale > 0 is no longer supported by the reference either "
                    "(assert(false) at surf_trav.cu:2875)");
    const int M = fused->l_dist_max_sample;
    if (M < 0) return fail(ASURF_E_INVALID, "l_dist_max_sample < 0");
    Launch L;
    if (int e = check_grid(grid)) return e;
    if (int e = check_rays(rays)) return e;
    if (!opt) return fail(ASURF_E_INVALID, "render options are NULL");
    if (int e = prepare(grid, rays, opt, M, workspace, workspace_bytes, stream, L, false, false,
                        use_pipeline(grid, rays, opt, counters)))
        return e;
    if (!grads) return fail(ASURF_E_INVALID, "grads is NULL");
    if (rays->n_rays == 0) return 0;
    if (!fused->rgb_gt || !rgb_out) return fail(ASURF_E_INVALID, "rgb_gt / rgb_out is NULL");
    const long long Qn = fused->global_n_rays > 0 ? fused->global_n_rays : rays->n_rays;
    KLoss loss;  // launcher-side scaling, surf_trav.cu:3896-3915
    loss.sparsity_loss = fused->sparsity_loss;
    loss.lambda_l2 = fused->lambda_l2;
    loss.lambda_l1 = fused->lambda_l1;
    loss.lambda_l_dist = fused->lambda_l_dist / Qn;
    loss.lambda_l_entropy = fused->lambda_l_entropy / Qn;
    loss.no_norm_weight_l_entropy = fused->no_norm_weight_l_entropy;
    loss.lambda_l_dist_a = fused->lambda_l_dist_a / Qn;
    loss.lambda_l_entropy_a = fused->lambda_l_entropy_a / Qn;
    loss.lambda_l_samp_dist = fused->lambda_l_samp_dist / Qn;
    loss.lambda_l_di = fused->lambda_l_di;
    loss.l_di_alpha_thresh = fused->l_di_alpha_thresh;
    loss.surf_sparse_alpha_thresh = fused->surf_sparse_alpha_thresh;
    loss.lambda_inplace_surf_sparse = fused->lambda_inplace_surf_sparse;
    loss.lambda_inwards_norm_loss = fused->lambda_inwards_norm_loss;
    loss.lambda_conv_mode_samp = fused->lambda_conv_mode_samp;
    loss.norm_l2 = 2.f / (3 * (int)Qn);  // surf_trav.cu:3308-3309
    loss.norm_l1 = 1.f / (3 * (int)Qn);
    const KGrads G = make_grads(grads);
    int rc = 0;
    // ---- rays in a coherent order (an extension; results are those of the given order, colours land where the
    // caller's rays are): order_rays 2 = always, 1 = never, 0 = when the last batch through this workspace arrived
    // shuffled.  Everything below works on the ordered copies; the permutation never leaves the workspace. ----
    asurf_rays rays_ord = *rays;
    const float* rgb_gt_use = fused->rgb_gt;
    float* rgb_use = rgb_out;
    const unsigned* perm = nullptr;
    bool probe_after = false;
    const asurf_rays* rays_given = rays;
    const bool tiles = rays->n_rays >= kOrderMinRays && fused->order_rays != 1 && fused->image_width >= 16 &&
                       rays->n_rays % fused->image_width == 0 && rays->n_rays / fused->image_width >= 16 && config().tiles;
    // rays DECLARED to come in image order (asurf_fused_params::image_width / asurf_rays::image_width) are coherent by
    // that declaration: the call neither asks what the last batch looked like nor probes this one (a kernel, a copy
    // to the host and an event less a call; the call's time is the same within the spread: 0.785 ms either way)
    const int declared_w = fused->image_width > 0 ? fused->image_width : rays->image_width;
    const bool declared_image = declared_w > 0 && rays->n_rays % declared_w == 0;
    if (tiles) {   // an image batch: 8 x 8 pixel tiles (asurf_order.hip, k_tile_scatter)
        unsigned* pm = reinterpret_cast<unsigned*>(L.ws + L.ML.off_ord_perm);
        float* o_s = reinterpret_cast<float*>(L.ws + L.ML.off_ord_o);
        float* d_s = reinterpret_cast<float*>(L.ws + L.ML.off_ord_d);
        float* gt_s = reinterpret_cast<float*>(L.ws + L.ML.off_ord_gt);
        asurf_order::tile_order(rays->origins, rays->dirs, fused->rgb_gt, rays->n_rays, fused->image_width, pm, o_s, d_s, gt_s,
                                L.stream);
        prof_after("ray_order", 1, L.stream);
        rays_ord.origins = o_s;
        rays_ord.dirs = d_s;
        rgb_gt_use = gt_s;
        rgb_use = reinterpret_cast<float*>(L.ws + L.ML.off_ord_rgb);
        perm = pm;
    } else if (rays->n_rays >= kOrderMinRays && fused->order_rays != 1 && (!declared_image || fused->order_rays == 2)) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(L.stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
        int* d_probe = reinterpret_cast<int*>(L.ws + L.ML.off_ord_probe);
        bool order = fused->order_rays == 2;
        int one_origin = 0;
        if (fused->order_rays == 0 && !capturing) {
            // (a call that is being captured into a graph renders as given: what its replays do must not depend on
            // what happened to be known when it was captured -- ADVICE r3; order_rays = 2 orders them)
            order = order_probe(workspace, rays->origins, rays->dirs, rays->n_rays, d_probe, L.stream, capturing,
                                &one_origin, false) == 0;
            probe_after = true;
        }
        if (order) {
            unsigned* pm = reinterpret_cast<unsigned*>(L.ws + L.ML.off_ord_perm);
            float* o_s = reinterpret_cast<float*>(L.ws + L.ML.off_ord_o);
            float* d_s = reinterpret_cast<float*>(L.ws + L.ML.off_ord_d);
            float* gt_s = reinterpret_cast<float*>(L.ws + L.ML.off_ord_gt);
            if (config().order_radix) {   // rocPRIM's radix sort + a gather (ASURF_ORDER=radix: the form the strip sort is tested against)
                if (asurf_order::order32(rays->origins, rays->dirs, rays->n_rays, pm, L.ws + L.ML.off_ord_tmp,
                                         L.ML.ord_tmp_bytes, L.stream, one_origin ? 24 : 30) != 0)
                    return fail(ASURF_E_LAUNCH, "ray ordering failed");
                asurf_order::gather(rays->origins, rays->dirs, fused->rgb_gt, pm, rays->n_rays, o_s, d_s, gt_s, L.stream);
            } else if (asurf_order::order_strips(rays->origins, rays->dirs, fused->rgb_gt, rays->n_rays, pm, o_s, d_s, gt_s,
                                                 L.ws + L.ML.off_ord_tmp, L.ML.ord_tmp_bytes, L.stream) != 0) {
                return fail(ASURF_E_LAUNCH, "ray ordering failed");
            }
            prof_after("ray_order", 1, L.stream);
            rays_ord.origins = o_s;
            rays_ord.dirs = d_s;
            rgb_gt_use = gt_s;
            rgb_use = reinterpret_cast<float*>(L.ws + L.ML.off_ord_rgb);
            perm = pm;
        }
    }
    rays = &rays_ord;
    // Replay instead of re-walking when the backward provably visits the forward's samples:
    // not with 0 < stop_thresh < 2e-8 (the two passes clamp log(1-alpha) differently,
    // surf_trav.cu:397 vs :2099) and not with more level sets than the record encodes.
    const Config cfg = config();
    const bool replay = !cfg.no_replay && grid->level_set_num <= 32 &&
                        (opt->stop_thresh <= 0.f || opt->stop_thresh >= 2e-8f);
    const bool pipe = use_pipeline(grid, rays, opt, counters);
    const int replay_mode = (cfg.no_split_bwd || rays->n_rays >= (1ll << kRecRayBits)) ? 1 : 3;
    // The rays the pipeline hands to the one-kernel path (tables, pool or records too small: none at all on most calls)
    // are rendered and differentiated by two kernels of their own -- one-kernel forward, walking backward -- that depend
    // on k_comp and on nothing after it.  On the call's stream each of them is a dependent launch in the chain (8 us
    // apiece whether or not a single ray was handed back).  With ASURF_SIDE_STREAM=1 they run on a stream of the
    // library's own beside k_bwd_seq / k_bwd_sh / k_bwd_geo and the call's stream waits for them at the end -- built,
    // measured and left OFF: the two event hand-overs cost more than the two launches (5 184 rays: 0.173 against 0.162 ms
    // per step, 0.143 against 0.131 with links_epoch; 640 k rays: 0.952 against 0.943).  tests/test_hip_parity.py keeps
    // the form alive.
    SideStream* side = (pipe && replay && replay_mode == 3 && !g_prof_on && cfg.side_stream) ? side_stream() : nullptr;
    Launch Lside = L;
    if (side != nullptr) Lside.stream = side->stream;
    const Launch* fb = side != nullptr ? &Lside : nullptr;
    ASURF_DISPATCH_B(grid->basis_dim, run_forward<B_>(L, rays, rgb_use, M, true, replay, counters, pipe, fb, side));
    if (rc) return rc;
    if (replay) {
        ASURF_DISPATCH_B(grid->basis_dim,
                         run_backward<B_>(L, rays, loss, G, rgb_gt_use, rgb_use, M, true, replay_mode, pipe,
                                          perm == nullptr ? (fused->image_width ? fused->image_width : rays_given->image_width) : 0));
        if (rc) return rc;
        ASURF_DISPATCH_B(grid->basis_dim,
                         run_backward<B_>(side != nullptr ? Lside : L, rays, loss, G, rgb_gt_use, rgb_use, M, true, 2, false, 0,
                                          pipe));
        if (rc) return rc;
        if (side != nullptr) {   // join
            if (hipEventRecord(side->ev_join, side->stream) != hipSuccess ||
                hipStreamWaitEvent(L.stream, side->ev_join, 0) != hipSuccess)
                return fail(ASURF_E_LAUNCH, "side stream join failed");
        }
    } else {
        ASURF_DISPATCH_B(grid->basis_dim, run_backward<B_>(L, rays, loss, G, rgb_gt_use, rgb_use, M, true, 0));
    }
    if (rc) return rc;
    if (perm != nullptr) {
        asurf_order::scatter_rgb(rgb_use, perm, rays->n_rays, rgb_out, L.stream);
        prof_after("ray_order_scatter", 2, L.stream);
    }
    if (probe_after) {   // how coherent was THIS batch as it came?  (read by the next call through this workspace)
        int one_origin = 0;
        (void)order_probe(workspace, rays_given->origins, rays_given->dirs, rays_given->n_rays,
                          reinterpret_cast<int*>(L.ws + L.ML.off_ord_probe), L.stream, false, &one_origin, true);
    }
    return launch_error("fused call epilogue");
}

static int optim_launch(bool rms_mode, float* data, float* rms, float* grad, int64_t n_rows,
                        int32_t n_cols, const uint8_t* mask, const int64_t* indices,
                        int64_t n_indices, float beta, float lr, float epsilon, float minval,
                        float lr_last, void* stream, const int* n_idx_dev = nullptr) {
    if (!data || !grad || (rms_mode && !rms)) return fail(ASURF_E_INVALID, "optim step: NULL tensor");
    if (n_rows < 0 || n_cols <= 0 || n_indices < 0) return fail(ASURF_E_INVALID, "optim step: bad shape");
    if (lr_last < 0.f) lr_last = lr;  // optim_kernel.cu:169
    const int mode = n_idx_dev ? 2 : (mask ? 1 : (indices ? 2 : 0));
    if (n_idx_dev) n_indices = n_rows;  // launch shape only (the kernels stride to what they find on the device)
    if (mode == 2 && n_indices == 0) return 0;  // empty index list: nothing to do
    const long long total = (mode == 2 ? n_indices : n_rows) * (long long)n_cols;
    if (total == 0) return 0;
    auto* idx = reinterpret_cast<const long long*>(indices);
    if (n_cols <= 64) {  // a row per power-of-two lane group: no division, masked-out rows cost one byte
        int cp_shift = 0;
        while ((1 << cp_shift) < n_cols) ++cp_shift;
        const long long lanes = (mode == 2 ? n_indices : n_rows) << cp_shift;
        long long blocks = (lanes + 255) / 256;
        if (blocks > 256 * 32) blocks = 256 * 32;
        if (rms_mode)
            hipLaunchKernelGGL((k_optim_rows<true>), dim3((unsigned)blocks), dim3(256), 0,
                               static_cast<hipStream_t>(stream), data, rms, grad, (long long)n_rows, n_cols, cp_shift,
                               mode, mask, idx, (long long)n_indices, beta, lr, epsilon, minval, lr_last, n_idx_dev);
        else
            hipLaunchKernelGGL((k_optim_rows<false>), dim3((unsigned)blocks), dim3(256), 0,
                               static_cast<hipStream_t>(stream), data, rms, grad, (long long)n_rows, n_cols, cp_shift,
                               mode, mask, idx, (long long)n_indices, beta, lr, epsilon, minval, lr_last, n_idx_dev);
        return launch_error("k_optim_rows");
    }
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride beyond 16 blocks per CU
    if (rms_mode)
        hipLaunchKernelGGL((k_optim_step<true>), dim3((unsigned)blocks), dim3(256), 0,
                           static_cast<hipStream_t>(stream), data, rms, grad, (long long)n_rows, n_cols,
                           mode, mask, idx, (long long)n_indices, beta, lr, epsilon, minval, lr_last, n_idx_dev);
    else
        hipLaunchKernelGGL((k_optim_step<false>), dim3((unsigned)blocks), dim3(256), 0,
                           static_cast<hipStream_t>(stream), data, rms, grad, (long long)n_rows, n_cols,
                           mode, mask, idx, (long long)n_indices, beta, lr, epsilon, minval, lr_last, n_idx_dev);
    return launch_error("k_optim_step");
}

int asurf_rmsprop_step(float* data, float* rms, float* grad, int64_t n_rows, int32_t n_cols,
                       const uint8_t* mask, const int64_t* indices, int64_t n_indices, float beta,
                       float lr, float epsilon, float minval, float lr_last, void* stream) {
    return optim_launch(true, data, rms, grad, n_rows, n_cols, mask, indices, n_indices, beta, lr,
                        epsilon, minval, lr_last, stream);
}

int asurf_sgd_step(float* data, float* grad, int64_t n_rows, int32_t n_cols, const uint8_t* mask,
                   const int64_t* indices, int64_t n_indices, float lr, float lr_last, void* stream) {
    return optim_launch(false, data, nullptr, grad, n_rows, n_cols, mask, indices, n_indices, 0.f, lr,
                        0.f, 0.f, lr_last, stream);
}

size_t asurf_optim_scratch_bytes(int64_t n_rows) {
    const size_t n = (size_t)(n_rows > 0 ? n_rows : 0);
    const size_t n_seg = (n + kSegRows - 1) / kSegRows;   // (the segmented list: whole segments of rows + a count each)
    return std::max(n * 8, n_seg * kSegRows * 4 + n_seg * 4) + 256;
}

// mask -> row list on the device, then the index-list step over it (its length stays on the device)
// The masked steps build the row list on the device (k_compact_rows) so that a sparse mask costs its rows only; a mask
// that names a quarter of the rows or more is walked itself, and then the list was built for nothing (a memset and a
// kernel, 17 us a step, three steps an iteration -- and with the shipped configuration's regularisers touching every
// non-empty cell, the masks of a training iteration are full).  What the last list of a mask held is therefore
// remembered (read back asynchronously): after a dense one the next 15 steps on the same mask go straight to the mask
// form, the 16th builds the list again to see whether that is still so.  Only speed depends on it: the mask form is
// correct for every mask.
struct MaskDensity {
    const void* mask = nullptr;
    long long n_rows = 0;
    int* h_count = nullptr;   // pinned
    hipEvent_t ev = nullptr;
    bool pending = false;
    int dense_streak = 0;     // > 0: steps left that skip the list
    unsigned long long used = 0;
};
constexpr int kMaskDensityEntries = 8;
std::mutex g_mask_density_mu;
MaskDensity g_mask_density[kMaskDensityEntries];
unsigned long long g_mask_density_clock = 0;

// true: skip the list this time
bool mask_density_skip(const void* mask, long long n_rows, hipStream_t s, MaskDensity** slot_out) {
    *slot_out = nullptr;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return false;
    std::lock_guard<std::mutex> lock(g_mask_density_mu);
    MaskDensity* slot = nullptr;
    for (MaskDensity& e : g_mask_density)
        if (e.mask == mask && e.n_rows == n_rows) slot = &e;
    if (!slot) {
        slot = &g_mask_density[0];
        for (MaskDensity& e : g_mask_density)
            if (e.mask == nullptr || e.used < slot->used) {
                slot = &e;
                if (e.mask == nullptr) break;
            }
        if (slot->pending && slot->ev) (void)hipEventSynchronize(slot->ev);
        slot->mask = mask;
        slot->n_rows = n_rows;
        slot->pending = false;
        slot->dense_streak = 0;
    }
    slot->used = ++g_mask_density_clock;
    if (slot->pending) {
        if (hipEventQuery(slot->ev) == hipSuccess) {
            slot->pending = false;
            if (4ll * (*slot->h_count) >= n_rows) slot->dense_streak = 15;
        } else {
            (void)hipGetLastError();
        }
    }
    if (slot->dense_streak > 0) {
        --slot->dense_streak;
        return true;
    }
    if (!slot->pending) *slot_out = slot;   // the caller reads the new list's length back into it
    return false;
}

void mask_density_record(MaskDensity* slot, const int* d_count, hipStream_t s) {
    if (!slot) return;
    std::lock_guard<std::mutex> lock(g_mask_density_mu);
    if (!slot->h_count && hipHostMalloc(reinterpret_cast<void**>(&slot->h_count), sizeof(int)) != hipSuccess) {
        slot->h_count = nullptr;
        return;
    }
    if (!slot->ev && hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming) != hipSuccess) {
        slot->ev = nullptr;
        return;
    }
    if (hipMemcpyAsync(slot->h_count, d_count, sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess &&
        hipEventRecord(slot->ev, s) == hipSuccess)
        slot->pending = true;
}

static int optim_masked(bool rms_mode, float* data, float* rms, float* grad, int64_t n_rows, int32_t n_cols,
                        const uint8_t* mask, void* scratch, size_t scratch_bytes, float beta, float lr, float epsilon,
                        float minval, float lr_last, void* stream) {
    if (!mask) return fail(ASURF_E_INVALID, "optim step: mask is NULL");
    if (n_rows <= 0) return 0;
    if (!scratch || scratch_bytes < asurf_optim_scratch_bytes(n_rows))
        return fail(ASURF_E_WORKSPACE, "optim step: scratch of %zu bytes needed", asurf_optim_scratch_bytes(n_rows));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_cols <= 64 && n_rows < (1ll << 31)) {
        // the segmented list (asurf_optim.h): n_rows ints of rows, then a count per segment
        if (!data || !grad || (rms_mode && !rms)) return fail(ASURF_E_INVALID, "optim step: NULL tensor");
        if (lr_last < 0.f) lr_last = lr;  // optim_kernel.cu:169
        const long long n_seg = (n_rows + kSegRows - 1) / kSegRows;
        if (!config().optim_two_step) {   // one kernel: the segment's rows listed in LDS
            int cps = 0;
            while ((1 << cps) < n_cols) ++cps;
            const unsigned nblk = (unsigned)std::min<long long>(n_seg, 8192);
            if (rms_mode)
                hipLaunchKernelGGL((k_optim_masked<true>), dim3(nblk), dim3(256), 0, s, data, rms, grad, (long long)n_rows,
                                   n_cols, cps, mask, (int)n_seg, beta, lr, epsilon, minval, lr_last);
            else
                hipLaunchKernelGGL((k_optim_masked<false>), dim3(nblk), dim3(256), 0, s, data, rms, grad, (long long)n_rows,
                                   n_cols, cps, mask, (int)n_seg, beta, lr, epsilon, minval, lr_last);
            return launch_error("k_optim_masked");
        }
        int* seg_rows = reinterpret_cast<int*>(static_cast<char*>(scratch) + 256);
        int* seg_count = seg_rows + n_seg * kSegRows;
        hipLaunchKernelGGL(k_compact_segments, dim3((unsigned)((n_rows + kSegBlockRows - 1) / kSegBlockRows)),
                           dim3(256), 0, s, mask, (long long)n_rows, seg_rows, seg_count);
        int cp_shift = 0;
        while ((1 << cp_shift) < n_cols) ++cp_shift;
        const unsigned blocks = (unsigned)std::min<long long>(n_seg, 2048);
        if (rms_mode)
            hipLaunchKernelGGL((k_optim_segments<true>), dim3(blocks), dim3(256), 0, s, data, rms, grad, n_cols, cp_shift,
                               seg_rows, seg_count, (int)n_seg, beta, lr, epsilon, minval, lr_last);
        else
            hipLaunchKernelGGL((k_optim_segments<false>), dim3(blocks), dim3(256), 0, s, data, rms, grad, n_cols, cp_shift,
                               seg_rows, seg_count, (int)n_seg, beta, lr, epsilon, minval, lr_last);
        return launch_error("k_optim_segments");
    }
    int* count = static_cast<int*>(scratch);
    long long* rows = reinterpret_cast<long long*>(static_cast<char*>(scratch) + 256);
    MaskDensity* slot = nullptr;
    if (mask_density_skip(mask, n_rows, s, &slot))   // the mask itself (dense the last time it was listed)
        return optim_launch(rms_mode, data, rms, grad, n_rows, n_cols, mask, nullptr, 0, beta, lr, epsilon, minval, lr_last,
                            stream);
    if (hipMemsetAsync(count, 0, sizeof(int), s) != hipSuccess) return fail(ASURF_E_LAUNCH, "hipMemsetAsync failed");
    hipLaunchKernelGGL(k_compact_rows, dim3((unsigned)((n_rows + kCompactRowsPerBlock - 1) / kCompactRowsPerBlock)),
                       dim3(256), 0, s, mask, (long long)n_rows, rows, count);
    if (int e = launch_error("k_compact_rows")) return e;
    // (the step kernel reads the count: a list that names a quarter of the rows or more is not walked, the mask is)
    const int rc = optim_launch(rms_mode, data, rms, grad, n_rows, n_cols, mask, reinterpret_cast<const int64_t*>(rows), 0,
                                beta, lr, epsilon, minval, lr_last, stream, count);
    mask_density_record(slot, count, s);
    return rc;
}

int asurf_rmsprop_step_masked(float* data, float* rms, float* grad, int64_t n_rows, int32_t n_cols,
                              const uint8_t* mask, void* scratch, size_t scratch_bytes, float beta, float lr,
                              float epsilon, float minval, float lr_last, void* stream) {
    return optim_masked(true, data, rms, grad, n_rows, n_cols, mask, scratch, scratch_bytes, beta, lr, epsilon,
                        minval, lr_last, stream);
}

int asurf_sgd_step_masked(float* data, float* grad, int64_t n_rows, int32_t n_cols, const uint8_t* mask,
                          void* scratch, size_t scratch_bytes, float lr, float lr_last, void* stream) {
    return optim_masked(false, data, nullptr, grad, n_rows, n_cols, mask, scratch, scratch_bytes, 0.f, lr, 0.f, 0.f,
                        lr_last, stream);
}

// ---- multi-GPU gradient exchange, device side (asurf_exchange.h) ----
size_t asurf_rows_list_tmp_bytes(int64_t n_rows) {
    const int64_t nb = n_rows > 0 ? (n_rows + kRowsPerBlock - 1) / kRowsPerBlock : 0;
    return (size_t)((nb * 4 + 255) / 256 * 256);
}

int asurf_rows_list(const uint8_t* mask, int64_t n_rows, int32_t* rows_out, int32_t* count_out, void* tmp, size_t tmp_bytes,
                    void* stream) {
    if (n_rows < 0 || n_rows >= (1ll << 31)) return fail(ASURF_E_INVALID, "rows_list: n_rows out of range");
    if (!count_out) return fail(ASURF_E_INVALID, "rows_list: count_out is NULL");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (n_rows == 0) {
        if (hipMemsetAsync(count_out, 0, sizeof(int32_t), s) != hipSuccess) return fail(ASURF_E_LAUNCH, "hipMemsetAsync failed");
        return 0;
    }
    if (!mask || !rows_out || !tmp || tmp_bytes < asurf_rows_list_tmp_bytes(n_rows))
        return fail(ASURF_E_INVALID, "rows_list: NULL tensor or scratch too small");
    const int nb = (int)((n_rows + kRowsPerBlock - 1) / kRowsPerBlock);
    int* counts = static_cast<int*>(tmp);
    hipLaunchKernelGGL(k_rows_count, dim3((unsigned)nb), dim3(256), 0, s, mask, (long long)n_rows, counts);
    hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, s, counts, nb, count_out);
    hipLaunchKernelGGL(k_rows_fill, dim3((unsigned)nb), dim3(256), 0, s, mask, (long long)n_rows, counts, rows_out);
    return launch_error("rows_list");
}

int asurf_rows_pack(const int32_t* rows, int64_t n_list, int32_t sh_dim, const float* grad_sh, const float* grad_density,
                    const float* grad_surface, float* packed, void* stream) {
    if (n_list < 0 || sh_dim <= 0) return fail(ASURF_E_INVALID, "rows_pack: bad shape");
    if (n_list == 0) return 0;
    if (!rows || !grad_sh || !grad_density || !packed) return fail(ASURF_E_INVALID, "rows_pack: NULL tensor");
    const long long total = (long long)n_list * (sh_dim + 2);
    const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, 256 * 32);
    hipLaunchKernelGGL(k_rows_pack, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), rows, (long long)n_list,
                       sh_dim, grad_sh, grad_density, grad_surface, packed);
    return launch_error("rows_pack");
}

static int rows_unpack_impl(bool add, const int32_t* rows, int64_t n_list, int32_t sh_dim, const float* packed,
                            float* grad_sh, float* grad_density, float* grad_surface, uint8_t* mask, void* stream) {
    if (n_list < 0 || sh_dim <= 0) return fail(ASURF_E_INVALID, "rows_unpack: bad shape");
    if (n_list == 0) return 0;
    if (!rows || !grad_sh || !grad_density || !packed) return fail(ASURF_E_INVALID, "rows_unpack: NULL tensor");
    const long long total = (long long)n_list * (sh_dim + 2);
    const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, 256 * 32);
    if (add)
        hipLaunchKernelGGL(k_rows_unpack<true>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), rows,
                           (long long)n_list, sh_dim, packed, grad_sh, grad_density, grad_surface, mask);
    else
        hipLaunchKernelGGL(k_rows_unpack<false>, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), rows,
                           (long long)n_list, sh_dim, packed, grad_sh, grad_density, grad_surface, mask);
    return launch_error("rows_unpack");
}

int asurf_rows_unpack(const int32_t* rows, int64_t n_list, int32_t sh_dim, const float* packed, float* grad_sh,
                      float* grad_density, float* grad_surface, uint8_t* mask, void* stream) {
    return rows_unpack_impl(false, rows, n_list, sh_dim, packed, grad_sh, grad_density, grad_surface, mask, stream);
}

int asurf_rows_unpack_add(const int32_t* rows, int64_t n_list, int32_t sh_dim, const float* packed, float* grad_sh,
                          float* grad_density, float* grad_surface, uint8_t* mask, void* stream) {
    return rows_unpack_impl(true, rows, n_list, sh_dim, packed, grad_sh, grad_density, grad_surface, mask, stream);
}

int asurf_profile_enable(int on) {
    g_prof_on = on != 0;
    return 0;
}

int asurf_profile_read(float* ms, int n) {
    if (!ms || n < 3) return fail(ASURF_E_INVALID, "asurf_profile_read needs room for 3 floats");
    ms[0] = ms[1] = ms[2] = 0.f;
    if (!g_prof_on || g_spans < 1) return fail(ASURF_E_INVALID, "profiling is off or no call was made");
    if (hipEventSynchronize(g_ev[g_spans]) != hipSuccess) return fail(ASURF_E_LAUNCH, "hipEventSynchronize");
    for (int i = 0; i < g_spans; ++i) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_ev[i], g_ev[i + 1]) != hipSuccess) return fail(ASURF_E_LAUNCH, "hipEventElapsedTime");
        ms[g_span_group[i]] += t;
    }
    return 0;
}

int asurf_config_set(const char* name, const char* value) {
    if (!name) return fail(ASURF_E_INVALID, "asurf_config_set: name is NULL");
    (void)config();  // the environment is read first, once; this call then overrides it
    std::lock_guard<std::mutex> lock(g_cfg_mu);
    if (!config_assign(g_cfg, name, value)) return fail(ASURF_E_INVALID, "asurf_config_set: unknown switch '%s'", name);
    return 0;
}

int asurf_debug_read_timeline(void* dst, int max_recs) {
    if (!dst || max_recs < 1) return fail(ASURF_E_INVALID, "asurf_debug_read_timeline: NULL buffer"), -1;
#ifdef ASURF_TIMELINE
    const int n = std::min(max_recs, kTimelineCap);
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_timeline), (size_t)n * sizeof(TimelineRec)) != hipSuccess)
        return fail(ASURF_E_LAUNCH, "asurf_debug_read_timeline: copy failed"), -1;
    return n;
#else
    return fail(ASURF_E_INVALID, "asurf_debug_read_timeline: this is not the instrumented build (make timeline)"), -1;
#endif
}

int asurf_profile_read_spans(char* names, size_t names_cap, float* ms, int max_spans) {
    if (!names || !ms || max_spans < 1 || names_cap < 2)
        return fail(ASURF_E_INVALID, "asurf_profile_read_spans: NULL buffer"), -1;
    if (!g_prof_on || g_spans < 1) return fail(ASURF_E_INVALID, "profiling is off or no call was made"), -1;
    if (hipEventSynchronize(g_ev[g_spans]) != hipSuccess) return fail(ASURF_E_LAUNCH, "hipEventSynchronize"), -1;
    size_t used = 0;
    int n = 0;
    names[0] = 0;
    for (int i = 0; i < g_spans && n < max_spans; ++i, ++n) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, g_ev[i], g_ev[i + 1]) != hipSuccess)
            return fail(ASURF_E_LAUNCH, "hipEventElapsedTime"), -1;
        ms[n] = t;
        const int w = snprintf(names + used, names_cap - used, "%s%s", n ? "," : "", g_span_name[i]);
        if (w < 0 || (size_t)w >= names_cap - used) return fail(ASURF_E_INVALID, "names buffer too small"), -1;
        used += (size_t)w;
    }
    return n;
}

int asurf_test_cubic_root_grad(const double* fs, const int32_t* st_ids, float* grad_fs, float* st,
                               int64_t n, int32_t use_vieta, void* stream) {
    if (n < 0 || (n > 0 && (!fs || !st_ids || !grad_fs || !st)))
        return fail(ASURF_E_INVALID, "test_cubic_root_grad: NULL tensor");
    if (n == 0) return 0;
    if (!use_vieta) {
        hipLaunchKernelGGL(k_test_cubic_root_grad_legacy, dim3((unsigned)((n + 127) / 128)), dim3(128), 0,
                           static_cast<hipStream_t>(stream), fs, st_ids, grad_fs, st, (long long)n);
        return launch_error("k_test_cubic_root_grad_legacy");
    }
    hipLaunchKernelGGL(k_test_cubic_root_grad, dim3((unsigned)((n + 127) / 128)), dim3(128), 0,
                       static_cast<hipStream_t>(stream), fs, st_ids, grad_fs, st, (long long)n);
    return launch_error("k_test_cubic_root_grad");
}

int asurf_volume_render_surface(const asurf_grid* grid, const asurf_rays* rays,
                                const asurf_ray_vox* ray_vox, const asurf_render_options* opt,
                                float* rgb_out, void* stream) {
    if (int e = check_legacy(grid, rays, ray_vox, opt)) return e;
    if (rays->n_rays == 0) return 0;
    if (!rgb_out) return fail(ASURF_E_INVALID, "rgb_out is NULL");
    const KGrid g = make_kgrid_plain(grid);
    const KOpt o = make_kopt(opt);
    const dim3 grid_dim((unsigned)((rays->n_rays + kBlock - 1) / kBlock));
    hipStream_t s = static_cast<hipStream_t>(stream);
    int rc = 0;
ASURF_DISPATCH_B(grid->basis_dim, (leg_fwd<B_>(grid_dim, s, g, o, rays, ray_vox, rgb_out)));
    (void)rc;
    return launch_error("k_surface_fwd");
}

namespace {
int launch_surface_backward(const asurf_grid* grid, const asurf_rays* rays, const asurf_ray_vox* ray_vox,
                            const asurf_render_options* opt, const float* grad_in,
                            const float* color_cache, int grad_in_is_rgb, float sparsity_loss,
                            const asurf_grads* grads, hipStream_t s) {
    const KGrid g = make_kgrid_plain(grid);
    const KOpt o = make_kopt(opt);
    const KGrads G = make_grads(grads);
    const dim3 grid_dim((unsigned)((rays->n_rays + kBlock - 1) / kBlock));
    int rc = 0;
ASURF_DISPATCH_B(grid->basis_dim, (leg_bwd<B_>(grid_dim, s, g, o, G, rays, ray_vox, grad_in, color_cache,
                                                grad_in_is_rgb, sparsity_loss)));
    (void)rc;
    return launch_error("k_surface_bwd");
}
}  // namespace

int asurf_volume_render_surface_backward(const asurf_grid* grid, const asurf_rays* rays,
                                         const asurf_ray_vox* ray_vox,
                                         const asurf_render_options* opt, const float* grad_out,
                                         const float* color_cache, const asurf_grads* grads,
                                         void* stream) {
    if (int e = check_legacy(grid, rays, ray_vox, opt)) return e;
    if (!grads) return fail(ASURF_E_INVALID, "grads is NULL");
    if (rays->n_rays == 0) return 0;
    if (!grad_out || !color_cache) return fail(ASURF_E_INVALID, "grad_out / color_cache is NULL");
    return launch_surface_backward(grid, rays, ray_vox, opt, grad_out, color_cache, 0, 0.f, grads,
                                   static_cast<hipStream_t>(stream));
}

// render_lerp_kernel_surface.cu:1325-1439.  beta_loss only reaches the background model
// (:636-640), which this build does not render (check_grid rejects MSI layers).
int asurf_volume_render_surface_fused(const asurf_grid* grid, const asurf_rays* rays,
                                      const asurf_ray_vox* ray_vox,
                                      const asurf_render_options* opt, const float* rgb_gt,
                                      float beta_loss, float sparsity_loss, float* rgb_out,
                                      const asurf_grads* grads, void* stream) {
    (void)beta_loss;
    if (int e = check_legacy(grid, rays, ray_vox, opt)) return e;
    if (!grads) return fail(ASURF_E_INVALID, "grads is NULL");
    if (rays->n_rays == 0) return 0;
    if (!rgb_gt || !rgb_out) return fail(ASURF_E_INVALID, "rgb_gt / rgb_out is NULL");
    if (int e = asurf_volume_render_surface(grid, rays, ray_vox, opt, rgb_out, stream)) return e;
    return launch_surface_backward(grid, rays, ray_vox, opt, rgb_gt, rgb_out, 1, sparsity_loss, grads,
                                   static_cast<hipStream_t>(stream));
}

// Voxel lists of the legacy backend from the surf_trav DDA (two passes around a host-side
// exclusive scan of vox_num).
int asurf_ray_voxel_count(const asurf_grid* grid, const asurf_rays* rays,
                          const asurf_render_options* opt, int32_t skip_empty, int32_t* vox_num,
                          void* stream) {
    if (int e = check_grid(grid)) return e;
    if (int e = check_rays(rays)) return e;
    if (!opt) return fail(ASURF_E_INVALID, "opt is NULL");
    if (rays->n_rays == 0) return 0;
    if (!vox_num) return fail(ASURF_E_INVALID, "vox_num is NULL");
    const KGrid g = make_kgrid_plain(grid);
    const KOpt o = make_kopt(opt);
    hipLaunchKernelGGL((k_voxel_lists<false>), dim3((unsigned)((rays->n_rays + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, static_cast<hipStream_t>(stream), g, o, rays->origins, rays->dirs,
                       (long long)rays->n_rays, (int)skip_empty, vox_num, (const int*)nullptr,
                       (int*)nullptr, 0ll);
    return launch_error("k_voxel_lists<count>");
}

int asurf_ray_voxel_fill(const asurf_grid* grid, const asurf_rays* rays,
                         const asurf_render_options* opt, int32_t skip_empty,
                         const asurf_ray_vox* ray_vox, int32_t* voxel_ls_out, void* stream) {
    if (int e = check_grid(grid)) return e;
    if (int e = check_rays(rays)) return e;
    if (!opt) return fail(ASURF_E_INVALID, "opt is NULL");
    if (!ray_vox || ray_vox->n_vox < 0) return fail(ASURF_E_INVALID, "ray_vox is NULL / n_vox < 0");
    if (rays->n_rays == 0 || ray_vox->n_vox == 0) return 0;
    if (!ray_vox->vox_start_i || !ray_vox->vox_num || !voxel_ls_out)
        return fail(ASURF_E_INVALID, "vox_start_i / vox_num / voxel_ls_out is NULL");
    const KGrid g = make_kgrid_plain(grid);
    const KOpt o = make_kopt(opt);
    hipLaunchKernelGGL((k_voxel_lists<true>), dim3((unsigned)((rays->n_rays + kBlock - 1) / kBlock)),
                       dim3(kBlock), 0, static_cast<hipStream_t>(stream), g, o, rays->origins, rays->dirs,
                       (long long)rays->n_rays, (int)skip_empty, const_cast<int32_t*>(ray_vox->vox_num),
                       ray_vox->vox_start_i, voxel_ls_out, (long long)ray_vox->n_vox);
    return launch_error("k_voxel_lists<fill>");
}

// ---- regularisers (asurf_reg.h) ----
namespace {
int reg_grid(const int32_t* links, const int32_t size[3], RegGrid& G) {
    if (!links || !size) return fail(ASURF_E_INVALID, "links / size is NULL");
    for (int i = 0; i < 3; ++i)
        if (size[i] < 1) return fail(ASURF_E_INVALID, "links.shape[%d] = %d", i, size[i]);
    if ((int64_t)size[0] * size[1] * size[2] >= (1ll << 31)) return fail(ASURF_E_INVALID, "links has >= 2^31 elements");
    G.links = links;
    for (int i = 0; i < 3; ++i) G.size[i] = size[i];
    G.offx = size[1] * size[2];
    G.offy = size[2];
    return 0;
}
unsigned reg_blocks(long long Q) { return (unsigned)((Q + 255) / 256); }
}  // namespace

int asurf_tv_grad_sparse(const int32_t* links, const int32_t size[3], const float* data, int32_t n_cols,
                         const int32_t* rand_cells, int64_t n_cells, uint8_t* mask_out,
                         int32_t start_dim, int32_t end_dim, float scale, int32_t ignore_edge,
                         int32_t ignore_last_z, float* grad_data, void* stream) {
    RegGrid G;
    if (int e = reg_grid(links, size, G)) return e;
    if (n_cells < 0 || start_dim < 0 || end_dim > n_cols || end_dim <= start_dim)
        return fail(ASURF_E_INVALID, "tv_grad_sparse: bad n_cells / dim range");
    if (n_cells == 0) return 0;
    if (!data || !rand_cells || !grad_data) return fail(ASURF_E_INVALID, "tv_grad_sparse: NULL tensor");
    const long long Q = (long long)n_cells * (end_dim - start_dim);
    hipLaunchKernelGGL((k_tv_grad_sparse<false>), dim3(reg_blocks(Q)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), G, data, (const float*)nullptr, (int)n_cols, 0,
                       rand_cells, (int)start_dim, (int)end_dim, scale / (float)(int)n_cells, Q,
                       (int)ignore_edge, 0.f, (int)ignore_last_z, 0, mask_out, grad_data);
    return launch_error("k_tv_grad_sparse");
}

int asurf_tv(const int32_t* links, const int32_t size[3], const float* data, int32_t n_cols,
             int32_t start_dim, int32_t end_dim, int32_t ignore_edge, float* out, void* stream) {
    RegGrid G;
    if (int e = reg_grid(links, size, G)) return e;
    if (start_dim < 0 || end_dim > n_cols || end_dim <= start_dim) return fail(ASURF_E_INVALID, "tv: bad dim range");
    if (!out) return fail(ASURF_E_INVALID, "tv: NULL output");
    if (size[0] < 2 || size[1] < 2 || size[2] < 2) return 0;
    if (!data) return fail(ASURF_E_INVALID, "tv: NULL tensor");
    const long long nl = (long long)(size[0] - 1) * (size[1] - 1) * (size[2] - 1);
    const long long Q = nl * (end_dim - start_dim);
    const long long want = (Q + 255) / 256;
    const int blocks = (int)(want < 4096 ? want : 4096);  // grid-stride: a few thousand atomics at most
    hipLaunchKernelGGL(k_tv_value, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), G, data,
                       (int)n_cols, (int)start_dim, (int)end_dim, 1.f / (float)nl, Q, (int)ignore_edge, out);
    return launch_error("k_tv_value");
}

int asurf_surf_tv_grad_sparse(const int32_t* links, const int32_t size[3], const float* data,
                              int32_t n_cols, const float* density_data, int32_t density_cols,
                              const int32_t* rand_cells, int64_t n_cells, uint8_t* mask_out,
                              int32_t start_dim, int32_t end_dim, float scale, int32_t ignore_edge,
                              float edge_value, int32_t ignore_last_z, int32_t alpha_dependency,
                              float* grad_data, void* stream) {
    RegGrid G;
    if (int e = reg_grid(links, size, G)) return e;
    if (n_cells < 0 || start_dim < 0 || end_dim > n_cols || end_dim <= start_dim)
        return fail(ASURF_E_INVALID, "surf_tv_grad_sparse: bad n_cells / dim range");
    if (n_cells == 0) return 0;
    if (!data || !rand_cells || !grad_data) return fail(ASURF_E_INVALID, "surf_tv_grad_sparse: NULL tensor");
    if (alpha_dependency && (!density_data || end_dim > density_cols))
        return fail(ASURF_E_INVALID, "surf_tv_grad_sparse: density_data missing / too narrow");
    const long long Q = (long long)n_cells * (end_dim - start_dim);
    hipLaunchKernelGGL((k_tv_grad_sparse<true>), dim3(reg_blocks(Q)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), G, data, density_data, (int)n_cols,
                       (int)density_cols, rand_cells, (int)start_dim, (int)end_dim,
                       scale / (float)(int)n_cells, Q, (int)ignore_edge, edge_value, (int)ignore_last_z,
                       (int)alpha_dependency, mask_out, grad_data);
    return launch_error("k_surf_tv_grad_sparse");
}

int asurf_surface_normal_grad_sparse(const int32_t* links, const int32_t size[3],
                                     const float* surface_data, const int32_t* rand_cells,
                                     int64_t n_cells, uint8_t* mask_out, float lv_set, float scale,
                                     int32_t con_check, int32_t ignore_empty, int32_t use_l1,
                                     float* grad_data, void* stream) {
    RegGrid G;
    if (int e = reg_grid(links, size, G)) return e;
    if (n_cells < 0) return fail(ASURF_E_INVALID, "surface_normal_grad_sparse: n_cells < 0");
    if (n_cells == 0) return 0;
    if (!surface_data || !rand_cells || !grad_data)
        return fail(ASURF_E_INVALID, "surface_normal_grad_sparse: NULL tensor");
    hipLaunchKernelGGL(k_surface_normal_grad_sparse, dim3(reg_blocks(n_cells)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), G, surface_data, rand_cells, lv_set,
                       scale / (float)(int)n_cells, (long long)n_cells, (int)con_check, (int)ignore_empty,
                       (int)use_l1, mask_out, grad_data);
    return launch_error("k_surface_normal_grad_sparse");
}

int asurf_alpha_surf_sparsify_grad_sparse(const int32_t* links, const int32_t size[3],
                                          const float* alpha_data, const float* surf_data,
                                          const int32_t* rand_cells, int64_t n_cells, uint8_t* mask_out,
                                          float scale_alpha, float scale_surf, int32_t surf_sparse_decrease,
                                          float surf_sparse_thresh, float alpha_bound, float surf_bound,
                                          float* grad_alpha, float* grad_surf, void* stream) {
    RegGrid G;
    if (int e = reg_grid(links, size, G)) return e;
    if (n_cells < 0) return fail(ASURF_E_INVALID, "alpha_surf_sparsify_grad_sparse: n_cells < 0");
    if (n_cells == 0) return 0;
    if (!alpha_data || !surf_data || !rand_cells || !grad_alpha || !grad_surf)
        return fail(ASURF_E_INVALID, "alpha_surf_sparsify_grad_sparse: NULL tensor");
    hipLaunchKernelGGL(k_alpha_surf_sparsify, dim3(reg_blocks(n_cells)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), G, alpha_data, surf_data, rand_cells, scale_alpha,
                       scale_surf, (int)surf_sparse_decrease, surf_sparse_thresh, alpha_bound, surf_bound,
                       (long long)n_cells, mask_out, grad_alpha, grad_surf);
    return launch_error("k_alpha_surf_sparsify");
}

int asurf_surface_eikonal_grad_sparse(const int32_t* links, const int32_t size[3],
                                      const float* surface_data, const int32_t* rand_cells,
                                      int64_t n_cells, uint8_t* mask_out, float scale, float h,
                                      float* grad_data, float* loss_out, void* stream) {
    RegGrid G;
    if (int e = reg_grid(links, size, G)) return e;
    if (n_cells < 0 || !(h > 0.f)) return fail(ASURF_E_INVALID, "surface_eikonal_grad_sparse: n_cells < 0 or h <= 0");
    if (n_cells == 0) return 0;
    if (!surface_data || !rand_cells || !grad_data)
        return fail(ASURF_E_INVALID, "surface_eikonal_grad_sparse: NULL tensor");
    hipLaunchKernelGGL(k_surface_eikonal_grad_sparse, dim3(std::min(reg_blocks(n_cells), 1024u)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), G, surface_data, rand_cells,
                       scale / (float)(int)n_cells, 1.f / (h * h), (long long)n_cells, mask_out, grad_data,
                       loss_out);
    return launch_error("k_surface_eikonal_grad_sparse");
}

// ---- evaluation renders (asurf_eval.h) ----
int asurf_render_eval_surf_trav(const asurf_grid* grid, const asurf_rays* rays,
                                const asurf_render_options* opt, int32_t mode, float param,
                                int32_t max_sample, float* out0, float* out1, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (mode < 0 || mode > 5) return fail(ASURF_E_INVALID, "render_eval: mode %d", mode);
    if (!opt) return fail(ASURF_E_INVALID, "render options are NULL");
    // no density gate, no fake samples: the walk needs links + a level set in range only
    asurf_render_options eo = *opt;
    eo.sigma_thresh = -INFINITY;
    eo.surf_fake_sample = 0;
    eo.only_outward_intersect = 0;  // (these renders take every crossing, surf_trav.cu:564-1705)
    // through k_sweep + k_geom when the caller's workspace holds the per-ray tables (asurf_workspace_bytes with the
    // call's ray count); with a pyramid-only workspace, or where the one-kernel form is needed, the stepping walk
    const bool pipe = !config().eval_walk && check_grid(grid) == 0 && check_rays(rays) == 0 &&
                      rays->n_rays > 0 && workspace_bytes >= layout_for(grid->size, rays->n_rays, 0).total &&
                      use_pipeline(grid, rays, &eo, nullptr);
    Launch L;
    if (int e = prepare(grid, rays, &eo, 0, workspace, workspace_bytes, stream, L, !pipe)) return e;
    if (grid->size[0] > 1024 || grid->size[1] > 1024 || grid->size[2] > 1024)
        return fail(ASURF_E_INVALID, "render_eval: grid sizes above 1024 are not supported");
    if (rays->n_rays == 0) return 0;
    if (!out0 || (mode == 5 && (!out1 || max_sample < 1)))
        return fail(ASURF_E_INVALID, "render_eval: output tensor missing / max_sample < 1");
    switch (mode) {
        case 0: return launch_eval<EVAL_EXPECTED>(L, rays, param, max_sample, out0, out1, pipe);
        case 1: return launch_eval<EVAL_MODE_DEPTH>(L, rays, param, max_sample, out0, out1, pipe);
        case 2: return launch_eval<EVAL_FIRST_DEPTH>(L, rays, param, max_sample, out0, out1, pipe);
        case 3: return launch_eval<EVAL_FIRST_ALPHA>(L, rays, param, max_sample, out0, out1, pipe);
        case 4: return launch_eval<EVAL_NORMAL>(L, rays, param, max_sample, out0, out1, pipe);
        default: return launch_eval<EVAL_EXTRACT>(L, rays, param, max_sample, out0, out1, pipe);
    }
}

// ---- grid maintenance (asurf_misc.h) ----
int asurf_dilate(const uint8_t* grid, const int32_t size[3], uint8_t* out, void* stream) {
    if (!grid || !out || !size) return fail(ASURF_E_INVALID, "dilate: NULL tensor");
    const long long n = (long long)size[0] * size[1] * size[2];
    if (size[0] < 1 || size[1] < 1 || size[2] < 1 || n >= (1ll << 31)) return fail(ASURF_E_INVALID, "dilate: bad shape");
    hipLaunchKernelGGL(k_dilate, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       grid, out, size[0], size[1], size[2]);
    return launch_error("k_dilate");
}

size_t asurf_accel_dist_prop_tmp_bytes(const int32_t size[3]) {
    if (!size) return 0;
    long long sx = size[0], sy = size[1], sz = size[2], req = 0;
    while (sx > 1 && sy > 1 && sz > 1) {
        sx = div2_ceil((int)sx);
        sy = div2_ceil((int)sy);
        sz = div2_ceil((int)sz);
        req += sx * sy * sz;
    }
    return (size_t)req;
}

int asurf_accel_dist_prop(int32_t* links, const int32_t size[3], uint8_t* tmp, size_t tmp_bytes, void* stream) {
    if (!links || !size) return fail(ASURF_E_INVALID, "accel_dist_prop: NULL tensor");
    const long long n = (long long)size[0] * size[1] * size[2];
    if (size[0] < 1 || size[1] < 1 || size[2] < 1 || n >= (1ll << 31))
        return fail(ASURF_E_INVALID, "accel_dist_prop: bad shape");
    const size_t need = asurf_accel_dist_prop_tmp_bytes(size);
    if (need > 0 && (!tmp || tmp_bytes < need))
        return fail(ASURF_E_WORKSPACE, "accel_dist_prop: tmp of %zu bytes needed", need);
    const dim3 grid_dim((unsigned)((n + 255) / 256));
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL((k_accel_dist<true>), grid_dim, dim3(256), 0, s, links, tmp, size[0], size[1], size[2]);
    hipLaunchKernelGGL((k_accel_dist<false>), grid_dim, dim3(256), 0, s, links, tmp, size[0], size[1], size[2]);
    return launch_error("k_accel_dist");
}

namespace {
int sample_xf(const int32_t* links, const int32_t size[3], const float* offset, const float* scaling, SampleXf& X) {
    if (!links || !size || !offset || !scaling) return fail(ASURF_E_INVALID, "sample: NULL links / size / transform");
    for (int i = 0; i < 3; ++i) {
        if (size[i] < 2) return fail(ASURF_E_INVALID, "sample: links.shape[%d] = %d < 2", i, size[i]);
        X.size[i] = size[i];
        X.offset[i] = offset[i];
        X.scaling[i] = scaling[i];
    }
    return 0;
}
}  // namespace

int asurf_sample_plane(const int32_t* links, const int32_t size[3], const float* plane, int32_t n_cols,
                       float empty_value, const float* points, int64_t n_points, const float offset[3],
                       const float scaling[3], float* out, void* stream) {
    SampleXf X;
    if (int e = sample_xf(links, size, offset, scaling, X)) return e;
    if (n_points < 0 || n_cols < 1) return fail(ASURF_E_INVALID, "sample: n_points < 0 or n_cols < 1");
    if (n_points == 0) return 0;
    if (!plane || !points || !out) return fail(ASURF_E_INVALID, "sample: NULL tensor");
    const long long Q = (long long)n_points * n_cols;
    hipLaunchKernelGGL(k_sample_plane, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), links, X, plane, (int)n_cols, empty_value, points,
                       (long long)n_points, out);
    return launch_error("k_sample_plane");
}

int asurf_cubic_extract_iso_pts(const int32_t* links, const int32_t size[3], const float* level_data,
                                const float* mask_data, const int32_t* cell_ids, int64_t n_cells,
                                int32_t n_sample, float density_thresh, float* out, void* stream) {
    if (!links || !size || size[0] < 1 || size[1] < 1 || size[2] < 1)
        return fail(ASURF_E_INVALID, "cubic_extract_iso_pts: bad link grid");
    if (n_cells < 0 || n_sample < 2 || n_sample > 1024)
        return fail(ASURF_E_INVALID, "cubic_extract_iso_pts: n_cells < 0 or n_sample outside [2, 1024]");
    if (n_cells == 0) return 0;
    if (!level_data || !mask_data || !cell_ids || !out)
        return fail(ASURF_E_INVALID, "cubic_extract_iso_pts: NULL tensor");
    const long long threads = (long long)n_cells * n_sample * n_sample;
    if (threads > (1ll << 40)) return fail(ASURF_E_INVALID, "cubic_extract_iso_pts: batch too large");
    hipLaunchKernelGGL(k_iso_pts, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), links, (int)size[0], (int)size[1], (int)size[2],
                       level_data, mask_data, cell_ids, (long long)n_cells, (int)n_sample, density_thresh, out);
    return launch_error("k_iso_pts");
}

int asurf_sample_plane_backward(const int32_t* links, const int32_t size[3], int32_t n_cols,
                                const float* points, int64_t n_points, const float offset[3],
                                const float scaling[3], const float* grad_out, float* grad_plane,
                                void* stream) {
    SampleXf X;
    if (int e = sample_xf(links, size, offset, scaling, X)) return e;
    if (n_points < 0 || n_cols < 1) return fail(ASURF_E_INVALID, "sample backward: n_points < 0 or n_cols < 1");
    if (n_points == 0) return 0;
    if (!points || !grad_out || !grad_plane) return fail(ASURF_E_INVALID, "sample backward: NULL tensor");
    const long long Q = (long long)n_points * n_cols;
    hipLaunchKernelGGL(k_sample_plane_backward, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), links, X, (int)n_cols, points, (long long)n_points,
                       grad_out, grad_plane);
    return launch_error("k_sample_plane_backward");
}

// ---- weight renders (asurf_misc.h) ----
static int weight_render_common(const int32_t size[3], const asurf_camera* cam, float step_size,
                                const float* offset, const float* scaling, const float* out, CamK& K) {
    if (!size || !cam || !offset || !scaling || !out) return fail(ASURF_E_INVALID, "weight_render: NULL argument");
    const long long n = (long long)size[0] * size[1] * size[2];
    if (size[0] < 2 || size[1] < 2 || size[2] < 2 || n >= (1ll << 31))
        return fail(ASURF_E_INVALID, "weight_render: bad grid shape");
    if (!(step_size > 0.f)) return fail(ASURF_E_INVALID, "weight_render: step_size must be > 0");
    if (cam->width < 0 || cam->height < 0 || (long long)cam->width * cam->height >= (1ll << 31))
        return fail(ASURF_E_INVALID, "weight_render: bad image size");
    if (!(cam->fx != 0.f) || !(cam->fy != 0.f)) return fail(ASURF_E_INVALID, "weight_render: zero focal length");
    for (int i = 0; i < 12; ++i) K.c2w[i] = cam->c2w[i];
    K.fx = cam->fx;
    K.fy = cam->fy;
    K.cx = cam->cx;
    K.cy = cam->cy;
    K.width = cam->width;
    K.height = cam->height;
    return 0;
}

int asurf_grid_weight_render(const float* data, const int32_t size[3], const asurf_camera* cam, float step_size,
                             float stop_thresh, int32_t last_sample_opaque, const float offset[3],
                             const float scaling[3], float* grid_weight_out, void* stream) {
    CamK K;
    if (!data) return fail(ASURF_E_INVALID, "grid_weight_render: data is NULL");
    if (int e = weight_render_common(size, cam, step_size, offset, scaling, grid_weight_out, K)) return e;
    const long long Q = (long long)K.width * K.height;
    if (Q == 0) return 0;
    hipLaunchKernelGGL((k_weight_render<false>), dim3((unsigned)((Q + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), data, (const int32_t*)nullptr, size[0], size[1], size[2], K,
                       step_size, stop_thresh, last_sample_opaque ? 1 : 0, offset[0], offset[1], offset[2],
                       scaling[0], scaling[1], scaling[2], grid_weight_out);
    return launch_error("k_weight_render");
}

int asurf_sparse_grid_weight_render(const int32_t* links, const float* density_data, const int32_t size[3],
                                    const asurf_camera* cam, float step_size, float stop_thresh,
                                    const float offset[3], const float scaling[3], float* grid_weight_out,
                                    void* stream) {
    CamK K;
    if (!links || !density_data) return fail(ASURF_E_INVALID, "sparse_grid_weight_render: NULL tensor");
    if (int e = weight_render_common(size, cam, step_size, offset, scaling, grid_weight_out, K)) return e;
    const long long Q = (long long)K.width * K.height;
    if (Q == 0) return 0;
    hipLaunchKernelGGL((k_weight_render<true>), dim3((unsigned)((Q + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), density_data, links, size[0], size[1], size[2], K, step_size,
                       stop_thresh, 0, offset[0], offset[1], offset[2], scaling[0], scaling[1], scaling[2],
                       grid_weight_out);
    return launch_error("k_weight_render");
}

// ---- cuvol renderer (asurf_cuvol.h) ----
int asurf_volume_render_cuvol(const asurf_grid* grid, const asurf_rays* rays,
                              const asurf_render_options* opt, float* rgb_out, void* workspace,
                              size_t workspace_bytes, void* stream) {
    Launch L;
    if (opt && !(opt->step_size > 0.f)) return fail(ASURF_E_INVALID, "cuvol: step_size must be > 0");
    if (int e = prepare(grid, rays, opt, 0, workspace, workspace_bytes, stream, L, true, true)) return e;
    if (rays->n_rays == 0) return 0;
    if (!rgb_out) return fail(ASURF_E_INVALID, "rgb_out is NULL");
    int rc = 0;
    ASURF_DISPATCH_B(grid->basis_dim, (cuvol_fwd<B_>(L, rays, opt->last_sample_opaque, rgb_out, nullptr)));
    return rc;
}

int asurf_volume_render_cuvol_backward(const asurf_grid* grid, const asurf_rays* rays,
                                       const asurf_render_options* opt, const float* grad_out,
                                       const float* color_cache, const asurf_grads* grads, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    Launch L;
    if (opt && !(opt->step_size > 0.f)) return fail(ASURF_E_INVALID, "cuvol: step_size must be > 0");
    if (int e = prepare(grid, rays, opt, 0, workspace, workspace_bytes, stream, L, true, true)) return e;
    if (!grads) return fail(ASURF_E_INVALID, "grads is NULL");
    if (rays->n_rays == 0) return 0;
    if (!grad_out || !color_cache) return fail(ASURF_E_INVALID, "grad_out / color_cache is NULL");
    const KGrads G = make_grads(grads);
    int rc = 0;
    ASURF_DISPATCH_B(grid->basis_dim, (cuvol_bwd<B_>(L, rays, opt->last_sample_opaque, G, grad_out, color_cache, 0,
                                                     nullptr, 0.f, 0.f)));
    return rc;
}

// render_lerp_kernel_cuvol.cu:1272-1354: forward, then backward with the MSE gradient formed in the
// kernel; beta_loss / Q as there (:1323); the log-transmittance hand-over lives in the workspace.
int asurf_volume_render_cuvol_fused(const asurf_grid* grid, const asurf_rays* rays,
                                    const asurf_render_options* opt, const float* rgb_gt, float beta_loss,
                                    float sparsity_loss, float* rgb_out, const asurf_grads* grads,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    Launch L;
    if (opt && !(opt->step_size > 0.f)) return fail(ASURF_E_INVALID, "cuvol: step_size must be > 0");
    if (int e = prepare(grid, rays, opt, 0, workspace, workspace_bytes, stream, L, false, true)) return e;
    if (!grads) return fail(ASURF_E_INVALID, "grads is NULL");
    if (rays->n_rays == 0) return 0;
    if (!rgb_gt || !rgb_out) return fail(ASURF_E_INVALID, "rgb_gt / rgb_out is NULL");
    const KGrads G = make_grads(grads);
    float* log_t = beta_loss > 0.f ? reinterpret_cast<float*>(L.ws + L.ML.off_cache_n) : nullptr;
    int rc = 0;
    ASURF_DISPATCH_B(grid->basis_dim, (cuvol_fwd<B_>(L, rays, opt->last_sample_opaque, rgb_out, log_t)));
    if (rc) return rc;
    ASURF_DISPATCH_B(grid->basis_dim, (cuvol_bwd<B_>(L, rays, opt->last_sample_opaque, G, rgb_gt, rgb_out, 1, log_t,
                                                     beta_loss / (float)rays->n_rays, sparsity_loss)));
    return rc;
}

// render_lerp_kernel_cuvol.cu:1356-1442: mode 0 expected_term, 1 mode_term (param = weight_thresh), 2 med_term
// (out0 = depths, out1 = sigmas, [Q, max_sample], zeroed by the caller), 3 sigma_thresh (param = the threshold)
int asurf_render_eval_cuvol(const asurf_grid* grid, const asurf_rays* rays, const asurf_render_options* opt,
                            int32_t mode, float param, int32_t max_sample, float* out0, float* out1,
                            void* workspace, size_t workspace_bytes, void* stream) {
    if (mode < 0 || mode > 3) return fail(ASURF_E_INVALID, "render_eval_cuvol: mode must be 0..3");
    if (!opt) return fail(ASURF_E_INVALID, "opt is NULL");
    if (!(opt->step_size > 0.f)) return fail(ASURF_E_INVALID, "cuvol: step_size must be > 0");
    asurf_render_options o2 = *opt;
    if (mode == 3) o2.sigma_thresh = param;
    Launch L;
    if (int e = prepare(grid, rays, &o2, 0, workspace, workspace_bytes, stream, L, true, true)) return e;
    if (rays->n_rays == 0) return 0;
    if (!out0 || (mode == 2 && (!out1 || max_sample < 1)))
        return fail(ASURF_E_INVALID, "render_eval_cuvol: output tensor missing");
    switch (mode) {
        case 0: return cuvol_eval<0>(L, rays, param, max_sample, out0, out1);
        case 1: return cuvol_eval<1>(L, rays, param, max_sample, out0, out1);
        case 2: return cuvol_eval<2>(L, rays, param, max_sample, out0, out1);
        default: return cuvol_eval<3>(L, rays, param, max_sample, out0, out1);
    }
}

int asurf_test_quotient(const float* x, const float* d, int64_t n, float* fast_out, float* ieee_out, void* stream) {
    if (n < 0 || (n > 0 && (!x || !d || !fast_out || !ieee_out))) return fail(ASURF_E_INVALID, "test_quotient: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_test_quotient, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, d, (long long)n, fast_out, ieee_out);
    return launch_error("k_test_quotient");
}

}  // extern "C"
