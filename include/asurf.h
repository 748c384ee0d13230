/*
 * asurf.h -- C ABI of the MI355X-native alphaSurf hot path (libasurf_hip.so).
 *
 * This is the drop-in boundary of the build: plain pointers and sizes, no torch
 * types.  Every entry point names the reference interface it replaces; paths are
 * relative to the reference checkout (ChikaYan/alphasurf).
 *
 *   reference pybind module      svox2/csrc/svox2.cpp:139-291   (`svox2.csrc`)
 *   reference structs            svox2/csrc/include/data_spec.hpp:39-201
 *
 * Conventions
 *   - All `const float*` / `int32_t*` / `uint8_t*` members are DEVICE pointers
 *     (HBM resident, contiguous, row-major) unless a comment says HOST.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls
 *     are asynchronous with respect to the host, exactly like the reference
 *     launchers (no implicit synchronisation).
 *   - Every call returns 0 on success and a negative ASURF_E_* code on failure;
 *     asurf_last_error() returns a thread-local human readable message.  The
 *     Python binding turns a non-zero return into RuntimeError, mirroring the
 *     TORCH_CHECK -> RuntimeError behaviour of the reference
 *     (svox2/csrc/include/util.hpp:3-13, data_spec.hpp:58-81).
 *   - Gradient buffers are ACCUMULATED into (+=), never zeroed, and mask_out is
 *     OR-ed, as in the reference (render_util.cuh:94-154, 1794-1821).
 *   - No entry point allocates device memory.  Scratch comes from the caller
 *     through (workspace, workspace_bytes); query the size with
 *     asurf_workspace_bytes().  This keeps the launchers hipGraph-capturable.
 */
#ifndef ASURF_H_
#define ASURF_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASURF_ABI_VERSION 4

/* error codes */
#define ASURF_OK 0
#define ASURF_E_INVALID (-1)     /* bad argument (the TORCH_CHECK class of errors) */
#define ASURF_E_UNSUPPORTED (-2) /* feature of the reference ABI outside this build's scope */
#define ASURF_E_WORKSPACE (-3)   /* workspace missing / too small */
#define ASURF_E_LAUNCH (-4)      /* HIP launch error */

/* enums shared with the reference (data_spec.hpp:8-36) */
#define ASURF_BASIS_TYPE_SH 1
#define ASURF_BASIS_TYPE_3D_TEXTURE 4
#define ASURF_BASIS_TYPE_MLP 255
#define ASURF_SURFACE_TYPE_NONE 100
#define ASURF_SURFACE_TYPE_SDF 102
#define ASURF_SURFACE_TYPE_UDF 103
#define ASURF_CUBIC_TYPE_NO_ROOT 200
#define ASURF_CUBIC_TYPE_LINEAR 201
#define ASURF_CUBIC_TYPE_POLY_ONE_R 202
#define ASURF_CUBIC_TYPE_POLY 203
#define ASURF_CUBIC_TYPE_CUBIC_THREE_R 205
#define ASURF_CUBIC_TYPE_CUBIC_ONE_R_ 206
#define ASURF_SIGMOID_FN 0
#define ASURF_EXP_FN 1

/* Replaces SparseGridSpec / PackedSparseGridSpec
 * (data_spec.hpp:39-82, data_spec_packed.cuh:11-74). */
typedef struct asurf_grid {
    const int32_t* links;        /* [size0,size1,size2] int32, value = row id or <0 (empty) */
    const float* density_data;   /* [capacity, 1] raw opacity */
    const float* surface_data;   /* [capacity, 1] SDF / UDF scalar */
    const float* sh_data;        /* [capacity, 3*basis_dim] */
    const float* level_set_data; /* [level_set_num] */
    int64_t capacity;            /* rows in the data tensors (N) */
    int32_t size[3];             /* links.shape */
    int32_t level_set_num;
    int32_t basis_dim;           /* 1, 4, 9 or 16 SH functions per colour channel */
    int32_t basis_type;          /* ASURF_BASIS_TYPE_SH only */
    int32_t surface_type;
    float offset[3];             /* HOST values: _offset * gsz (svox2.py:6250-6253) */
    float scaling[3];            /* HOST values: _scaling * gsz */
    float fake_sample_std;
    float truncated_vol_render_a;
    int32_t background_nlayers;  /* must be 0: MSI background is outside this build's scope */
    /* 0, or a promise that saves most of the occupancy-pyramid pass: a non-zero value says "since the last call that
     * carried this same value on this same workspace buffer, neither the contents of `links` nor the buffer (not
     * freed, not written by the caller) have changed".  The pass then keeps the list of the rows of bricks that hold
     * any vertex -- the rest of the link volume is empty space it need not read again.  A training loop changes
     * densities and surfaces in every iteration and `links` only when it prunes or up-samples: pass a counter
     * bumped there, mixed with one bumped whenever the workspace is reallocated.  (No reference counterpart: the
     * reference walks `links` in every kernel.)  The first call with a new value runs the full pass and reads one
     * counter back asynchronously; on a stream that is being captured into a graph a new value is not learnt (the
     * full pass runs, as with 0).
     * With 0 the call vouches for the list ITSELF (round 4; grids from 128^3 to below 512^3, not while capturing): it
     * sums all of `links` into a 128-bit checksum (0.017 ms at 256^3) and the pass over the listed rows compares, on the
     * device, with the checksum the list was built under -- 0.052 ms against the full pass's 0.064, against 0.035 with the
     * promise.  `links` rewritten in place make that very call take every row (and the next one rebuild list and sum);
     * ASURF_SUM_LIST=0 keeps the full pass. */
    uint64_t links_epoch;
} asurf_grid;

/* Replaces RaysSpec / PackedRaysSpec (data_spec.hpp:142-154). */
typedef struct asurf_rays {
    const float* origins;  /* [n_rays,3] world space */
    const float* dirs;     /* [n_rays,3] world space, pre-normalised by the caller */
    const uint8_t* masks;  /* [n_rays] bool; carried for ABI parity, unused (surf_trav.cu:1715) */
    int64_t n_rays;
    /* extension (ABI 4; 0: unknown): > 0 states that the rays are a row-major image (or a stack of images) of this many
     * rays a row.  The cuvol kernels then give a wave an 8 x 8 pixel tile instead of 64 pixels of a row and serve a
     * round's records in the order of their cells: a tile's rays pass through the same few cells, whose SH rows are
     * then fetched once a round and whose gradient rows leave merged.  (The surf_trav fused call takes the same hint
     * through asurf_fused_params::image_width, or from here when that is 0.) */
    int32_t image_width;
    int32_t pad_;
} asurf_rays;

/* Replaces RayVoxIntersecSpec (data_spec.hpp:156-166), legacy `surface` backend only. */
typedef struct asurf_ray_vox {
    const int32_t* voxel_ls;    /* [n_vox,3] */
    const int32_t* vox_start_i; /* [n_rays] */
    const int32_t* vox_num;     /* [n_rays] */
    int64_t n_vox;
} asurf_ray_vox;

/* Replaces RenderOptions (data_spec.hpp:168-201); same field order. */
typedef struct asurf_render_options {
    float background_brightness;
    float step_size;
    float sigma_thresh;
    float stop_thresh;
    float near_clip;
    int32_t use_spheric_clip;
    int32_t last_sample_opaque;
    int32_t surf_fake_sample;
    float surf_fake_sample_min_vox_len;
    int32_t limited_fake_sample;
    int32_t no_surf_grad_from_sh;
    int32_t alpha_activation_type;
    int32_t fake_sample_l_dist;
    int32_t fake_sample_normalize_surf;
    int32_t only_outward_intersect;
    int32_t truncated_vol_render;
    float trunc_vol_weight_min;
} asurf_render_options;

/* Replaces GridOutputGrads / PackedGridOutputGrads (data_spec.hpp:84-120). Any
 * pointer may be NULL (the matching gradient is then skipped). */
typedef struct asurf_grads {
    float* grad_density_out;         /* [capacity,1] */
    float* grad_sh_out;              /* [capacity,3*basis_dim] */
    float* grad_surface_out;         /* [capacity,1] */
    float* grad_fake_sample_std_out; /* [1] */
    uint8_t* mask_out;               /* [capacity] bool */
    /* extension (ABI 4; NULL: none): a second mask that receives the same marks.  The reference's training loop copies
     * the render's mask for the SH step before the regularisers add their rows to it (svox2.py: sparse_sh_grad_indexer
     * = sparse_grad_indexer.clone()): a launch of its own in front of them, 5 us of a 0.25 ms iteration -- with two
     * fresh masks handed to the fused call the copy is gone. */
    uint8_t* mask_sh_out;
} asurf_grads;

/* The 20 scalars + rgb_gt of volume_render_surf_trav_fused
 * (svox2.cpp:38-40; surf_trav.cu:3802-3828), same order. */
typedef struct asurf_fused_params {
    const float* rgb_gt; /* [n_rays,3] */
    float beta_loss;
    float sparsity_loss;
    float fused_surf_norm_reg_scale; /* must be 0 (reference asserts, surf_trav.cu:2872-2875) */
    int32_t fused_surf_norm_reg_con_check;
    int32_t fused_surf_norm_reg_ignore_empty;
    float lambda_l2;
    float lambda_l1;
    float lambda_l_dist;
    float lambda_l_entropy;
    int32_t no_norm_weight_l_entropy;
    float lambda_l_dist_a;
    float lambda_l_entropy_a;
    float lambda_l_samp_dist;
    float lambda_l_di;
    float l_di_alpha_thresh;
    float surf_sparse_alpha_thresh;
    float lambda_inplace_surf_sparse;
    float lambda_inwards_norm_loss;
    float lambda_conv_mode_samp;
    int32_t l_dist_max_sample;
    /* Multi-GPU extension (no reference counterpart): the loss normaliser 1/(3Q)
     * and the /Q lambdas (surf_trav.cu:3308-3309, 3896-3908) use this GLOBAL ray
     * count when > 0, else rays.n_rays. */
    int64_t global_n_rays;
    /* Ray ordering (extension; the reference renders a batch in the order given).  A shuffled batch costs the kernels
     * that run one ray per lane their coherence (361 against 650 Mrays/s at 640 k rays in round 2).  For batches of
     * 65 536 rays and more the call can put the rays in a coherent order ITSELF -- a 32-bit key per ray (origin hash,
     * Morton code of the direction), one radix sort, one gather of rays and targets into the workspace -- render
     * them, and scatter the colours back to where the caller's rays are: rgb_out, the gradients and the masks are
     * those of the given order (up to the order of float atomics, as always).
     *   0: automatic -- ordered when the LAST batch rendered through this workspace arrived shuffled (a probe of
     *      consecutive ray pairs, read back asynchronously: the first call renders as given, and so does a call that
     *      is being captured into a graph -- its replays must not depend on when it was captured).  Which internal
     *      order a call takes therefore depends on timing; the results differ by the order of float atomics only;
     *   1: never;  2: always. */
    int32_t order_rays;
    /* Image batches (extension): > 0 states that the batch is a row-major image (or a stack of images) of this many
     * rays a row, n_rays a multiple of it; 0: unknown.  The backward pass of batches that fill the chip (64 rays a
     * wave) then lists the samples it replays by 8 x 8 PIXEL TILES instead of by runs of 64 pixels of a row
     * (k_bwd_seq; the forward kernels and every result are untouched): a chunk of that list -- what a wave of
     * k_bwd_sh / k_bwd_geo sorts by voxel and merges before it adds to memory -- is then a patch of the surface
     * instead of a line across it, which falls into half as many voxels.  Measured on MI355X (config 2): k_bwd_sh
     * 0.171 -> 0.124 ms, k_bwd_geo 0.107 -> 0.083 ms, the call 730 -> 800 Mrays/s; the tile's width is
     * ASURF_BWD_TILE (4 / 8 / 16 / 32; 0: rows), images it does not partition are taken by rows.
     * ASURF_TILES=1 (off: measured slower) renders the WHOLE call in tile order -- the rays copied inside the
     * call, the colours scattered back.  A batch declared an image is also taken as coherent: under order_rays = 0
     * the call neither consults nor runs the coherence probe (one kernel, one copy to the host and one event less a
     * call); order_rays = 2 still orders it.  (The field was `reserved_`, always 0, in ABI version 3: same layout.) */
    int32_t image_width;
} asurf_fused_params;

/* Per-call counters (optional; pass NULL to skip).  Device buffer of
 * ASURF_N_COUNTERS uint64, accumulated with atomics by the kernels when non-NULL:
 * used by bench.py to price the algorithmic bytes of SURVEY.md section 8(d). */
#define ASURF_N_COUNTERS 4
#define ASURF_CNT_TRAVERSED 0 /* V_t: voxels visited by the DDA */
#define ASURF_CNT_ACTIVE 1    /* V_a: voxels with 8 live links that pass the density gate */
#define ASURF_CNT_SAMPLES 2   /* S: samples composited (real + fake) */
#define ASURF_CNT_RAYS 3

int asurf_abi_version(void);
const char* asurf_last_error(void);

/* Scratch bytes needed by any of the surf_trav / surface entry points below for
 * this grid size / ray count / sample-cache depth. */
size_t asurf_workspace_bytes(const int32_t size[3], int64_t n_rays, int32_t l_dist_max_sample);

/* The same with room for `rec_depth` sample records a ray (32, 64 or 128; rounded up, at most 128).  The fused call
 * keeps a ray's samples for its backward pass (the reference walks the ray a second time, surf_trav.cu:1925-3140); a
 * ray with more samples than the workspace has records for is differentiated by the walking kernel instead -- same
 * result, a hundred times the time.  asurf_workspace_bytes() is this with 32: ample for the live configuration (0.7
 * samples a ray), short for fake samples (surf_fake_sample: 11 a ray on the bench scene, a fifth of the rays beyond 32).  A call uses the
 * deepest records the workspace it is handed has room for: the depth is a property of the buffer, not of the call. */
size_t asurf_workspace_bytes_rec(const int32_t size[3], int64_t n_rays, int32_t l_dist_max_sample, int32_t rec_depth);

/* Tensor volume_render_surf_trav(SparseGridSpec&, RaysSpec&, RenderOptions&)
 *   svox2.cpp:33, surf_trav.cu:3596-3654.  rgb_out: [n_rays,3] (written). */
int asurf_volume_render_surf_trav(const asurf_grid* grid, const asurf_rays* rays,
                                  const asurf_render_options* opt, float* rgb_out,
                                  void* workspace, size_t workspace_bytes,
                                  uint64_t* counters, void* stream);

/* void volume_render_surf_trav_backward(grid, rays, opt, grad_out, color_cache, grads)
 *   svox2.cpp:36-37, surf_trav.cu:3708-3800. */
int asurf_volume_render_surf_trav_backward(const asurf_grid* grid, const asurf_rays* rays,
                                           const asurf_render_options* opt,
                                           const float* grad_out, const float* color_cache,
                                           const asurf_grads* grads, void* workspace,
                                           size_t workspace_bytes, uint64_t* counters,
                                           void* stream);

/* void volume_render_surf_trav_fused(grid, rays, opt, rgb_gt, 20 scalars, rgb_out, grads)
 *   svox2.cpp:38-40, surf_trav.cu:3802-3942. */
int asurf_volume_render_surf_trav_fused(const asurf_grid* grid, const asurf_rays* rays,
                                        const asurf_render_options* opt,
                                        const asurf_fused_params* fused, float* rgb_out,
                                        const asurf_grads* grads, void* workspace,
                                        size_t workspace_bytes, uint64_t* counters,
                                        void* stream);

/* Legacy `surface` backend (precomputed per-ray voxel lists):
 *   volume_render_surface / _backward / _fused, svox2.cpp:25-31,
 *   render_lerp_kernel_surface.cu:1198-1439. */
int asurf_volume_render_surface(const asurf_grid* grid, const asurf_rays* rays,
                                const asurf_ray_vox* ray_vox, const asurf_render_options* opt,
                                float* rgb_out, void* stream);
int asurf_volume_render_surface_backward(const asurf_grid* grid, const asurf_rays* rays,
                                         const asurf_ray_vox* ray_vox,
                                         const asurf_render_options* opt, const float* grad_out,
                                         const float* color_cache, const asurf_grads* grads,
                                         void* stream);
int asurf_volume_render_surface_fused(const asurf_grid* grid, const asurf_rays* rays,
                                      const asurf_ray_vox* ray_vox,
                                      const asurf_render_options* opt, const float* rgb_gt,
                                      float beta_loss, float sparsity_loss, float* rgb_out,
                                      const asurf_grads* grads, void* stream);

/* Per-ray voxel lists for the legacy backend, produced by the surf_trav DDA
 * (surf_trav.cu:75-197).  Stands in for SparseGrid.find_ray_voxels_intersection
 * (svox2.py:1515-1594, callers :3421, :3572), whose kaolin SPC ray trace cannot be built in the
 * reference.  Two passes: asurf_ray_voxel_count fills vox_num[n_rays]; the caller takes the
 * exclusive scan (vox_start_i) and the total (n_vox), then asurf_ray_voxel_fill writes
 * voxel_ls_out[n_vox,3].  skip_empty != 0 leaves out voxels with a missing link (the renderer
 * skips them anyway); both passes must use the same value. */
int asurf_ray_voxel_count(const asurf_grid* grid, const asurf_rays* rays,
                          const asurf_render_options* opt, int32_t skip_empty, int32_t* vox_num,
                          void* stream);
int asurf_ray_voxel_fill(const asurf_grid* grid, const asurf_rays* rays,
                         const asurf_render_options* opt, int32_t skip_empty,
                         const asurf_ray_vox* ray_vox, int32_t* voxel_ls_out, void* stream);

/* Tensor test_cubic_root_grad(fs f64[N,4], st_ids i32[N], grad_fs f32[N,4] in/out,
 *                             st f32[N,3] out, bool use_vieta)
 *   svox2.cpp:12, test_cuda.cu:32-122.  All pointers are device pointers. */
int asurf_test_cubic_root_grad(const double* fs, const int32_t* st_ids, float* grad_fs,
                               float* st, int64_t n, int32_t use_vieta, void* stream);

/* void rmsprop_step(data, rms, grad, indexer, beta, lr, epsilon, minval, lr_last)
 * void sgd_step(data, grad, indexer, lr, lr_last)
 *   svox2.cpp (REG rmsprop_step / sgd_step), optim_kernel.cu:154-265.  data/rms/grad are
 *   [n_rows, n_cols] f32.  The reference's `indexer` tensor becomes: mask != NULL -> bool row
 *   mask [n_rows]; else indices != NULL -> int64 row ids [n_indices] (n_indices == 0 is a no-op,
 *   as in the reference); both NULL -> every row.  rms: rms==0 ? g^2 :
 *   lerp(g^2, rms, beta); data = max(data - lr*g/(sqrt(rms)+eps), minval); grad = 0.  The last
 *   column uses lr_last (lr_last < 0 means lr). */
int asurf_rmsprop_step(float* data, float* rms, float* grad, int64_t n_rows, int32_t n_cols,
                       const uint8_t* mask, const int64_t* indices, int64_t n_indices, float beta,
                       float lr, float epsilon, float minval, float lr_last, void* stream);
int asurf_sgd_step(float* data, float* grad, int64_t n_rows, int32_t n_cols, const uint8_t* mask,
                   const int64_t* indices, int64_t n_indices, float lr, float lr_last, void* stream);

/* The same steps for a bool row mask, without the reference's host-side choice between mask and index list
 * (svox2.py:6314-6333: `torch.count_nonzero(mask).item()`, a device -> host round trip per step): the mask is
 * compacted into a row list on the device and the index-list kernel strides over it, its length read on the
 * device.  Same rows, same arithmetic as asurf_rmsprop_step / asurf_sgd_step with `mask`.  `scratch`: device
 * memory of asurf_optim_scratch_bytes(n_rows) bytes, contents irrelevant before and after the call. */
size_t asurf_optim_scratch_bytes(int64_t n_rows);
int asurf_rmsprop_step_masked(float* data, float* rms, float* grad, int64_t n_rows, int32_t n_cols,
                              const uint8_t* mask, void* scratch, size_t scratch_bytes, float beta, float lr,
                              float epsilon, float minval, float lr_last, void* stream);
int asurf_sgd_step_masked(float* data, float* grad, int64_t n_rows, int32_t n_cols, const uint8_t* mask,
                          void* scratch, size_t scratch_bytes, float lr, float lr_last, void* stream);

/* Evaluation renders: traversal + cubic core without colour, density
 * gate, window or fake samples.  `workspace` as for the training calls (asurf_workspace_bytes with
 * l_dist_max_sample = 0).  mode / param / outputs:
 *   0 volume_render_expected_term_surf_trav  svox2.cpp REG, surf_trav.cu:564-794, 3944-3962   out0[Q]
 *   1 volume_render_mode_term_surf_trav      :796-1000, 3964-3983   param = weight_thresh     out0[Q]
 *   2 volume_render_sigma_thresh_surf_trav   :1003-1168, 3985-4006  param = alpha threshold   out0[Q]
 *   3 volume_render_alpha_surf_trav          :1170-1335, 4008-4029  param = alpha threshold   out0[Q]
 *   4 render_normal_surf_trav                :1339-1531, 4031-4051                            out0[Q,3]
 *   5 extract_pts_surf_trav                  :1536-1705, 4053-4081  param = alpha_thresh;
 *       out0 = depths, out1 = alphas, both [Q,max_sample] and ZEROED by the caller. */
int asurf_render_eval_surf_trav(const asurf_grid* grid, const asurf_rays* rays,
                                const asurf_render_options* opt, int32_t mode, float param,
                                int32_t max_sample, float* out0, float* out1, void* workspace,
                                size_t workspace_bytes, void* stream);

/* tv -- svox2.cpp:114, loss_kernel.cu:73-117, 1214-1247: the dense TV value (sum over channels
 * [start_dim, end_dim), mean over the (X-1)(Y-1)(Z-1) cells) ADDED to the device float *out (the
 * reference returns a fresh zero tensor: the caller zeroes).  use_logalpha / ndc arguments of the
 * reference are dead in its kernel and not part of this signature. */
int asurf_tv(const int32_t* links, const int32_t size[3], const float* data, int32_t n_cols,
             int32_t start_dim, int32_t end_dim, int32_t ignore_edge, float* out, void* stream);

/* Per-iteration regularisers.  `links` is the int32[size0,size1,size2]
 * link grid, `rand_cells` a device list of n_cells flat cell indices (svox2.py:6335-6383),
 * `mask_out` the bool[N] sparse_grad_indexer or NULL; gradients are ADDED to grad_*.
 * `scale` is the caller's weight: the launchers divide it by n_cells as the reference's do.
 *   tv_grad_sparse                  svox2.cpp (REG tv_grad_sparse), loss_kernel.cu:739-808, 1327-1373
 *   surf_tv_grad_sparse             loss_kernel.cu:810-895, 1375-1427
 *   surface_normal_grad_sparse      loss_kernel.cu:398-436, 1572-1622; render_util.cuh:1823-2130
 *   alpha_surf_sparsify_grad_sparse loss_kernel.cu:665-738, 1512-1570 (scales used as given)
 *   surface_eikonal_grad_sparse     the reference evaluates this loss with PyTorch autograd
 *                                   (svox2.py:5433-5472, called from opt.py:1002-1009); *loss_out
 *                                   (device float, may be NULL) is incremented by the loss value. */
int asurf_tv_grad_sparse(const int32_t* links, const int32_t size[3], const float* data, int32_t n_cols,
                         const int32_t* rand_cells, int64_t n_cells, uint8_t* mask_out,
                         int32_t start_dim, int32_t end_dim, float scale, int32_t ignore_edge,
                         int32_t ignore_last_z, float* grad_data, void* stream);
int asurf_surf_tv_grad_sparse(const int32_t* links, const int32_t size[3], const float* data,
                              int32_t n_cols, const float* density_data, int32_t density_cols,
                              const int32_t* rand_cells, int64_t n_cells, uint8_t* mask_out,
                              int32_t start_dim, int32_t end_dim, float scale, int32_t ignore_edge,
                              float edge_value, int32_t ignore_last_z, int32_t alpha_dependency,
                              float* grad_data, void* stream);
int asurf_surface_normal_grad_sparse(const int32_t* links, const int32_t size[3],
                                     const float* surface_data, const int32_t* rand_cells,
                                     int64_t n_cells, uint8_t* mask_out, float lv_set, float scale,
                                     int32_t con_check, int32_t ignore_empty, int32_t use_l1,
                                     float* grad_data, void* stream);
int asurf_alpha_surf_sparsify_grad_sparse(const int32_t* links, const int32_t size[3],
                                          const float* alpha_data, const float* surf_data,
                                          const int32_t* rand_cells, int64_t n_cells, uint8_t* mask_out,
                                          float scale_alpha, float scale_surf, int32_t surf_sparse_decrease,
                                          float surf_sparse_thresh, float alpha_bound, float surf_bound,
                                          float* grad_alpha, float* grad_surf, void* stream);
int asurf_surface_eikonal_grad_sparse(const int32_t* links, const int32_t size[3],
                                      const float* surface_data, const int32_t* rand_cells,
                                      int64_t n_cells, uint8_t* mask_out, float scale, float h,
                                      float* grad_data, float* loss_out, void* stream);

/* Grid maintenance.
 *   dilate           svox2.cpp REG dilate, misc_kernel.cu:24-58, 1005-1020: 3x3x3 dilation of a bool grid.
 *   accel_dist_prop  misc_kernel.cu:113-180, 1022-1061: every empty link becomes -(1 + levels of the
 *                    power-of-two pyramid that are still empty around it); `tmp` is scratch of
 *                    asurf_accel_dist_prop_tmp_bytes(size) bytes, ZEROED by the caller. */
int asurf_dilate(const uint8_t* grid, const int32_t size[3], uint8_t* out, void* stream);
size_t asurf_accel_dist_prop_tmp_bytes(const int32_t size[3]);
int asurf_accel_dist_prop(int32_t* links, const int32_t size[3], uint8_t* tmp, size_t tmp_bytes,
                          void* stream);

/* The Plenoxels volume renderer (`cuvol` backend):
 *   volume_render_cuvol / _backward / _fused, svox2.cpp REG, render_lerp_kernel_cuvol.cu:1120-1354.
 * surface_data may be NULL.  Workspace: asurf_workspace_bytes(size, 0, 0) for the first two,
 * asurf_workspace_bytes(size, n_rays, 0) for the fused call.  The MSI background, learned bases and
 * the random density perturbation of the reference are not built. */
int asurf_volume_render_cuvol(const asurf_grid* grid, const asurf_rays* rays,
                              const asurf_render_options* opt, float* rgb_out, void* workspace,
                              size_t workspace_bytes, void* stream);
int asurf_volume_render_cuvol_backward(const asurf_grid* grid, const asurf_rays* rays,
                                       const asurf_render_options* opt, const float* grad_out,
                                       const float* color_cache, const asurf_grads* grads, void* workspace,
                                       size_t workspace_bytes, void* stream);
int asurf_volume_render_cuvol_fused(const asurf_grid* grid, const asurf_rays* rays,
                                    const asurf_render_options* opt, const float* rgb_gt, float beta_loss,
                                    float sparsity_loss, float* rgb_out, const asurf_grads* grads,
                                    void* workspace, size_t workspace_bytes, void* stream);

/* Depth renders of the cuvol march (render_lerp_kernel_cuvol.cu:1356-1442).  mode 0
 * volume_render_expected_term and 1 volume_render_mode_term (param = weight_thresh; out0[Q]),
 * 2 volume_render_med_term (out0 = depths, out1 = sigmas, both [Q, max_sample] and zeroed by the
 * caller), 3 volume_render_sigma_thresh (param = the threshold; out0[Q]).  Workspace as for
 * asurf_volume_render_cuvol. */
int asurf_render_eval_cuvol(const asurf_grid* grid, const asurf_rays* rays,
                            const asurf_render_options* opt, int32_t mode, float param,
                            int32_t max_sample, float* out0, float* out1, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Pinhole camera of the image-order entry points (CameraSpec, data_spec.hpp:84-112): c2w holds the
 * rows of the 3x4 camera-to-world matrix (HOST values); NDC cameras are not built. */
typedef struct asurf_camera {
    float c2w[12];
    float fx, fy, cx, cy;
    int32_t width, height;
} asurf_camera;

/* grid_weight_render (svox2.cpp REG, misc_kernel.cu:186-284, 888-913, 1084-1111): marches every pixel
 * ray of `cam` through the DENSE density volume data[X,Y,Z] and keeps, per vertex, the largest
 * compositing weight of a sample in one of its 8 cells (max into grid_weight_out[X,Y,Z], which the
 * caller zeroes before the first camera). */
int asurf_grid_weight_render(const float* data, const int32_t size[3], const asurf_camera* cam,
                             float step_size, float stop_thresh, int32_t last_sample_opaque,
                             const float offset[3], const float scaling[3], float* grid_weight_out,
                             void* stream);
/* sparse_grid_weight_render (misc_kernel.cu:286-400, 915-937, 1113-1138): the same march through a
 * sparse grid (missing corners read 0); the vertex keeps the largest TRANSMITTANCE in front of a sample. */
int asurf_sparse_grid_weight_render(const int32_t* links, const float* density_data,
                                    const int32_t size[3], const asurf_camera* cam, float step_size,
                                    float stop_thresh, const float offset[3], const float scaling[3],
                                    float* grid_weight_out, void* stream);

/* Test hook: the walk's exit-plane quotient x / d formed from RN(1/d) (asurf_kernels.h `quotient`) and the
 * IEEE division, element by element, for a bitwise comparison on the device. */
int asurf_test_quotient(const float* x, const float* d, int64_t n, float* fast_out, float* ieee_out,
                        void* stream);

/* Ray ordering for coherence (new; the reference renders a batch in the order it is given).  Writes a
 * permutation perm_out[n_rays] (int32) that groups rays of equal origin and orders each group along a
 * Morton curve of the unit direction: gather the rays (and rgb_gt) with it, render, scatter rgb_out back.
 * tmp: asurf_ray_order_tmp_bytes(n_rays) bytes of device scratch. */
size_t asurf_ray_order_tmp_bytes(int64_t n_rays);
int asurf_ray_order(const float* origins, const float* dirs, int64_t n_rays, int32_t* perm_out, void* tmp,
                    size_t tmp_bytes, void* stream);

/* Trilinear sampling of an [n_rows, n_cols] parameter plane at `points` float[n_points,3]
 * (sample_grid / sample_grid_sh_surf / sample_grid_raw_alpha, svox2_kernel.cu:11-150, 384-489):
 * p * scaling + offset, clamped to the grid, an empty corner reads as `empty_value`; out is
 * float[n_points, n_cols].  The backward (sample_grid_backward, :151-246, 491-540) ADDS
 * d out / d plane * grad_out into grad_plane. */
int asurf_sample_plane(const int32_t* links, const int32_t size[3], const float* plane, int32_t n_cols,
                       float empty_value, const float* points, int64_t n_points, const float offset[3],
                       const float scaling[3], float* out, void* stream);
int asurf_sample_plane_backward(const int32_t* links, const int32_t size[3], int32_t n_cols,
                                const float* points, int64_t n_points, const float offset[3],
                                const float scaling[3], const float* grad_out, float* grad_plane,
                                void* stream);

/* cubic_extract_iso_pts -- svox2.cpp:21-22, svox2_kernel.cu:248-372, 542-583: for each listed cell whose 8
 * vertices exist, n_sample x n_sample axis-parallel lines per direction; the first root in [0,1] of
 * `level_data` (float[N], surface minus level set) along a line whose trilinear `mask_data` (float[N], the
 * density) reaches density_thresh becomes a grid-space point of out float[n_cells, 3 n_sample^2, 3]
 * (rows without a point stay as the caller's zeros; cell ids with a coordinate on the last layer are skipped). */
int asurf_cubic_extract_iso_pts(const int32_t* links, const int32_t size[3], const float* level_data,
                                const float* mask_data, const int32_t* cell_ids, int64_t n_cells,
                                int32_t n_sample, float density_thresh, float* out, void* stream);

/* Measurement hooks (no reference counterpart; used by bench.py).  When enabled, the
 * surf_trav launchers record a hipEvent on the call's stream after every kernel launch.
 * asurf_profile_read() waits for the last call and returns, in milliseconds,
 * ms[0] = occupancy-pyramid pass, ms[1] = sum of the forward kernels, ms[2] = sum of the
 * backward kernels (0 for a group the call did not launch).
 * asurf_profile_read_spans() returns the per-kernel times of the last call: the number of
 * spans (or -1), their names as one comma-separated string and ms[i] for each.
 * Not graph-capturable while enabled. */
int asurf_profile_enable(int on);
int asurf_profile_read(float* ms, int n);
int asurf_profile_read_spans(char* names, size_t names_cap, float* ms, int max_spans);

/* Multi-GPU gradient exchange, device side (extension; alphasurf_amd/dist.py).  The ranks
 * all-reduce their touched-row masks first (max), so that every rank holds the same union mask;
 *   asurf_rows_list   the rows the mask names, in increasing order (the same list on every rank: no atomics), their
 *                     number in count_out[0] on the device; tmp: asurf_rows_list_tmp_bytes(n_rows);
 *   asurf_rows_pack   packed[u] = grad_sh[rows[u]] | grad_density[rows[u]] | grad_surface[rows[u]]  (sh_dim + 2 floats);
 *   asurf_rows_unpack the inverse after the all-reduce of `packed`; sets mask[rows[u]] = 1 (mask may be NULL);
 *   asurf_rows_unpack_add  the same, ADDING the sums to what the buffers hold: the second (third ...) partial batch of a
 *                     step whose batch is exchanged in parts under the render of the next part (dist.PipelinedReduce). */
size_t asurf_rows_list_tmp_bytes(int64_t n_rows);
int asurf_rows_list(const uint8_t* mask, int64_t n_rows, int32_t* rows_out, int32_t* count_out, void* tmp, size_t tmp_bytes,
                    void* stream);
int asurf_rows_pack(const int32_t* rows, int64_t n_list, int32_t sh_dim, const float* grad_sh, const float* grad_density,
                    const float* grad_surface, float* packed, void* stream);
int asurf_rows_unpack(const int32_t* rows, int64_t n_list, int32_t sh_dim, const float* packed, float* grad_sh,
                      float* grad_density, float* grad_surface, uint8_t* mask, void* stream);
int asurf_rows_unpack_add(const int32_t* rows, int64_t n_list, int32_t sh_dim, const float* packed, float* grad_sh,
                          float* grad_density, float* grad_surface, uint8_t* mask, void* stream);

/* Internal switches (no reference counterpart).  Each selects between internal forms that compute the same results
 * -- the parity tests compare the forms with each other: ASURF_DEBUG (8: per-voxel stepping walk, 16: print the
 * pipeline's counters after a call, 32: k_geom without its no-root filter), ASURF_WALK ("step"), ASURF_WALK_SEG,
 * ASURF_RPW, ASURF_CUVOL_RPW, ASURF_NO_PIPELINE, ASURF_NO_REPLAY, ASURF_NO_SPLIT_BWD, ASURF_EVAL ("walk"),
 * ASURF_SWEEP ("block": the block-synchronous k_sweep), ASURF_SWEEP_WPB, ASURF_SWEEP_RPW, ASURF_MASKS ("row": the
 * per-voxel occupancy pass; "split" / "single": the full pass in three steps / as one kernel whatever the grid's size),
 * ASURF_ORDER ("radix": rocPRIM's sort instead of the strip sort), ASURF_SIDE_STREAM (the fused call's fallback
 * kernels on a stream of the library's own), and round 4's: ASURF_BWD_SORT ("0": the backward kernels take the chunks
 * of the record list as they lie instead of sorted by voxel; "wide": two chunks sorted together whatever the list's
 * length), ASURF_BWD_TILE (0 / 4 / 8 / 16 / 32: width of the pixel tile of an image batch's record list, default 8),
 * ASURF_BWD_SH ("tab": the chunk-wide merge through an LDS hash table, measured slower), ASURF_TILES ("1": an image
 * batch rendered in tile order throughout, measured slower), ASURF_LAZY (n: batches of up to n rays over planes of
 * `links` alone, measured slower; default 0), ASURF_OPTIM ("list": the masked optimizer steps as list + step kernels),
 * ASURF_RELAXED (n: batches of up to n rays take k_geom / k_bwd_geo compiled without an occupancy bound; default 196608),
 * ASURF_CUVOL_TILE (0 / 4 / 8 / 16: width of the pixel tile of an image batch in the cuvol kernels; 0: rows),
 * ASURF_SUM_LIST ("0": the default call takes the full occupancy pass every time; "1": the checksum-vouched list of live
 * rows on grids of any size).
 * The environment variables of these names are read ONCE, by the first entry point that runs; afterwards only this
 * function changes a switch (value NULL = default).  Nothing on the call path reads the environment.  The
 * measurement aids that make results invalid (ASURF_DEBUG bits 1, 2, 4, 256 * n) exist in the instrumented build
 * only (`make -C alphasurf_amd/csrc timeline`); the shipped library ignores them.  Returns 0 or ASURF_E_INVALID. */
int asurf_config_set(const char* name, const char* value);

/* Diagnostics: with ASURF_TIMELINE=<kernel id> in the environment (1 k_build_masks, 2 k_sweep, 3 k_geom, 6 k_bwd_sh,
 * 7 k_bwd_geo) every wave of that kernel records {start, end} on the 100 MHz wall clock and {HW_ID, XCC_ID};
 * this copies the first max_recs records (24 bytes each: u64 t0, u64 t1, u32 hw_id, u32 xcc_id; index = wave number
 * in launch order) of the LAST launch to host memory after a device synchronize.  Returns the record count or -1. */
int asurf_debug_read_timeline(void* dst, int max_recs);

#ifdef __cplusplus
}
#endif
#endif /* ASURF_H_ */
