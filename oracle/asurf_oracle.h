/*
 * asurf_oracle.h -- CPU ORACLE (test infrastructure, not product code).
 *
 * Scalar, one-ray-at-a-time restatement of the reference's CUDA semantics for the
 * surf_trav hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (alphasurf_amd/, libasurf_hip.so)
 * never links, imports or falls back to it.
 *
 * Parity pinning: checked against golden vectors produced by importing the
 * reference's own pure-PyTorch renderer (svox2/svox2.py:1596-2857) in the build
 * container -- tests/golden/gen_golden.py, tests/test_oracle_golden.py.
 *
 * Struct layouts intentionally match include/asurf.h field for field so a test can
 * describe a scene once; here every pointer is a HOST pointer.
 */
#ifndef ASURF_ORACLE_H_
#define ASURF_ORACLE_H_
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_grid {
    const int32_t* links;
    const float* density_data;
    const float* surface_data;
    const float* sh_data;
    const float* level_set_data;
    int64_t capacity;
    int32_t size[3];
    int32_t level_set_num;
    int32_t basis_dim;
    int32_t basis_type;
    int32_t surface_type;
    float offset[3];
    float scaling[3];
    float fake_sample_std;
    float truncated_vol_render_a;
    int32_t background_nlayers;
} oracle_grid;

typedef struct oracle_rays {
    const float* origins;
    const float* dirs;
    const uint8_t* masks;
    int64_t n_rays;
} oracle_rays;

typedef struct oracle_ray_vox {
    const int32_t* voxel_ls;
    const int32_t* vox_start_i;
    const int32_t* vox_num;
    int64_t n_vox;
} oracle_ray_vox;

typedef struct oracle_render_options {
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
} oracle_render_options;

typedef struct oracle_grads {
    float* grad_density_out;
    float* grad_sh_out;
    float* grad_surface_out;
    float* grad_fake_sample_std_out;
    uint8_t* mask_out;
} oracle_grads;

typedef struct oracle_fused_params {
    const float* rgb_gt;
    float beta_loss;
    float sparsity_loss;
    float fused_surf_norm_reg_scale;
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
    int64_t global_n_rays;
} oracle_fused_params;

/* Optional per-ray statistics (any pointer may be NULL).  n_traversed / n_active /
 * n_samples are the V_t / V_a / S counters of SURVEY.md section 8(d). */
typedef struct oracle_stats {
    int32_t* n_traversed; /* [n_rays] */
    int32_t* n_active;    /* [n_rays] */
    int32_t* n_samples;   /* [n_rays] */
    /* optional per-sample dump of the forward pass, max_dump entries per ray */
    int32_t max_dump;
    float* dump_t;        /* [n_rays,max_dump] t_close+st (real) or voxel mid t (fake) */
    float* dump_alpha;    /* [n_rays,max_dump] re-weighted alpha */
    float* dump_weight;   /* [n_rays,max_dump] */
    int32_t* dump_voxel;  /* [n_rays,max_dump,3] */
    int32_t* dump_flags;  /* [n_rays,max_dump] bit0 = fake sample */
    int32_t* n_surf;      /* [n_rays] active voxels that hold a level set (cubic solved) */
} oracle_stats;

int oracle_num_threads(void);
void oracle_set_num_threads(int n);

/* surf_trav.cu:3596-3654 */
int oracle_volume_render_surf_trav(const oracle_grid* grid, const oracle_rays* rays,
                                   const oracle_render_options* opt, float* rgb_out,
                                   const oracle_stats* stats);
/* surf_trav.cu:3708-3800 */
int oracle_volume_render_surf_trav_backward(const oracle_grid* grid, const oracle_rays* rays,
                                            const oracle_render_options* opt,
                                            const float* grad_out, const float* color_cache,
                                            const oracle_grads* grads);
/* surf_trav.cu:3802-3942 */
int oracle_volume_render_surf_trav_fused(const oracle_grid* grid, const oracle_rays* rays,
                                         const oracle_render_options* opt,
                                         const oracle_fused_params* fused, float* rgb_out,
                                         const oracle_grads* grads, const oracle_stats* stats);

/* render_lerp_kernel_surface.cu:1198-1439 (legacy voxel-list backend) */
int oracle_volume_render_surface(const oracle_grid* grid, const oracle_rays* rays,
                                 const oracle_ray_vox* ray_vox,
                                 const oracle_render_options* opt, float* rgb_out);
int oracle_volume_render_surface_backward(const oracle_grid* grid, const oracle_rays* rays,
                                          const oracle_ray_vox* ray_vox,
                                          const oracle_render_options* opt,
                                          const float* grad_out, const float* color_cache,
                                          const oracle_grads* grads);
int oracle_volume_render_surface_fused(const oracle_grid* grid, const oracle_rays* rays,
                                       const oracle_ray_vox* ray_vox,
                                       const oracle_render_options* opt, const float* rgb_gt,
                                       float beta_loss, float sparsity_loss, float* rgb_out,
                                       const oracle_grads* grads);

/* Evaluation renders, surf_trav.cu:564-1705 / 3944-4081: mode 0 expected depth, 1 mode depth
 * (param = weight_thresh), 2 first-crossing depth (param = alpha threshold), 3 first-crossing
 * alpha, 4 normal (out0[Q,3]), 5 extract_pts (param = alpha_thresh; out0/out1[Q,max_sample],
 * zeroed by the caller). */
int oracle_render_eval(const oracle_grid* grid, const oracle_rays* rays, const oracle_render_options* opt,
                       int32_t mode, float param, int32_t max_sample, float* out0, float* out1);

/* diagnostic: evaluate the legacy d f_k / d s_corner expression (render_util.cuh:1705-1775)
 * in float64 instead of the reference's float32 */
void oracle_set_legacy_f64_surface_grad(int on);

/* test_cuda.cu:32-122 */
int oracle_test_cubic_root_grad(const double* fs, const int32_t* st_ids, float* grad_fs,
                                float* st, int64_t n, int32_t use_vieta);

/* Voxel lists for the legacy backend, produced by the surf_trav DDA
 * (surf_trav.cu:75-197): returns the total count; fills voxel_ls[cap,3],
 * vox_start_i[n_rays], vox_num[n_rays] (voxel_ls may be NULL to count only). */
int64_t oracle_trace_voxel_lists(const oracle_grid* grid, const oracle_rays* rays,
                                 const oracle_render_options* opt, int32_t* voxel_ls,
                                 int64_t cap, int32_t* vox_start_i, int32_t* vox_num);

/* The rays as the kernels' prologue leaves them (surf_trav.cu:3163-3171): out[8 q .. 8 q + 7] = origin[3] and
 * unit direction[3] in grid coordinates, tmin, tmax. */
int oracle_setup_rays(const oracle_grid* grid, const oracle_rays* rays, const oracle_render_options* opt,
                      float* out);

/* SH basis at unit directions (render_util.cuh:405-436; svox2/utils.py:119-166) */
void oracle_eval_sh(int32_t basis_dim, const float* dirs, int64_t n, float* out);

/* render_lerp_kernel_cuvol.cu:1120-1354 (the Plenoxels `cuvol` renderer; surface_data may be NULL) */
int oracle_volume_render_cuvol(const oracle_grid* grid, const oracle_rays* rays,
                               const oracle_render_options* opt, float* rgb_out);
int oracle_volume_render_cuvol_backward(const oracle_grid* grid, const oracle_rays* rays,
                                        const oracle_render_options* opt, const float* grad_out,
                                        const float* color_cache, const oracle_grads* grads);
int oracle_volume_render_cuvol_fused(const oracle_grid* grid, const oracle_rays* rays,
                                     const oracle_render_options* opt, const float* rgb_gt,
                                     float beta_loss, float sparsity_loss, float* rgb_out,
                                     const oracle_grads* grads);

/* svox2_kernel.cu:248-372, 542-583 (cubic_extract_iso_pts): iso-surface points of the listed cells along
 * axis-parallel lines; out [n_cells, 3 n_sample^2, 3] zeroed by the caller.  PARITY UNPINNED as a whole (the
 * reference's kernel needs CUDA); its cubic, root solver and trilerp are the pinned ones of the renderer. */
int oracle_cubic_extract_iso_pts(const int32_t* links, const int32_t* size, const float* level_data,
                                 const float* mask_data, const int32_t* cell_ids, int64_t n_cells,
                                 int32_t n_sample, float density_thresh, float* out);

/* misc_kernel.cu:186-400, 888-937: per-vertex max of the compositing weight (links == NULL, data = dense
 * [X,Y,Z] volume) or of the transmittance in front of a sample (sparse grid) over a camera's pixel rays */
int oracle_weight_render(const float* data, const int32_t* links, const int32_t* size, const float* c2w,
                         float fx, float fy, float cx, float cy, int32_t width, int32_t height,
                         float step_size, float stop_thresh, int32_t last_sample_opaque,
                         const float* offset, const float* scaling, float* out);

/* render_lerp_kernel_cuvol.cu:127-370, 1356-1442: mode 0 expected term, 1 mode term (param = weight_thresh),
 * 2 med term (out0 depths / out1 sigmas [Q, max_sample]), 3 first sample above param */
int oracle_render_eval_cuvol(const oracle_grid* grid, const oracle_rays* rays,
                             const oracle_render_options* opt, int32_t mode, float param,
                             int32_t max_sample, float* out0, float* out1);

#ifdef __cplusplus
}
#endif
#endif
