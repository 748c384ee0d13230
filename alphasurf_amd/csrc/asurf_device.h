// asurf_device.h -- device-side types and math of the surf_trav hot path (gfx950).
//
// Product code: compiled only by hipcc into libasurf_hip.so.  Citations are file:line in
// the reference checkout (surf_trav.cu = svox2/csrc/render_lerp_kernel_surf_trav.cu,
// render_util.cuh = svox2/csrc/include/render_util.cuh).
//
// Arithmetic policy (-ffp-contract=off): identical to the reference's CUDA semantics as
// pinned by the CPU oracle -- float/double where the reference uses them, explicit
// fmaf only where the reference writes fmaf()/lerp(), IEEE division in the DDA so that
// the visited-voxel sequence is bit-reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asurf {

constexpr int kWave = 64;
constexpr int kBlock = 256;            // 4 waves; one ray per lane
constexpr int kMaxBasis = 16;
constexpr int kSphPad = kMaxBasis + 1; // LDS row stride of the per-ray SH table
constexpr double kPi = 3.1415926535897931e+0;  // render_util.cuh:7

enum CubicType {  // data_spec.hpp:24-31
    CUBIC_NO_ROOT = 200,
    CUBIC_LINEAR = 201,
    CUBIC_POLY_ONE_R = 202,
    CUBIC_POLY = 203,
    CUBIC_THREE_R = 205,
    CUBIC_ONE_R_ = 206,
};

// Grid description handed to every kernel by value.
struct KGrid {
    const int32_t* __restrict__ links;
    const float* __restrict__ density;
    const float* __restrict__ surface;
    const float* __restrict__ sh;
    const float* __restrict__ level_set;
    int size[3];
    int offx, offy;  // element strides of links along x and y
    int level_set_num;
    float offset[3], scaling[3];
    float fake_sample_std;
    float trunc_a;
    // occupancy pyramid built per call (asurf_masks.h)
    const unsigned long long* __restrict__ fine;       // plane the walk tests (work or live)
    const unsigned long long* __restrict__ fine_live;  // links + density gate only (counters)
    const unsigned int* __restrict__ coarse;           // one bit per 8^3 brick, for `fine`
    // `fine` again with the brick's 8 words cut along y resp. z instead of x (asurf_sweep.h): word = the
    // voxel's index along axis a, bit = 8 * (index along axis a+1) + (index along axis a+2), axes mod 3
    const unsigned long long* __restrict__ fine_y;
    const unsigned long long* __restrict__ fine_z;
    int nb[3];   // bricks per axis
    int nwz;     // 32-bit coarse words per (bx,by) row
    int coarse_words;
};

struct KOpt {
    float background_brightness, step_size, sigma_thresh, stop_thresh, near_clip;
    int use_spheric_clip, surf_fake_sample;
    float surf_fake_sample_min_vox_len;
    int limited_fake_sample, no_surf_grad_from_sh, alpha_activation_type, fake_sample_l_dist,
        fake_sample_normalize_surf, only_outward_intersect, truncated_vol_render;
    float trunc_vol_weight_min;
    // The ASURF_DEBUG switch (asurf_config_set, include/asurf.h); 0 in production.  8: per-voxel walk without brick
    // jumps, k_walk instead of k_sweep (the tests' reference); 16: the pipeline's counters are printed after the call;
    // 32: k_geom without its no-root filter (variants/geom_filter_check.py) -- all three leave the results what they
    // are.  The measurement aids that do not -- 1, 2, 4: traversal only / no level sets / no candidates in the
    // one-kernel path; 256 * n: k_sweep leaves after phase n - 1 -- are compiled into the instrumented build only
    // (`make timeline`: -DASURF_INSTRUMENT); the host masks their bits off otherwise.
    int debug;
    // 1: the occupancy planes say "all 8 corner links exist" only (small batches, asurf_masks2.h): whoever reads a flagged
    // voxel applies the value gates of surf_trav.cu:230-239 itself (k_sweep_wave; the one-kernel paths)
    int lazy_gates;
};
#ifdef ASURF_INSTRUMENT
#define ASURF_AID(o, bit) (((o).debug & (bit)) != 0)
#else
#define ASURF_AID(o, bit) false
#endif

struct KGrads {
    float* grad_density;
    float* grad_sh;
    float* grad_surface;
    float* grad_fake_sample_std;
    uint8_t* mask;
    uint8_t* mask2;   // asurf_grads::mask_sh_out: the same marks once more
};

struct KLoss {  // already scaled by the launcher (surf_trav.cu:3896-3915)
    float sparsity_loss;
    float lambda_l2, lambda_l1;
    float lambda_l_dist, lambda_l_entropy;
    int no_norm_weight_l_entropy;
    float lambda_l_dist_a, lambda_l_entropy_a, lambda_l_samp_dist, lambda_l_di, l_di_alpha_thresh;
    float surf_sparse_alpha_thresh, lambda_inplace_surf_sparse, lambda_inwards_norm_loss;
    float lambda_conv_mode_samp;
    float norm_l2, norm_l1;  // 2/(3Q), 1/(3Q) with the (global) ray count
};

// Wave timeline (diagnostics; compiled in by `make -C alphasurf_amd/csrc timeline` only -- the two timestamps a wave carries cost
// k_geom 64 bytes of scratch per lane and 0.06 ms).  With ASURF_TIMELINE=<kernel id> the waves of ONE kernel leave
// their start and end on the 100 MHz wall clock and where they ran; asurf_debug_read_timeline() copies the table out
// (variants/timeline.py).  ids: 1 k_build_masks, 2 k_sweep, 3 k_geom, 6 k_bwd_sh, 7 k_bwd_geo.
constexpr int kTimelineCap = 1 << 18;
struct TimelineRec {
    unsigned long long t0, t1;
    unsigned hw_id, xcc_id;
};
#ifdef ASURF_TIMELINE
__device__ int g_tl_kernel = 0;
__device__ TimelineRec g_timeline[kTimelineCap];
struct TlScope {
    unsigned long long t0;
    bool on;
    __device__ __forceinline__ explicit TlScope(int kernel_id) {
        on = (g_tl_kernel == kernel_id);
        t0 = on ? wall_clock64() : 0ull;
    }
    __device__ __forceinline__ void end() const {
        if (on && (threadIdx.x & 63) == 0) {
            const unsigned w = blockIdx.x * ((blockDim.x + 63u) >> 6) + (threadIdx.x >> 6);
            if (w < (unsigned)kTimelineCap) {
                TimelineRec rec;
                rec.t0 = t0;
                rec.t1 = wall_clock64();
                rec.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
                rec.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
                g_timeline[w] = rec;
            }
        }
    }
};
#else
struct TlScope {
    __device__ __forceinline__ explicit TlScope(int) {}
    __device__ __forceinline__ void end() const {}
};
#endif

__device__ __forceinline__ float lerpf(float a, float b, float w) { return fmaf(w, b - a, a); }
__device__ __forceinline__ float sqrf(float x) { return x * x; }
__device__ __forceinline__ double sqrd(double x) { return x * x; }
__device__ __forceinline__ double cubd(double x) { return x * x * x; }

__device__ __forceinline__ int bcast_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float bcast_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// Sum over the 64 lanes, returned to every lane.  DPP row shifts + the two row broadcasts leave
// the total in lane 63 (six VALU adds, no LDS crossbar as __shfl_xor's ds_bpermute would use); a
// lane whose DPP source is outside its row, or whose row is masked off, adds 0.  All 64 lanes must
// be active.
__device__ __forceinline__ float wave_sum(float v) {
#define ASURF_DPP_ADD(ctrl, row_mask)                                                                  \
    v += __builtin_bit_cast(                                                                           \
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false))
    ASURF_DPP_ADD(0x111, 0xf);  // row_shr:1
    ASURF_DPP_ADD(0x112, 0xf);  // row_shr:2
    ASURF_DPP_ADD(0x114, 0xf);  // row_shr:4
    ASURF_DPP_ADD(0x118, 0xf);  // row_shr:8  -> lane 15 of every row holds the row's sum
    ASURF_DPP_ADD(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
    ASURF_DPP_ADD(0x143, 0xc);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
#undef ASURF_DPP_ADD
    return bcast_f(v, 63);
}

// SH basis at the world-space view direction: render_util.cuh:405-436 (1/4/9 terms);
// the 16-term band is the block the reference keeps commented out (:414-423), equal to
// svox2/utils.py:147-154.
template <int B>
__device__ __forceinline__ void eval_sh(const float* dir, float* out) {
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    out[0] = C0;
    if constexpr (B >= 4) {
        const float x = dir[0], y = dir[1], z = dir[2];
        out[1] = -C1 * y;
        out[2] = C1 * z;
        out[3] = -C1 * x;
        if constexpr (B >= 9) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
            out[4] = 1.0925484305920792f * xy;
            out[5] = -1.0925484305920792f * yz;
            out[6] = (float)((double)0.31539156525252005f * (2.0 * (double)zz - (double)xx - (double)yy));
            out[7] = -1.0925484305920792f * xz;
            out[8] = 0.5462742152960396f * (xx - yy);
            if constexpr (B >= 16) {
                out[9] = -0.5900435899266435f * y * (3 * xx - yy);
                out[10] = 2.890611442640554f * xy * z;
                out[11] = -0.4570457994644658f * y * (4 * zz - xx - yy);
                out[12] = 0.3731763325901154f * z * (2 * zz - 3 * xx - 3 * yy);
                out[13] = -0.4570457994644658f * x * (4 * zz - xx - yy);
                out[14] = 1.445305721320277f * z * (xx - yy);
                out[15] = -0.5900435899266435f * x * (xx - 3 * yy);
            }
        }
    }
}

struct Ray {
    float o[3];
    float d[3];
    float tmin, tmax;
};

// Kernel prologue: world -> grid coordinates and the ray/grid slab test.
// surf_trav.cu:3163-3171; render_util.cuh:536-547, 651-693; cuda_util.cuh:63-69.
__device__ __forceinline__ void setup_ray(const KGrid& g, const KOpt& o, Ray& r) {
#pragma unroll
    for (int k = 0; k < 3; ++k) r.o[k] = fmaf(r.o[k], g.scaling[k], g.offset[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) r.d[k] *= g.scaling[k];
    const float rn = 1.f / sqrtf(r.d[0] * r.d[0] + r.d[1] * r.d[1] + r.d[2] * r.d[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) r.d[k] *= rn;
    const float world_step = rn * o.step_size;
    if (o.use_spheric_clip) {  // render_util.cuh:661-680
        float so[3], sd[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float ss = 2.f / float(g.size[i]);
            so[i] = fmaf(r.o[i] + 0.5f, ss, -1.f);
            sd[i] = r.d[i] * ss;
        }
        const float q2a = 2 * (sd[0] * sd[0] + sd[1] * sd[1] + sd[2] * sd[2]);
        const float qb = 2 * (so[0] * sd[0] + so[1] * sd[1] + so[2] * sd[2]);
        const float f = qb * qb - 2 * q2a * (so[0] * so[0] + so[1] * so[1] + so[2] * so[2]);
        const float r_in = 1.f - o.near_clip;
        const float det_out = f + 2 * q2a * 1.f * 1.f;
        const float det_in = f + 2 * q2a * r_in * r_in;
        bool ok = true;
        if (det_out < 0) ok = false;
        else r.tmax = (-qb + sqrtf(det_out)) / q2a;
        if (ok) {
            if (det_in < 0) ok = false;
            else r.tmin = (-qb - sqrtf(det_in)) / q2a;
        }
        if (!ok) {
            r.tmin = 1e-9f;
            r.tmax = 0.f;
        }
    } else {  // render_util.cuh:681-692
        r.tmin = o.near_clip / world_step * o.step_size;
        r.tmax = 2e3f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float invdir = 1.0f / r.d[i];
            const float t1 = (-0.5f - r.o[i]) * invdir;
            const float t2 = ((float)g.size[i] - 0.5f - r.o[i]) * invdir;
            if (r.d[i] != 0.f) {
                r.tmin = fmaxf(r.tmin, fminf(t1, t2));
                r.tmax = fminf(r.tmax, fmaxf(t1, t2));
            }
        }
    }
}

// nested-lerp trilinear interpolation, corner order 000,001,010,011,100,... (z fastest):
// render_util.cuh:72-92
__device__ __forceinline__ float trilerp8(const float* v, const float* pos) {
    const float ix0y0 = lerpf(v[0], v[1], pos[2]);
    const float ix0y1 = lerpf(v[2], v[3], pos[2]);
    const float ix0 = lerpf(ix0y0, ix0y1, pos[1]);
    const float ix1y0 = lerpf(v[4], v[5], pos[2]);
    const float ix1y1 = lerpf(v[6], v[7], pos[2]);
    const float ix1 = lerpf(ix1y0, ix1y1, pos[1]);
    return lerpf(ix0, ix1, pos[0]);
}

// scatter weights of the backward trilerp, render_util.cuh:103-120
__device__ __forceinline__ void corner_weights(const float* pos, float g, float* w) {
    const float ay = 1.f - pos[1], az = 1.f - pos[2];
    float xo = (1.0f - pos[0]) * g;
    w[0] = ay * az * xo;
    w[1] = ay * pos[2] * xo;
    w[2] = pos[1] * az * xo;
    w[3] = pos[1] * pos[2] * xo;
    xo = pos[0] * g;
    w[4] = ay * az * xo;
    w[5] = ay * pos[2] * xo;
    w[6] = pos[1] * az * xo;
    w[7] = pos[1] * pos[2] * xo;
}

// d(trilerp)/d(pos) * grad_in accumulated into out, render_util.cuh:156-204
__device__ __forceinline__ void trilerp_dpos(const float* v, const float* pos, float grad_in,
                                             float* out) {
    const float ix0y0 = lerpf(v[0], v[1], pos[2]);
    const float ix0y1 = lerpf(v[2], v[3], pos[2]);
    const float ix0 = lerpf(ix0y0, ix0y1, pos[1]);
    const float ix1y0 = lerpf(v[4], v[5], pos[2]);
    const float ix1y1 = lerpf(v[6], v[7], pos[2]);
    const float ix1 = lerpf(ix1y0, ix1y1, pos[1]);
    out[0] += grad_in * (ix1 - ix0);
    out[1] += grad_in * ((1 - pos[0]) * (ix0y1 - ix0y0) + (pos[0]) * (ix1y1 - ix1y0));
    out[2] += grad_in * ((1 - pos[0]) * ((1 - pos[1]) * (v[1] - v[0]) + (pos[1]) * (v[3] - v[2])) +
                         (pos[0]) * ((1 - pos[1]) * (v[5] - v[4]) + (pos[1]) * (v[7] - v[6])));
}

// gradient of the trilinear surface field, render_util.cuh:2190-2236
__device__ __forceinline__ void field_grad(const float* v, const float* pos, float* out) {
    const float ix0y0 = lerpf(v[0], v[1], pos[2]);
    const float ix0y1 = lerpf(v[2], v[3], pos[2]);
    const float ix0 = lerpf(ix0y0, ix0y1, pos[1]);
    const float ix1y0 = lerpf(v[4], v[5], pos[2]);
    const float ix1y1 = lerpf(v[6], v[7], pos[2]);
    const float ix1 = lerpf(ix1y0, ix1y1, pos[1]);
    out[0] = ix1 - ix0;
    out[1] = (float)((double)pos[0] * (double)(-ix1y0 + ix1y1) +
                     (1. - (double)pos[0]) * (double)(-ix0y0 + ix0y1));
    out[2] = pos[0] * (pos[1] * (-v[6] + v[7]) + (1 - pos[1]) * (-v[4] + v[5])) +
             (1 - pos[0]) * (pos[1] * (-v[2] + v[3]) + (1 - pos[1]) * (-v[0] + v[1]));
}

// Trilinear field along the ray as a cubic in t (render_util.cuh:789-848), multiplied
// out axis by axis: z, then y, then x.
__device__ __forceinline__ void surface_cubic(const float* sf, const double* o, const float* df,
                                              double* f) {
    const double d0 = df[0], d1 = df[1], d2 = df[2];
    double a[4], b[4];
#pragma unroll
    for (int ij = 0; ij < 4; ++ij) {
        const double s0 = sf[2 * ij], s1 = sf[2 * ij + 1];
        a[ij] = s0 * (1 - o[2]) + s1 * o[2];
        b[ij] = d2 * (s1 - s0);
    }
    double c[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double a0 = a[2 * i], a1 = a[2 * i + 1], b0 = b[2 * i], b1 = b[2 * i + 1];
        c[i][0] = a0 * (1 - o[1]) + a1 * o[1];
        c[i][1] = (b0 * (1 - o[1]) + b1 * o[1]) + d1 * (a1 - a0);
        c[i][2] = d1 * (b1 - b0);
    }
    f[0] = c[0][0] * (1 - o[0]) + c[1][0] * o[0];
    f[1] = (c[0][1] * (1 - o[0]) + c[1][1] * o[0]) + d0 * (c[1][0] - c[0][0]);
    f[2] = (c[0][2] * (1 - o[0]) + c[1][2] * o[0]) + d0 * (c[1][1] - c[0][1]);
    f[3] = d0 * (c[1][2] - c[0][2]);
}

// d f_k / d s_ijk in float (render_util.cuh:850-934)
__device__ __forceinline__ void surface_cubic_grad(const float* o, const float* d, const float* gf,
                                                   float* gs) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int ix = (c >> 2) & 1, iy = (c >> 1) & 1, iz = c & 1;
        const float ax = ix ? o[0] : 1 - o[0], bx = ix ? d[0] : -d[0];
        const float ay = iy ? o[1] : 1 - o[1], by = iy ? d[1] : -d[1];
        const float az = iz ? o[2] : 1 - o[2], bz = iz ? d[2] : -d[2];
        const float c0 = ax * ay * az;
        const float c1 = bx * ay * az + ax * (by * az + ay * bz);
        const float c2 = bx * (by * az + ay * bz) + ax * by * bz;
        const float c3 = bx * by * bz;
        gs[c] = gf[0] * c0 + gf[1] * c1 + gf[2] * c2 + gf[3] * c3;
    }
}

// cubic_equation_solver_vieta, render_util.cuh:1126-1203
__device__ inline int solve_cubic_vieta(double f0, double f1, double f2, double f3, double* st) {
    const double eps = 1e-10;
    if (fabs(f3) < eps) {
        if (fabs(f2) < eps) {
            if (fabs(f1) < eps) return CUBIC_NO_ROOT;
            st[0] = -f0 / f1;
            return CUBIC_LINEAR;
        }
        const double D = sqrd(f1) - 4.0 * f2 * f0;
        const double sD = sqrt(D);
        if (D > 0) {
            if (f2 > 0) {
                st[0] = (-f1 - sD) / (2 * f2);
                st[1] = (-f1 + sD) / (2 * f2);
            } else {
                st[0] = (-f1 + sD) / (2 * f2);
                st[1] = (-f1 - sD) / (2 * f2);
            }
            if (fabs(st[0] - st[1]) < eps) {
                st[1] = -1;
                return CUBIC_POLY_ONE_R;
            }
            return CUBIC_POLY;
        }
        return CUBIC_NO_ROOT;
    }
    const double b = f2 / f3, c = f1 / f3, d = f0 / f3;
    const double Q = (sqrd(b) - 3. * c) / 9.;
    const double R = (2. * cubd(b) - 9. * b * c + 27. * d) / 54.;
    if (sqrd(R) < cubd(Q)) {
        const double theta = acos(R / sqrt(cubd(Q)));
        const double m2sq = -2. * sqrt(Q);
        // cos((theta -+ 2 pi) / 3) from one sincos: cos(x -+ 2pi/3) = -cos(x)/2 +- sqrt(3)/2 sin(x)
        double s3, c3;
        sincos(theta / 3., &s3, &c3);
        const double h = 0.86602540378443864676 * s3;
        st[0] = m2sq * c3 - b / 3.;
        st[1] = m2sq * (-0.5 * c3 + h) - b / 3.;
        st[2] = m2sq * (-0.5 * c3 - h) - b / 3.;
        return CUBIC_THREE_R;
    }
    const double sgn = (R > 0.) ? 1. : -1.;  // cuda_util.cuh:145
    const double A = -sgn * cbrt(fabs(R) + sqrt(sqrd(R) - cubd(Q)));
    const double Bq = (A == 0.) ? 0. : Q / A;
    st[0] = (A + Bq) - b / 3.;
    return CUBIC_ONE_R_;
}

// The root type solve_cubic_vieta would report, without evaluating the cubic's roots (the replay
// backward knows the sample position already and only needs the branch of the gradient).
__device__ inline int classify_cubic_vieta(double f0, double f1, double f2, double f3) {
    const double eps = 1e-10;
    if (fabs(f3) < eps) {
        if (fabs(f2) < eps) return (fabs(f1) < eps) ? CUBIC_NO_ROOT : CUBIC_LINEAR;
        const double D = sqrd(f1) - 4.0 * f2 * f0;
        if (!(D > 0)) return CUBIC_NO_ROOT;
        const double sD = sqrt(D);
        const double r0 = (-f1 - sD) / (2 * f2), r1 = (-f1 + sD) / (2 * f2);
        return (fabs(r0 - r1) < eps) ? CUBIC_POLY_ONE_R : CUBIC_POLY;
    }
    const double b = f2 / f3, c = f1 / f3, d = f0 / f3;
    const double Q = (sqrd(b) - 3. * c) / 9.;
    const double R = (2. * cubd(b) - 9. * b * c + 27. * d) / 54.;
    return (sqrd(R) < cubd(Q)) ? CUBIC_THREE_R : CUBIC_ONE_R_;
}

// calc_cubic_root_grad_vieta, render_util.cuh:1206-1415: g[k] holds dL/d(root) on entry
// and dL/d f_k on return.
__device__ inline void cubic_root_grad_vieta(int type, int st_id, const double* fs, float* g) {
    const double f0 = fs[0], f1 = fs[1], f2 = fs[2], f3 = fs[3];
    if (type == CUBIC_LINEAR) {
        g[0] *= (float)(-1. / f1);
        g[1] *= (float)(f0 / sqrd(f1));
        g[2] = 0.f;
        g[3] = 0.f;
        return;
    }
    if (type == CUBIC_POLY_ONE_R) {
        const double D = sqrd(f1) - 4. * f2 * f0;
        const double sD = sqrt(D);
        const double dt_dD = 1 / (4. * f2 * sD);
        g[0] *= (float)(-1 / sD);
        g[1] *= (float)((-1) / (2 * f2) + dt_dD * 2 * f1);
        g[2] *= (float)((f1 - sD) / (4 * sqrd(f2)) + dt_dD * (-4) * f0);
        g[3] = 0.f;
        return;
    }
    if (type == CUBIC_POLY) {
        const double D = sqrd(f1) - 4.0 * f2 * f0;
        const double sD = sqrt(D);
        const double sq2 = sqrd(f2);
        if (st_id == 0) {
            const double dt_dD = -1 / (4 * f2 * sD);
            g[0] *= (float)(1 / sD);
            g[1] *= (float)((-1) / (2 * f2) + dt_dD * 2 * f1);
            g[2] *= (float)((f1 + sD) / (2 * sq2) + dt_dD * (-4) * f0);
        } else {
            const double dt_dD = 1 / (4 * f2 * sD);
            g[0] *= (float)(-1 / sD);
            g[1] *= (float)((-1) / (2 * f2) + dt_dD * 2 * f1);
            g[2] *= (float)((f1 - sD) / (2 * sq2) + dt_dD * (-4) * f0);
        }
        g[3] = 0.f;
        return;
    }
    if (type != CUBIC_THREE_R && type != CUBIC_ONE_R_) return;

    const double b = f2 / f3, c = f1 / f3, d = f0 / f3;
    const double Q = (sqrd(b) - 3. * c) / 9.;
    const double R = (2. * cubd(b) - 9. * b * c + 27. * d) / 54.;
    const double dQ[4] = {0., -1. / (3. * f3), 2. * f2 / (9. * sqrd(f3)),
                          f1 / (3. * sqrd(f3)) - 2. * sqrd(f2) / (9 * cubd(f3))};
    const double dR[4] = {1. / (2. * f3), -f2 / (6. * sqrd(f3)),
                          -f1 / (6. * sqrd(f3)) + sqrd(f2) / (9 * cubd(f3)),
                          -f0 / (2. * sqrd(f3)) + f1 * f2 / (3. * cubd(f3)) -
                              cubd(f2) / (9 * (f3 * f3 * f3 * f3))};
    const double db[4] = {0., 0., 1. / f3, -f2 / sqrd(f3)};
    const double dst_db = -1. / 3.;
    if (type == CUBIC_THREE_R) {
        const double sqQ3 = sqrt(cubd(Q));
        const double theta = acos(R / sqQ3);
        const double rad = sqrt(1. - sqrd(R) / cubd(Q));
        const double dth_dQ = 3. * R / (2. * Q * rad * sqQ3);
        const double dth_dR = -1 / (rad * sqQ3);
        // one sincos of theta/3; the shifted angles through cos/sin(pi/3), cos/sin(pi/6)
        double s3, c3;
        sincos(theta / 3., &s3, &c3);
        const double kH = 0.86602540378443864676;  // sqrt(3)/2
        const double sqQ = sqrt(Q);
        double cs, sn;  // the (cos-like, sin-like) pair each root's derivative uses
        double dst_dQ, dst_dth;
        if (st_id == 0) {
            dst_dQ = -c3 / sqQ;
            dst_dth = 2. * sqQ * s3 / 3.;
        } else if (st_id == 1) {
            cs = 0.5 * c3 - kH * s3;  // cos(theta/3 + pi/3)
            sn = 0.5 * s3 + kH * c3;  // sin(theta/3 + pi/3)
            dst_dQ = cs / sqQ;
            dst_dth = -2. * sqQ * sn / 3.;
        } else {
            sn = kH * s3 + 0.5 * c3;  // sin(theta/3 + pi/6)
            cs = kH * c3 - 0.5 * s3;  // cos(theta/3 + pi/6)
            dst_dQ = sn / sqQ;
            dst_dth = 2. * sqQ * cs / 3.;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            g[k] *= (float)(dst_dth * (dth_dQ * dQ[k] + dth_dR * dR[k]) + dst_dQ * dQ[k] +
                            dst_db * db[k]);
        return;
    }
    const double sgn = (R > 0.) ? 1. : -1.;
    const double A = -sgn * cbrt(fabs(R) + sqrt(sqrd(R) - cubd(Q)));
    const double sq = fmax(sqrt(-cubd(Q) + sqrd(R)), 1e-10);
    const double dA_dR = (R >= 0.) ? (-(R / (3. * sq) + 1. / 3.) / fmax(cbrt(sqrd(R + sq)), 1e-10))
                                   : ((R / (3. * sq) - 1. / 3.) / fmax(cbrt(sqrd(-R + sq)), 1e-10));
    const double dA_dQ = (R >= 0.) ? (sqrd(Q) / (2. * sq * cbrt(sqrd(R + sq))))
                                   : (-sqrd(Q) / (2. * sq * cbrt(sqrd(-R + sq))));
    const double dB_dA = (A == 0.) ? 0. : -Q / sqrd(A);
    const double dB_dQ = (A == 0.) ? 0. : 1. / A;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        g[k] *= (float)((dB_dA + 1.) * (dA_dQ * dQ[k] + dA_dR * dR[k]) + dB_dQ * dQ[k] +
                        dst_db * db[k]);
}

// activations and the truncation window, render_util.cuh:2138-2188
__device__ __forceinline__ float alpha_act(float raw, int type) {
    if (type == 0) return (float)(1. / (1. + (double)expf(-raw)));
    return (raw >= 0.f) ? 1 - expf(-raw) : 0.f;
}
__device__ __forceinline__ float alpha_act_grad(float alpha, int type) {
    if (type == 0) return alpha * (1 - alpha);
    return (alpha > 0.f) ? 1 - alpha : 0.f;
}
__device__ __forceinline__ float trunc_rw(float x, float a, float clamp_min) {
    const float u = fminf(fmaxf(a - x, 0.f), 1.f);
    const float cs = cosf((float)(kPi * (double)u));
    return fmaxf(.5f * (1.f - cs), clamp_min);
}

// _LOG(max(x, 1e-8) / sum), surf_trav.cu:1769-1770
__device__ __forceinline__ float log_ratio(float x, float sum) {
    return logf((float)(fmax((double)x, 1e-8) / (double)sum));
}
__device__ __forceinline__ float clamp_lo(float x) { return (float)fmax((double)x, 1e-8); }

}  // namespace asurf
