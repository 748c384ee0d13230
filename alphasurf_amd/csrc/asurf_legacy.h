// Legacy `surface` backend (render_lerp_kernel_surface.cu): the caller supplies each ray's
// voxel list (RayVoxIntersecSpec) instead of the DDA; cubic coefficients are taken about the
// ray ORIGIN (global t), the solver is the f/g/h form, alpha is always a sigmoid, one level
// set, no sigma gate / window / fake samples / auxiliary losses.
//
// Same execution model as the surf_trav one-kernel path: one ray per lane walks its list and
// finds its next sample; the wave then shades every pending sample cooperatively
// (lanes = corner x SH coefficient, asurf_kernels.h) and the lane composites it.
#pragma once
#include "asurf_kernels.h"

namespace asurf {

constexpr int CUBIC_ONE_R = 204;  // data_spec.hpp:24-31 (triple root)

// render_util.cuh:753-787
__device__ __forceinline__ void surface_cubic_global(const double* s, const double* o, const double* d,
                                                     const int* l, double* f) {
    double lo[3], hi[3];  // trilinear factors at t = 0: corner bit clear / set
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        lo[k] = 1 - o[k] + l[k];
        hi[k] = o[k] - l[k];
    }
    double c[2], dd[2], e[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const double* sx = s + 4 * x;
        const double a0 = sx[0] * lo[2] + sx[1] * hi[2];
        const double a1 = sx[2] * lo[2] + sx[3] * hi[2];
        const double b0 = -sx[0] + sx[1];
        const double b1 = -sx[2] + sx[3];
        c[x] = a0 * lo[1] + a1 * hi[1];
        dd[x] = -(a0 * d[1] - d[2] * b0 * lo[1]) + (a1 * d[1] + d[2] * b1 * hi[1]);
        e[x] = -d[1] * d[2] * b0 + d[1] * d[2] * b1;
    }
    f[3] = -e[0] * d[0] + e[1] * d[0];
    f[2] = -dd[0] * d[0] + e[0] * lo[0] + dd[1] * d[0] + e[1] * hi[0];
    f[1] = -c[0] * d[0] + dd[0] * lo[0] + c[1] * d[0] + dd[1] * hi[0];
    f[0] = c[0] * lo[0] + c[1] * hi[0];
}

__device__ __forceinline__ double cond_cbrt(double x) { return (x >= 0) ? cbrt(x) : -1. * cbrt(-x); }
__device__ __forceinline__ double d_cond_cbrt(double x) {  // util.hpp:36
    return (x >= 0) ? ((x > 0) ? (1. / 3. / sqrd(cbrt(x))) : 1.) : cbrt(-x) / -3. / x;
}

// render_util.cuh:940-1102
__device__ inline int solve_cubic_legacy(double f0, double f1, double f2, double f3, double* st) {
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
    const double b = f2 / f3, c = f1 / f3, d = f0 / f3;  // the reference's a == 1 exactly
    const double f = ((3 * c) - sqrd(b)) / 3;
    const double g = ((2 * cubd(b)) - (9 * b * c) + (27 * d)) / 27;
    const double h = sqrd(g) / 4 + cubd(f) / 27;
    if ((fabs(f) < eps) && (fabs(g) < eps) && (fabs(h) < eps)) {
        st[0] = cond_cbrt(d);
        return CUBIC_ONE_R;
    }
    if (h <= 0) {
        if (h == 0) return CUBIC_NO_ROOT;
        const double i = sqrt((sqrd(g) / 4.) - h);
        const double j = cbrt(i);
        const double k = acos(-(g / (2 * i)));
        const double M = cos(k / 3.);
        const double N = sqrt(3.) * sin(k / 3.);
        const double P = (b / 3.) * -1;
        st[0] = -1 * j * (M + N) + P;
        st[1] = -1 * j * (M - N) + P;
        st[2] = 2 * j * M + P;
        return CUBIC_THREE_R;
    }
    const double R = -(g / 2.) + sqrt(h);
    const double T = -(g / 2.) - sqrt(h);
    st[0] = (cond_cbrt(R) + cond_cbrt(T)) - (b / 3.);
    return CUBIC_ONE_R_;
}

// render_util.cuh:1418-1702: grad[k] holds dL/dt on entry, dL/df_k on exit.
// m[k] = d t / d f_k of root st_id through the f/g/h chain (0 where the reference assigns 0).
__device__ inline void cubic_root_grad_legacy(int type, int st_id, const double* fs, float* grad) {
    double m[4] = {0., 0., 0., 0.};
    if (type == CUBIC_LINEAR) {
        m[0] = -1. / fs[1];
        m[1] = fs[0] / sqrd(fs[1]);
    } else if (type == CUBIC_POLY_ONE_R || type == CUBIC_POLY) {
        const double D = sqrd(fs[1]) - 4. * fs[2] * fs[0];
        const double sD = sqrt(D);
        // st_id 0 of the two-root case takes the "-sqrt" root's derivative whatever the sign of f2
        // (both branches of :1447-1462 are the same code)
        const bool small = (type == CUBIC_POLY) && (st_id == 0);
        const double sgn = small ? -1. : 1.;
        const double dt_dD = sgn / (4. * fs[2] * sD);
        const double den = (type == CUBIC_POLY) ? 2. * sqrd(fs[2]) : 4. * sqrd(fs[2]);
        m[0] = -sgn / sD;
        m[1] = (-1) / (2 * fs[2]) + (dt_dD * 2 * fs[1]);
        m[2] = (fs[1] - sgn * sD) / den + (dt_dD * (-4) * fs[0]);
    } else {
        const double b = fs[2] / fs[3], c = fs[1] / fs[3], d = fs[0] / fs[3];
        const double Db[4] = {0., 0., 1. / fs[3], (-fs[2]) / fs[3] / fs[3]};
        const double Dc[4] = {0., 1. / fs[3], 0., (-fs[1]) / fs[3] / fs[3]};
        const double Dd[4] = {1. / fs[3], 0., 0., (-fs[0]) / fs[3] / fs[3]};
        if (type == CUBIC_ONE_R) {
            const double dc = d_cond_cbrt(d);
            m[0] = dc * Dd[0];
            m[3] = dc * Dd[3];
        } else {
            const double f = ((3. * c) - sqrd(b)) / 3.;
            const double g = ((2. * cubd(b)) - (9. * b * c) + (27. * d)) / 27.;
            const double h = sqrd(g) / 4. + cubd(f) / 27.;
            const double Df_Db = -2. * b / 3., Df_Dc = 1.;
            const double Dg_Db = -c / 3. + 2. * sqrd(b) / 9., Dg_Dc = -b / 3., Dg_Dd = 1.;
            const double Dh_Df = sqrd(f) / 9., Dh_Dg = g / 2.;
            const double Dh_Db = Dh_Df * Df_Db + Dh_Dg * Dg_Db;
            const double Dh_Dc = Dh_Df * Df_Dc + Dh_Dg * Dg_Dc;
            const double Dh_Dd = Dh_Dg * Dg_Dd;
            double Dg[4], Dh[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                Dg[k] = Dg_Db * Db[k] + Dg_Dc * Dc[k] + Dg_Dd * Dd[k];
                Dh[k] = Dh_Db * Db[k] + Dh_Dc * Dc[k] + Dh_Dd * Dd[k];
            }
            const double Dst_Db = -1. / 3.;
            if (type == CUBIC_THREE_R) {
                const double w = sqrd(g) / 4. - h;
                const double i = sqrt(w);
                const double j = cbrt(i);
                const double k = acos(-(g / (2. * i)));
                const double M = cos(k / 3.);
                const double sn = sin(k / 3.);
                const double N = sqrt(3.) * sn;
                const double Dj_Dg = g / (12. * pow(w, 5. / 6.));
                const double Dj_Dh = -1. / (6. * pow(w, 5. / 6.));
                const double rad = sqrt(-sqrd(g) / (4. * w) + 1.);
                const double Dk_Dg = -(sqrd(g) / (8. * pow(w, 3. / 2.)) - 1. / (2. * sqrt(w))) / rad;
                const double Dk_Dh = g / (4. * pow(w, 3. / 2.) * rad);
                double Dst_Dj, Dst_Dk;
                if (st_id == 0) {
                    Dst_Dj = -M - N;
                    Dst_Dk = -j * (-sn / 3. + sqrt(3.) * M / 3.);
                } else if (st_id == 1) {
                    Dst_Dj = -M + N;
                    Dst_Dk = -j * (-sn / 3. - sqrt(3.) * M / 3.);
                } else {
                    Dst_Dj = 2. * M;
                    Dst_Dk = -2. * j * sn / 3.;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    m[q] = Dst_Dj * (Dj_Dg * Dg[q] + Dj_Dh * Dh[q]) +
                           Dst_Dk * (Dk_Dg * Dg[q] + Dk_Dh * Dh[q]) + Dst_Db * Db[q];
            } else {  // CUBIC_ONE_R_
                const double R = -(g / 2.) + sqrt(h);
                const double T = -(g / 2.) - sqrt(h);
                const double dR = d_cond_cbrt(R), dT = d_cond_cbrt(T);
                const double DR_Dh = 1. / (2. * sqrt(h)), DT_Dh = -1. / (2. * sqrt(h));
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    m[q] = dR * (-0.5 * Dg[q] + DR_Dh * Dh[q]) + dT * (-0.5 * Dg[q] + DT_Dh * Dh[q]) +
                           Dst_Db * Db[q];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) grad[k] = (m[k] == 0.) ? 0.f : grad[k] * (float)m[k];
}

// render_util.cuh:1705-1775: d f_k / d s_corner of surface_cubic_global in float32.  Each
// corner's weight along the ray is (A_x + B_x t)(A_y + B_y t)(A_z + B_z t); f_k is its t^k term.
__device__ __forceinline__ void surface_cubic_global_grad(const float* o, const float* d, const int* l,
                                                          const float* gf, float* gs) {
#pragma unroll
    for (int cr = 0; cr < 8; ++cr) {
        float A[3], Bc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const bool hi = (cr >> (2 - k)) & 1;
            A[k] = hi ? (-(float)l[k] + o[k]) : ((float)l[k] - o[k] + 1);
            Bc[k] = hi ? d[k] : -d[k];
        }
        const float t0 = A[0] * A[1] * A[2];
        const float t1 = Bc[0] * A[1] * A[2] + A[0] * (Bc[1] * A[2] + Bc[2] * A[1]);
        const float t2 = Bc[0] * (Bc[1] * A[2] + Bc[2] * A[1]) + Bc[1] * Bc[2] * A[0];
        const float t3 = Bc[0] * Bc[1] * Bc[2];
        gs[cr] = gf[0] * t0 + gf[1] * t1 + gf[2] * t2 + gf[3] * t3;
    }
}

// One listed voxel: gates + coefficients (surface.cu:66-126 == :385-416)
struct LegVox {
    int l[3];
    int lk[8];
    double fs[4];  // fs[0] already has the level set subtracted
    bool has_level;
};

__device__ __forceinline__ bool legacy_load(const KGrid& g, const Ray& r, const int* vox, LegVox& v) {
    v.l[0] = vox[0];
    v.l[1] = vox[1];
    v.l[2] = vox[2];
    // (negative coordinates would make the reference read out of bounds; they are skipped here)
    if (v.l[0] < 0 || v.l[1] < 0 || v.l[2] < 0 || v.l[0] + 1 >= g.size[0] || v.l[1] + 1 >= g.size[1] ||
        v.l[2] + 1 >= g.size[2])
        return false;
    const int32_t* p = g.links + ((size_t)v.l[0] * g.offx + (size_t)v.l[1] * g.offy + v.l[2]);
    v.lk[0] = p[0];
    v.lk[1] = p[1];
    v.lk[2] = p[g.offy];
    v.lk[3] = p[g.offy + 1];
    v.lk[4] = p[g.offx];
    v.lk[5] = p[g.offx + 1];
    v.lk[6] = p[g.offx + g.offy];
    v.lk[7] = p[g.offx + g.offy + 1];
    int any = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) any |= v.lk[i];
    if (any < 0) return false;
    double s[8];
    double mn = 0, mx = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        s[i] = (double)g.surface[v.lk[i]];
        mn = (i == 0) ? s[i] : fmin(mn, s[i]);
        mx = (i == 0) ? s[i] : fmax(mx, s[i]);
    }
    const double od[3] = {r.o[0], r.o[1], r.o[2]};
    const double dd[3] = {r.d[0], r.d[1], r.d[2]};
    surface_cubic_global(s, od, dd, v.l, v.fs);
    const double lv = (double)g.level_set[0];
    v.has_level = !((lv < mn) || (lv > mx));
    v.fs[0] -= lv;
    return true;
}

__device__ __forceinline__ bool legacy_position(const KGrid& g, const Ray& r, const LegVox& v, double st,
                                                float* pos) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pos[k] = fmaf((float)st, r.d[k], r.o[k]);
        pos[k] -= (float)min(v.l[k], g.size[k] - 2);
    }
    return !((pos[0] < 0) | (pos[0] > 1) | (pos[1] < 0) | (pos[1] > 1) | (pos[2] < 0) | (pos[2] > 1));
}

// Per-lane cursor over the ray's voxel list: advances to the next sample.
struct LegCursor {
    int vi, num;
    long long start;
    int j;        // next root to look at (3: voxel finished)
    int type;
    bool loaded;  // the current voxel passed the gates (the stop test applies to it)
    double st0, st1, st2;  // scalars: a runtime-indexed array would live in scratch
    LegVox v;
};

// Moves the lane to its next sample; returns false when the ray is finished.
// `stopped(log_transmit)` implements the reference's per-voxel stop test.
template <typename StopFn>
__device__ __forceinline__ bool legacy_next(const KGrid& g, const Ray& r, const int* voxel_ls, LegCursor& c,
                                            float* pos, int* root_id, StopFn&& stop_after_voxel) {
    for (;;) {
        while (c.j < 3) {
            const int j = c.j++;
            const double a0 = c.st0, a1 = c.st1, a2 = c.st2;
            const double stj = (j == 0) ? a0 : ((j == 1) ? a1 : a2);
            if (!(stj > 0)) continue;
            if (!legacy_position(g, r, c.v, stj, pos)) continue;
            *root_id = j;
            return true;
        }
        if (c.loaded) {
            c.loaded = false;
            if (stop_after_voxel()) return false;
        }
        if (++c.vi >= c.num) return false;
        if (!legacy_load(g, r, voxel_ls + 3 * (c.start + c.vi), c.v)) continue;
        c.loaded = true;
        c.st0 = c.st1 = c.st2 = -1;
        c.type = CUBIC_NO_ROOT;
        if (c.v.has_level) {
            double st[3] = {-1, -1, -1};
            c.type = solve_cubic_legacy(c.v.fs[0], c.v.fs[1], c.v.fs[2], c.v.fs[3], st);
            c.st0 = st[0];
            c.st1 = st[1];
            c.st2 = st[2];
            c.j = 0;
        }
    }
}

__device__ __forceinline__ void legacy_cursor_init(LegCursor& c, const int* vox_start, const int* vox_num,
                                                   long long n_vox, long long q, bool active) {
    c.vi = -1;
    c.j = 3;
    c.loaded = false;
    c.type = CUBIC_NO_ROOT;
    c.st0 = c.st1 = c.st2 = -1;
    c.start = active ? (long long)vox_start[q] : 0;
    int num = active ? vox_num[q] : 0;
    // never read past the list the caller handed over
    if (c.start < 0 || c.start > n_vox) num = 0;
    if ((long long)num > n_vox - c.start) num = (int)(n_vox - c.start);
    c.num = num;
}

// ------------------------------------------------------------------------------------
// Forward: trace_ray_surface (surface.cu:35-234), render_ray_kernel (:873-913)
// ------------------------------------------------------------------------------------
template <int B>
__global__ void __launch_bounds__(kBlock)
k_surface_fwd(KGrid g, KOpt o, const float* __restrict__ origins, const float* __restrict__ dirs,
              long long Q, const int* __restrict__ voxel_ls, const int* __restrict__ vox_start,
              const int* __restrict__ vox_num, long long n_vox, float* __restrict__ rgb_out) {
    __shared__ float s_sph[kBlock * kSphPad];
    __shared__ LaneRec s_rec[kBlock];
    const int tid = threadIdx.x;
    const long long q = (long long)blockIdx.x * kBlock + tid;
    const bool valid = q < Q;
    Ray r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.o[k] = valid ? origins[3 * q + k] : 0.f;
        r.d[k] = valid ? dirs[3 * q + k] : 1.f;
    }
    {
        float sph[B];
        eval_sh<B>(r.d, sph);
#pragma unroll
        for (int k = 0; k < B; ++k) s_sph[tid * kSphPad + k] = sph[k];
    }
    setup_ray(g, o, r);
    bool active = valid && !(r.tmin > r.tmax);
    const bool misses = valid && !active;
    LegCursor c;
    legacy_cursor_init(c, vox_start, vox_num, n_vox, q, active);
    float outv[3] = {0.f, 0.f, 0.f};
    float log_transmit = 0.f;
    const float* sph_wave = s_sph + (tid & ~63) * kSphPad;
    const LaneRec* rec_wave = s_rec + (tid & ~63);
    while (__any(active)) {
        float pos[3] = {0.f, 0.f, 0.f};
        int root_id = 0;
        bool hit = false;
        if (active) {
            hit = legacy_next(g, r, voxel_ls, c, pos, &root_id, [&]() {
                if (expf(log_transmit) < o.stop_thresh) {  // surface.cu:214-217
                    log_transmit = -1e3f;
                    return true;
                }
                return false;
            });
            if (!hit) active = false;
        }
        const unsigned long long need = __ballot(hit);
        if (!need) continue;
        if (hit) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s_rec[tid].link[i] = c.v.lk[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float color[3] = {0.f, 0.f, 0.f};
        coop_shade_fwd<B>(g, need, rec_wave, sph_wave, pos, color);
        if (hit) {
            float dens[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) dens[i] = g.density[c.v.lk[i]];
            const float alpha = alpha_act(trilerp8(dens, pos), 0);
            const float pcnt = -1 * logf(1 - alpha);
            const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
            log_transmit -= pcnt;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) outv[ch] += weight * fmaxf(color[ch] + 0.5f, 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (valid) {
        const float bg = o.background_brightness;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
            rgb_out[3 * q + ch] = misses ? bg : outv[ch] + expf(log_transmit) * bg;
    }
}

// ------------------------------------------------------------------------------------
// Backward: trace_ray_surface_backward (surface.cu:341-645), render_ray_backward_kernel
// (:954-1053).  GT: grad_in is rgb_gt and the MSE gradient is formed here (fused entry).
// ------------------------------------------------------------------------------------
template <int B>
__global__ void __launch_bounds__(kBlock)
k_surface_bwd(KGrid g, KOpt o, KGrads G, const float* __restrict__ origins, const float* __restrict__ dirs,
              long long Q, const int* __restrict__ voxel_ls, const int* __restrict__ vox_start,
              const int* __restrict__ vox_num, long long n_vox, const float* __restrict__ grad_in,
              const float* __restrict__ color_cache, int grad_in_is_rgb, float sparsity_loss) {
    __shared__ float s_sph[kBlock * kSphPad];
    __shared__ LaneRec s_rec[kBlock];
    const int tid = threadIdx.x;
    const long long q = (long long)blockIdx.x * kBlock + tid;
    const bool valid = q < Q;
    Ray r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.o[k] = valid ? origins[3 * q + k] : 0.f;
        r.d[k] = valid ? dirs[3 * q + k] : 1.f;
    }
    {
        float sph[B];
        eval_sh<B>(r.d, sph);
#pragma unroll
        for (int k = 0; k < B; ++k) s_sph[tid * kSphPad + k] = sph[k];
    }
    setup_ray(g, o, r);
    bool active = valid && !(r.tmin > r.tmax);
    float gout[3] = {0.f, 0.f, 0.f};
    float accum = 0.f;
    if (valid) {
        const float norm = 2.f / (3 * (int)Q);  // :1013-1018
        float cc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            cc[i] = color_cache[3 * q + i];
            gout[i] = grad_in_is_rgb ? (cc[i] - grad_in[3 * q + i]) * norm : grad_in[3 * q + i];
        }
        accum = fmaf(cc[0], gout[0], fmaf(cc[1], gout[1], cc[2] * gout[2]));
    }
    LegCursor c;
    legacy_cursor_init(c, vox_start, vox_num, n_vox, q, active);
    float log_transmit = 0.f;
    const float* sph_wave = s_sph + (tid & ~63) * kSphPad;
    const LaneRec* rec_wave = s_rec + (tid & ~63);
    while (__any(active)) {
        float pos[3] = {0.f, 0.f, 0.f};
        int root_id = 0;
        bool hit = false;
        if (active) {
            hit = legacy_next(g, r, voxel_ls, c, pos, &root_id,
                              [&]() { return expf(log_transmit) < o.stop_thresh; });  // :629-631
            if (!hit) active = false;
        }
        const unsigned long long need = __ballot(hit);
        if (!need) continue;
        float dens[8];
        float raw_alpha = 0.f, alpha = 0.f, pcnt = 0.f, weight = 0.f;
        if (hit) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s_rec[tid].link[i] = c.v.lk[i];
                dens[i] = g.density[c.v.lk[i]];
            }
            raw_alpha = trilerp8(dens, pos);
            alpha = alpha_act(raw_alpha, 0);
            pcnt = -1 * logf(1 - alpha);
            weight = expf(log_transmit) * (1.f - expf(-pcnt));
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float total_color = 0.f;
        float grad_xyz[3] = {0.f, 0.f, 0.f};
        coop_shade_bwd<B>(g, G, true, need, rec_wave, sph_wave, pos, weight, gout, &total_color, grad_xyz);
        if (hit) {
            accum -= weight * total_color;
            float curr_grad_alpha = accum / (alpha - 1) + total_color * expf(log_transmit);  // :495
            log_transmit -= pcnt;
            if (sparsity_loss > 0.f)  // Cauchy, :498-501
                curr_grad_alpha += sparsity_loss * (4 * alpha / (1 + 2 * (alpha * alpha)));
            const double sg = 1. / (1. + (double)expf(-raw_alpha));  // _D_SIGMOID, util.hpp:29
            const float curr_grad_raw_alpha = (float)((double)curr_grad_alpha * (sg * (1 - sg)));
            {
                float cw[8];
                corner_weights(pos, curr_grad_raw_alpha, cw);
                scatter8(G.grad_density, G.mask, G.mask2, c.v.lk, cw);
            }
            trilerp_dpos(dens, pos, curr_grad_raw_alpha, grad_xyz);
            const float grad_st = grad_xyz[0] * r.d[0] + grad_xyz[1] * r.d[1] + grad_xyz[2] * r.d[2];
            float grad_fs[4] = {grad_st, grad_st, grad_st, grad_st};
            cubic_root_grad_legacy(c.type, root_id, c.v.fs, grad_fs);
            int lc[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) lc[k] = min(c.v.l[k], g.size[k] - 2);
            float gs[8];
            surface_cubic_global_grad(r.o, r.d, lc, grad_fs, gs);
            scatter8(G.grad_surface, G.mask, G.mask2, c.v.lk, gs);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// test_cubic_root_grad with use_vieta == false: Vieta roots, f/g/h gradient (test_cuda.cu:32-106)
__global__ void k_test_cubic_root_grad_legacy(const double* __restrict__ fs, const int* __restrict__ st_ids,
                                              float* __restrict__ grad_fs, float* __restrict__ st_out,
                                              long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double f[4] = {fs[4 * i], fs[4 * i + 1], fs[4 * i + 2], fs[4 * i + 3]};
    double st[3] = {-1, -1, -1};
    const int type = solve_cubic_vieta(f[0], f[1], f[2], f[3], st);
#pragma unroll
    for (int j = 0; j < 3; ++j) st_out[3 * i + j] = (float)st[j];
    int root_num = 1;
    if (type == CUBIC_POLY) root_num = 2;
    else if (type == CUBIC_THREE_R) root_num = 3;
    float gr[4] = {grad_fs[4 * i], grad_fs[4 * i + 1], grad_fs[4 * i + 2], grad_fs[4 * i + 3]};
    if (type == CUBIC_NO_ROOT || st_ids[i] >= root_num) {
        gr[0] = gr[1] = gr[2] = gr[3] = 0.f;
    } else {
        cubic_root_grad_legacy(type, st_ids[i], f, gr);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) grad_fs[4 * i + j] = gr[j];
}

// ------------------------------------------------------------------------------------
// Voxel lists for the backend above, from the surf_trav DDA (surf_trav.cu:75-197).  Stands in
// for SparseGrid.find_ray_voxels_intersection (svox2.py:1515-1594), whose kaolin SPC ray trace is
// not constructible in the reference.  FILL = false counts, = true writes.
// skip_empty drops voxels with a missing link (the renderer skips them anyway).
// ------------------------------------------------------------------------------------
template <bool FILL>
__global__ void __launch_bounds__(kBlock)
k_voxel_lists(KGrid g, KOpt o, const float* __restrict__ origins, const float* __restrict__ dirs,
              long long Q, int skip_empty, int* __restrict__ vox_num, const int* __restrict__ vox_start,
              int* __restrict__ voxel_ls, long long n_vox) {
    const long long q = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (q >= Q) return;
    Ray r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.o[k] = origins[3 * q + k];
        r.d[k] = dirs[3 * q + k];
    }
    setup_ray(g, o, r);
    int n = 0;
    const long long start = FILL ? (long long)vox_start[q] : 0;
    const int cap = FILL ? vox_num[q] : 0;
    if (!(r.tmin > r.tmax)) {
        Walk w;
        walk_init(g, r, w);
        bool active = true;
        while (active) {
            walk_step(g, r, w);
            active = (w.t <= r.tmax);
            if (skip_empty && !links_present(g, w)) continue;
            if constexpr (FILL) {
                const long long at = start + n;
                if (n < cap && at >= 0 && at < n_vox) {
                    voxel_ls[3 * at] = w.vx;
                    voxel_ls[3 * at + 1] = w.vy;
                    voxel_ls[3 * at + 2] = w.vz;
                }
            }
            ++n;
        }
    }
    if constexpr (!FILL) vox_num[q] = n;
}

}  // namespace asurf
