// asurf_oracle.cpp -- CPU ORACLE for the alphaSurf surf_trav hot path.
//
// TEST INFRASTRUCTURE ONLY.  Scalar restatement (one ray at a time, OpenMP over
// rays) of the CUDA semantics of the reference renderer.  The product never links
// this file.  Citations are file:line in the reference checkout
// (svox2/csrc/render_lerp_kernel_surf_trav.cu = "surf_trav.cu",
//  svox2/csrc/include/render_util.cuh = "render_util.cuh").
//
// Arithmetic policy (compile with -ffp-contract=off): float where the reference is
// float, double where it is double, fused multiply-adds only where the reference
// writes fmaf()/lerp() explicitly (plus the voxel entry point o + t_close*d, which
// nvcc contracts).  min/max on floats follow CUDA's fminf/fmaxf NaN rules.
// expf/logf/cosf are libm (the reference's __expf/__logf/__cosf fast intrinsics are
// approximations of these).
#include "asurf_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

namespace {

constexpr double kPi = 3.1415926535897931e+0;  // render_util.cuh:7
constexpr int kMaxBasis = 25;
constexpr int kMaxCache = 1024;

enum CubicType {  // data_spec.hpp:24-31
    CUBIC_NO_ROOT = 200,
    CUBIC_LINEAR = 201,
    CUBIC_POLY_ONE_R = 202,
    CUBIC_POLY = 203,
    CUBIC_THREE_R = 205,
    CUBIC_ONE_R_ = 206,
};

inline float lerpf(float a, float b, float w) { return fmaf(w, b - a, a); }  // cuda_util.cuh:75-77
inline float sqrf(float x) { return x * x; }
inline double sqrd(double x) { return x * x; }
inline double cubd(double x) { return x * x * x; }

// ---------------------------------------------------------------------------------
// SH basis: render_util.cuh:405-436 (1/4/9); the 16-term band is the reference's own
// commented-out block (:414-423) == svox2/utils.py:147-154.
// ---------------------------------------------------------------------------------
void eval_sh(int basis_dim, const float* dir, float* out) {
    const float C0 = 0.28209479177387814f;
    const float C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                         -1.0925484305920792f, 0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                         0.3731763325901154f,  -0.4570457994644658f, 1.445305721320277f,
                         -0.5900435899266435f};
    out[0] = C0;
    const float x = dir[0], y = dir[1], z = dir[2];
    const float xx = x * x, yy = y * y, zz = z * z;
    const float xy = x * y, yz = y * z, xz = x * z;
    if (basis_dim >= 16) {
        out[9] = C3[0] * y * (3 * xx - yy);
        out[10] = C3[1] * xy * z;
        out[11] = C3[2] * y * (4 * zz - xx - yy);
        out[12] = C3[3] * z * (2 * zz - 3 * xx - 3 * yy);
        out[13] = C3[4] * x * (4 * zz - xx - yy);
        out[14] = C3[5] * z * (xx - yy);
        out[15] = C3[6] * x * (xx - 3 * yy);
    }
    if (basis_dim >= 9) {
        out[4] = C2[0] * xy;
        out[5] = C2[1] * yz;
        out[6] = (float)((double)C2[2] * (2.0 * (double)zz - (double)xx - (double)yy));  // :427
        out[7] = C2[3] * xz;
        out[8] = C2[4] * (xx - yy);
    }
    if (basis_dim >= 4) {
        out[1] = -C1 * y;
        out[2] = C1 * z;
        out[3] = -C1 * x;
    }
}

struct Ray {
    float origin[3];
    float dir[3];
    float tmin, tmax, world_step;
    float sph[kMaxBasis];
};

// Kernel prologue: SingleRaySpec::set + calc_sphfunc (on the WORLD direction) +
// ray_find_bounds.  surf_trav.cu:3163-3171, render_util.cuh:536-547, 651-693.
void setup_ray(const oracle_grid& g, const oracle_render_options& o, const float* org,
               const float* dir, Ray& r) {
    for (int k = 0; k < 3; ++k) {
        r.origin[k] = org[k];
        r.dir[k] = dir[k];
    }
    eval_sh(g.basis_dim, r.dir, r.sph);
    for (int k = 0; k < 3; ++k) r.origin[k] = fmaf(r.origin[k], g.scaling[k], g.offset[k]);
    for (int k = 0; k < 3; ++k) r.dir[k] *= g.scaling[k];
    const float rn = 1.f / sqrtf(r.dir[0] * r.dir[0] + r.dir[1] * r.dir[1] + r.dir[2] * r.dir[2]);
    for (int k = 0; k < 3; ++k) r.dir[k] *= rn;
    r.world_step = rn * o.step_size;

    if (o.use_spheric_clip) {  // render_util.cuh:661-680
        float so[3], sd[3];
        for (int i = 0; i < 3; ++i) {
            const float ss = 2.f / float(g.size[i]);
            so[i] = fmaf(r.origin[i] + 0.5f, ss, -1.f);
            sd[i] = r.dir[i] * ss;
        }
        const float q2a = 2 * (sd[0] * sd[0] + sd[1] * sd[1] + sd[2] * sd[2]);
        const float qb = 2 * (so[0] * sd[0] + so[1] * sd[1] + so[2] * sd[2]);
        const float f = qb * qb - 2 * q2a * (so[0] * so[0] + so[1] * so[1] + so[2] * so[2]);
        const float r_out = 1.f, r_in = 1.f - o.near_clip;
        const float det_out = f + 2 * q2a * r_out * r_out;
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
        r.tmin = o.near_clip / r.world_step * o.step_size;
        r.tmax = 2e3f;
        for (int i = 0; i < 3; ++i) {
            const float invdir = 1.0f / r.dir[i];
            const float t1 = (-0.5f - r.origin[i]) * invdir;
            const float t2 = ((float)g.size[i] - 0.5f - r.origin[i]) * invdir;
            if (r.dir[i] != 0.f) {
                r.tmin = fmaxf(r.tmin, fminf(t1, t2));
                r.tmax = fminf(r.tmax, fmaxf(t1, t2));
            }
        }
    }
}

// ---------------------------------------------------------------------------------
// One step of the voxel DDA.  surf_trav.cu:86-197 (forward) == :1834-1920 (backward).
// ---------------------------------------------------------------------------------
struct Dda {
    int32_t next_voxel[3];
    int32_t voxel[3];
    float t;
    float t_close, t_far;
};

void dda_init(const oracle_grid& g, const Ray& r, Dda& d) {  // surf_trav.cu:67-79
    d.t = r.tmin;
    for (int j = 0; j < 3; ++j) {
        int32_t v = (int32_t)fmaf(d.t, r.dir[j], r.origin[j]);
        v = std::min(std::max(v, 0), g.size[j] - 2);
        d.next_voxel[j] = v;
    }
}

void dda_step(const oracle_grid& g, const Ray& r, Dda& d) {
    for (int j = 0; j < 3; ++j) d.voxel[j] = d.next_voxel[j];
    float tc[3], tf[3];
    for (int j = 0; j < 3; ++j) {
        const int32_t close_plane = r.dir[j] > 0.f ? d.voxel[j] : d.voxel[j] + 1;
        const int32_t far_plane = r.dir[j] > 0.f ? d.voxel[j] + 1 : d.voxel[j];
        tc[j] = ((float)close_plane - r.origin[j]) / r.dir[j];
        tf[j] = ((float)far_plane - r.origin[j]) / r.dir[j];
    }
    d.t_close = fmaxf(fmaxf(fmaxf(tc[0], tc[1]), tc[2]), 0.f);
    d.t_far = fminf(fminf(tf[0], tf[1]), tf[2]);
    d.t = d.t_far;
    int ax;
    if (d.t_far == tf[0]) ax = 0;
    else if (d.t_far == tf[1]) ax = 1;
    else ax = 2;
    d.next_voxel[ax] += (r.dir[ax] > 0.f) ? 1 : -1;
    if (d.next_voxel[ax] < 0 || d.next_voxel[ax] >= g.size[ax] - 1) d.t = r.tmax + 1.f;
}

// The eight corner rows of a voxel; false when the voxel is (partly) empty.
// surf_trav.cu:212-221.
bool voxel_links(const oracle_grid& g, const int32_t* l, int32_t* lk) {
    if (l[0] + 1 >= g.size[0] || l[1] + 1 >= g.size[1] || l[2] + 1 >= g.size[2]) return false;
    const int64_t offy = g.size[2];
    const int64_t offx = (int64_t)g.size[1] * g.size[2];
    const int32_t* p = g.links + (offx * l[0] + offy * l[1] + l[2]);
    lk[0] = p[0];
    lk[1] = p[1];
    lk[2] = p[offy];
    lk[3] = p[offy + 1];
    lk[4] = p[offx];
    lk[5] = p[offx + 1];
    lk[6] = p[offx + offy];
    lk[7] = p[offx + offy + 1];
    for (int i = 0; i < 8; ++i)
        if (lk[i] < 0) return false;
    return true;
}

// trilerp_cuvol_one, render_util.cuh:72-92 (corner order 000,001,010,011,100,...;
// the last index is z, the first x).
float trilerp(const float* v, const float* pos) {
    const float ix0y0 = lerpf(v[0], v[1], pos[2]);
    const float ix0y1 = lerpf(v[2], v[3], pos[2]);
    const float ix0 = lerpf(ix0y0, ix0y1, pos[1]);
    const float ix1y0 = lerpf(v[4], v[5], pos[2]);
    const float ix1y1 = lerpf(v[6], v[7], pos[2]);
    const float ix1 = lerpf(ix1y0, ix1y1, pos[1]);
    return lerpf(ix0, ix1, pos[0]);
}

// trilinear corner weights for the backward scatter, render_util.cuh:103-120
void corner_weights(const float* pos, float g, float* w) {
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

// d(trilerp)/d(pos) * grad_in accumulated into out[3]; render_util.cuh:156-204
void trilerp_dpos(const float* v, const float* pos, float grad_in, float* out) {
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

// compute_field_grad, render_util.cuh:2190-2236
void field_grad(const float* v, const float* pos, float* out) {
    const float ix0y0 = lerpf(v[0], v[1], pos[2]);
    const float ix0y1 = lerpf(v[2], v[3], pos[2]);
    const float ix0 = lerpf(ix0y0, ix0y1, pos[1]);
    const float ix1y0 = lerpf(v[4], v[5], pos[2]);
    const float ix1y1 = lerpf(v[6], v[7], pos[2]);
    const float ix1 = lerpf(ix1y0, ix1y1, pos[1]);
    out[0] = ix1 - ix0;
    // the reference mixes float and double literals here (1. - pos[0]): evaluate in double
    out[1] = (float)((double)pos[0] * (double)(-ix1y0 + ix1y1) +
                     (1. - (double)pos[0]) * (double)(-ix0y0 + ix0y1));
    out[2] = pos[0] * (pos[1] * (-v[6] + v[7]) + (1 - pos[1]) * (-v[4] + v[5])) +
             (1 - pos[0]) * (pos[1] * (-v[2] + v[3]) + (1 - pos[1]) * (-v[0] + v[1]));
}

// ---------------------------------------------------------------------------------
// Trilinear field along the ray as a cubic in t.  render_util.cuh:789-848
// (surface_to_cubic_equation_01).  s(x,y,z) = sum_ijk s_ijk X_i Y_j Z_k with
// X_0 = (1-ox) - dx t, X_1 = ox + dx t, ... multiplied out z -> y -> x.
// ---------------------------------------------------------------------------------
void surface_cubic(const double* s, const double* o, const double* d, double* f) {
    // interpolate along z: m_ij(t) = a_ij + b_ij t
    double a[4], b[4];
    for (int ij = 0; ij < 4; ++ij) {
        a[ij] = s[2 * ij] * (1 - o[2]) + s[2 * ij + 1] * o[2];
        b[ij] = d[2] * (s[2 * ij + 1] - s[2 * ij]);
    }
    // along y: n_i(t) = c_i0 + c_i1 t + c_i2 t^2
    double c[2][3];
    for (int i = 0; i < 2; ++i) {
        const double a0 = a[2 * i], a1 = a[2 * i + 1], b0 = b[2 * i], b1 = b[2 * i + 1];
        c[i][0] = a0 * (1 - o[1]) + a1 * o[1];
        c[i][1] = (b0 * (1 - o[1]) + b1 * o[1]) + d[1] * (a1 - a0);
        c[i][2] = d[1] * (b1 - b0);
    }
    // along x
    f[0] = c[0][0] * (1 - o[0]) + c[1][0] * o[0];
    f[1] = (c[0][1] * (1 - o[0]) + c[1][1] * o[0]) + d[0] * (c[1][0] - c[0][0]);
    f[2] = (c[0][2] * (1 - o[0]) + c[1][2] * o[0]) + d[0] * (c[1][1] - c[0][1]);
    f[3] = d[0] * (c[1][2] - c[0][2]);
}

// d f_k / d s_ijk (float), render_util.cuh:850-934 (calc_surface_grad_01): the
// coefficient of t^k in X_i(t) Y_j(t) Z_k(t).
void surface_cubic_grad(const float* o, const float* d, const float* gf, float* gs) {
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

// ---------------------------------------------------------------------------------
// cubic_equation_solver_vieta, render_util.cuh:1126-1203
// ---------------------------------------------------------------------------------
int solve_cubic_vieta(double f0, double f1, double f2, double f3, double eps, double* st) {
    if (std::fabs(f3) < eps) {
        if (std::fabs(f2) < eps) {
            if (std::fabs(f1) < eps) return CUBIC_NO_ROOT;
            st[0] = -f0 / f1;
            return CUBIC_LINEAR;
        }
        const double D = sqrd(f1) - 4.0 * f2 * f0;
        const double sD = std::sqrt(D);
        if (D > 0) {
            if (f2 > 0) {
                st[0] = (-f1 - sD) / (2 * f2);
                st[1] = (-f1 + sD) / (2 * f2);
            } else {
                st[0] = (-f1 + sD) / (2 * f2);
                st[1] = (-f1 - sD) / (2 * f2);
            }
            if (std::fabs(st[0] - st[1]) < eps) {
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
        const double theta = std::acos(R / std::sqrt(cubd(Q)));
        st[0] = -2. * std::sqrt(Q) * std::cos(theta / 3.) - b / 3.;
        st[1] = -2. * std::sqrt(Q) * std::cos((theta - 2. * kPi) / 3.) - b / 3.;
        st[2] = -2. * std::sqrt(Q) * std::cos((theta + 2. * kPi) / 3.) - b / 3.;
        return CUBIC_THREE_R;
    }
    const double sgn = (R > 0.) ? 1. : -1.;  // cuda_util.cuh:145
    const double A = -sgn * std::cbrt(std::fabs(R) + std::sqrt(sqrd(R) - cubd(Q)));
    const double B = (A == 0.) ? 0. : Q / A;
    st[0] = (A + B) - b / 3.;
    return CUBIC_ONE_R_;
}

// calc_cubic_root_grad_vieta, render_util.cuh:1206-1415.  g[k] enters holding
// dL/d(root) and leaves holding dL/d f_k.
void cubic_root_grad_vieta(int type, int st_id, const double* fs, float* g) {
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
        const double sD = std::sqrt(D);
        const double dt_dD = 1 / (4. * f2 * sD);
        g[0] *= (float)(-1 / sD);
        g[1] *= (float)((-1) / (2 * f2) + dt_dD * 2 * f1);
        g[2] *= (float)((f1 - sD) / (4 * sqrd(f2)) + dt_dD * (-4) * f0);
        g[3] = 0.f;
        return;
    }
    if (type == CUBIC_POLY) {
        const double D = sqrd(f1) - 4.0 * f2 * f0;
        const double sD = std::sqrt(D);
        const double sq2 = sqrd(f2);
        // "S" = the -sqrt(D)/(2 f2)-type branch, "L" the other (macros at :1234-1244)
        const bool use_s = (st_id == 0);
        if (use_s) {
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
    if (type != CUBIC_THREE_R && type != CUBIC_ONE_R_) return;  // grads left untouched (:1265-1413)

    const double b = f2 / f3, c = f1 / f3, d = f0 / f3;
    const double Q = (sqrd(b) - 3. * c) / 9.;
    const double R = (2. * cubd(b) - 9. * b * c + 27. * d) / 54.;
    // d{Q,R,b}/d f_k, k = 0..3 (:1276-1289)
    const double dQ[4] = {0., -1. / (3. * f3), 2. * f2 / (9. * sqrd(f3)),
                          f1 / (3. * sqrd(f3)) - 2. * sqrd(f2) / (9 * cubd(f3))};
    const double dR[4] = {1. / (2. * f3), -f2 / (6. * sqrd(f3)),
                          -f1 / (6. * sqrd(f3)) + sqrd(f2) / (9 * cubd(f3)),
                          -f0 / (2. * sqrd(f3)) + f1 * f2 / (3. * cubd(f3)) -
                              cubd(f2) / (9 * (f3 * f3 * f3 * f3))};
    const double db[4] = {0., 0., 1. / f3, -f2 / sqrd(f3)};
    const double dst_db = -1. / 3.;

    if (type == CUBIC_THREE_R) {
        const double theta = std::acos(R / std::sqrt(cubd(Q)));
        const double dth_dQ =
            3. * R / (2. * Q * std::sqrt(1. - sqrd(R) / cubd(Q)) * std::sqrt(cubd(Q)));
        const double dth_dR = -1 / (std::sqrt(1 - sqrd(R) / cubd(Q)) * std::sqrt(cubd(Q)));
        double dst_dQ, dst_dth;
        if (st_id == 0) {
            dst_dQ = -std::cos(theta / 3.) / std::sqrt(Q);
            dst_dth = 2. * std::sqrt(Q) * std::sin(theta / 3.) / 3.;
        } else if (st_id == 1) {
            dst_dQ = std::cos(theta / 3. + kPi / 3.) / std::sqrt(Q);
            dst_dth = -2. * std::sqrt(Q) * std::sin(theta / 3. + kPi / 3.) / 3.;
        } else {
            dst_dQ = std::sin(theta / 3. + kPi / 6.) / std::sqrt(Q);
            dst_dth = 2. * std::sqrt(Q) * std::cos(theta / 3. + kPi / 6.) / 3.;
        }
        for (int k = 0; k < 4; ++k)
            g[k] *= (float)(dst_dth * (dth_dQ * dQ[k] + dth_dR * dR[k]) + dst_dQ * dQ[k] +
                            dst_db * db[k]);
        return;
    }
    // single real root (:1354-1411)
    const double sgn = (R > 0.) ? 1. : -1.;
    const double A = -sgn * std::cbrt(std::fabs(R) + std::sqrt(sqrd(R) - cubd(Q)));
    const double sq = std::fmax(std::sqrt(-cubd(Q) + sqrd(R)), 1e-10);
    const double dA_dR =
        (R >= 0.) ? (-(R / (3. * sq) + 1. / 3.) / std::fmax(std::cbrt(sqrd(R + sq)), 1e-10))
                  : ((R / (3. * sq) - 1. / 3.) / std::fmax(std::cbrt(sqrd(-R + sq)), 1e-10));
    const double dA_dQ = (R >= 0.) ? (sqrd(Q) / (2. * sq * std::cbrt(sqrd(R + sq))))
                                   : (-sqrd(Q) / (2. * sq * std::cbrt(sqrd(-R + sq))));
    const double dB_dA = (A == 0.) ? 0. : -Q / sqrd(A);
    const double dB_dQ = (A == 0.) ? 0. : 1. / A;
    for (int k = 0; k < 4; ++k)
        g[k] *= (float)((dB_dA + 1.) * (dA_dQ * dQ[k] + dA_dR * dR[k]) + dB_dQ * dQ[k] +
                        dst_db * db[k]);
}

// ---------------------------------------------------------------------------------
// activations & truncation window: render_util.cuh:2138-2188
// ---------------------------------------------------------------------------------
float alpha_act(float raw, int type) {
    if (type == 0) return (float)(1. / (1. + (double)expf(-raw)));  // _SIGMOID, util.hpp:16
    return (raw >= 0.) ? 1 - expf(-raw) : 0.f;
}
float alpha_act_grad(float alpha, int type) {
    if (type == 0) return alpha * (1 - alpha);
    return (alpha > 0.) ? 1 - alpha : 0.f;
}
float trunc_rw(float x, float a, float clamp_min) {
    const float u = fminf(fmaxf(a - x, 0.f), 1.f);
    const float cs = cosf((float)(kPi * (double)u));
    return fmaxf(.5f * (1.f - cs), clamp_min);
}

// _LOG(max(x, 1e-8) / sum): the max and the division are double, the log float
// (surf_trav.cu:1769-1770, 2193, 2282)
inline float log_ratio(float x, float sum) {
    return logf((float)(std::fmax((double)x, 1e-8) / (double)sum));
}
// sparsity ("log raw alpha") term added to d/d(rwalpha), surf_trav.cu:2286-2300
inline float sparsity_term(float curr, float sparsity_loss, float a, float weight, float wsum) {
    const float _1_a = (float)std::fmax((double)(1.f - a), 1e-8);
    const double add = -(double)sparsity_loss * (1.0 / std::fmin((double)(_1_a * logf(_1_a)), -1e-8)) *
                       (double)(1.f - weight / wsum);
    return (float)((double)curr + add);
}

// Per-voxel state shared by the forward and backward walks.
struct VoxelCtx {
    int32_t lk[8];
    float dens[8];
    double surf[8];
    double new_origin[3];
    double norm_origin[3];
    double fs[4];
    double smin, smax;
};

inline void atomic_addf(float* p, float v) {
#pragma omp atomic
    *p += v;
}

struct Counters {
    int n_trav = 0, n_active = 0, n_samples = 0, n_surf = 0;
};

struct Dump {
    int max_dump = 0;
    int n = 0;
    float* t = nullptr;
    float* alpha = nullptr;
    float* weight = nullptr;
    int32_t* voxel = nullptr;
    int32_t* flags = nullptr;
    void push(float tt, float a, float w, const int32_t* v, int fl) {
        if (n < max_dump) {
            if (t) t[n] = tt;
            if (alpha) alpha[n] = a;
            if (weight) weight[n] = w;
            if (voxel) {
                voxel[3 * n] = v[0];
                voxel[3 * n + 1] = v[1];
                voxel[3 * n + 2] = v[2];
            }
            if (flags) flags[n] = fl;
        }
        ++n;
    }
};

// Voxel gates + cubic coefficients.  Returns false if the voxel is skipped.
// surf_trav.cu:212-260.  *link_empty reports the "any link < 0" case (the backward
// treats it differently, :1925-1937).
bool load_voxel(const oracle_grid& g, const oracle_render_options& o, const Ray& r,
                const Dda& d, VoxelCtx& vc, bool* link_empty) {
    *link_empty = false;
    if (!voxel_links(g, d.voxel, vc.lk)) {
        *link_empty = true;
        return false;
    }
    bool all_below = true;
    for (int i = 0; i < 8; ++i) {
        vc.dens[i] = g.density_data[vc.lk[i]];
        if (!(vc.dens[i] < o.sigma_thresh)) all_below = false;
    }
    if (all_below) return false;
    for (int k = 0; k < 3; ++k) {
        vc.new_origin[k] = (double)fmaf(d.t_close, r.dir[k], r.origin[k]);
        vc.norm_origin[k] = vc.new_origin[k] - d.voxel[k];
    }
    for (int i = 0; i < 8; ++i) vc.surf[i] = (double)g.surface_data[vc.lk[i]];
    const double dd[3] = {r.dir[0], r.dir[1], r.dir[2]};
    surface_cubic(vc.surf, vc.norm_origin, dd, vc.fs);
    vc.smin = vc.surf[0];
    vc.smax = vc.surf[0];
    for (int i = 1; i < 8; ++i) {
        vc.smin = std::fmin(vc.smin, vc.surf[i]);
        vc.smax = std::fmax(vc.smax, vc.surf[i]);
    }
    return true;
}

// sample position inside the voxel for root st; false if outside [0,1]^3.
// surf_trav.cu:322-340
bool root_position(const oracle_grid& g, const Ray& r, const Dda& d, const VoxelCtx& vc,
                   double st, float* pos, int32_t* l) {
    for (int k = 0; k < 3; ++k) {
        pos[k] = fmaf((float)st, r.dir[k], (float)vc.new_origin[k]);
        l[k] = std::min(d.voxel[k], g.size[k] - 2);
        pos[k] -= (float)l[k];
    }
    if ((pos[0] < 0) | (pos[0] > 1) | (pos[1] < 0) | (pos[1] > 1) | (pos[2] < 0) | (pos[2] > 1))
        return false;
    return true;
}

// fake-sample geometry shared by fwd/bwd: surf_trav.cu:437-490 / :2471-2532
struct FakeCtx {
    float pos[3];
    double miu, std;
    float dist;
};
void fake_sample_geom(const oracle_grid& g, const oracle_render_options& o, const Ray& r,
                      const Dda& d, const VoxelCtx& vc, FakeCtx& fc) {
    const float tm = (d.t_far + d.t_close) / 2.f;
    for (int k = 0; k < 3; ++k) {
        fc.pos[k] = fmaf(tm, r.dir[k], r.origin[k]);
        const int32_t l = std::min(d.voxel[k], g.size[k] - 2);
        fc.pos[k] -= (float)l;
    }
}
void fake_sample_dist(const oracle_grid& g, const oracle_render_options& o, const VoxelCtx& vc,
                      FakeCtx& fc) {
    const double* s = vc.surf;
    fc.miu = (s[0] + s[1] + s[2] + s[3] + s[4] + s[5] + s[6] + s[7]) / 8;
    double var = 0;
    for (int i = 0; i < 8; ++i) var += sqrd(s[i] - fc.miu);
    // reference: sqrtf(max(1e-9f, <double>/8)) -- the max is taken in double, sqrtf in float
    fc.std = (double)sqrtf((float)std::fmax((double)1e-9f, var / 8));
    if (!o.fake_sample_normalize_surf) fc.std = 1.;
    float ns[8];
    for (int i = 0; i < 8; ++i) ns[i] = (float)(s[i] / fc.std);
    const float fs_s = trilerp(ns, fc.pos);
    float dist = INFINITY;
    for (int i = 0; i < g.level_set_num; ++i) {
        const float dd = fs_s - g.level_set_data[i];
        dist = (std::fabs(dd) < std::fabs(dist)) ? dd : dist;
    }
    fc.dist = dist;
}

// ---------------------------------------------------------------------------------
// Forward: trace_ray_surf_trav, surf_trav.cu:37-562
// ---------------------------------------------------------------------------------
void trace_forward(const oracle_grid& g, const oracle_render_options& o, const Ray& r,
                   int l_dist_max_sample, float* s_alpha, float* s_weight, float* s_ts,
                   float* out_rgb, Counters& cnt, Dump* dump) {
    const int B = g.basis_dim;
    int sample_i = 0;
    int intersect_i = -1;
    if (r.tmin > r.tmax) {
        for (int c = 0; c < 3; ++c) out_rgb[c] = o.background_brightness;
        return;
    }
    float outv[3] = {0.f, 0.f, 0.f};
    float log_transmit = 0.f;
    Dda d;
    dda_init(g, r, d);
    VoxelCtx vc;

    auto shade = [&](const int32_t* lk, const float* pos, float* color) {
        for (int c = 0; c < 3; ++c) {
            float acc = 0.f;
            for (int k = 0; k < B; ++k) {
                float v[8];
                for (int i = 0; i < 8; ++i)
                    v[i] = g.sh_data[(size_t)lk[i] * (3 * B) + c * B + k];
                acc += trilerp(v, pos) * r.sph[k];
            }
            color[c] = acc;
        }
    };

    while (d.t <= r.tmax) {
        dda_step(g, r, d);
        ++cnt.n_trav;
        bool link_empty;
        if (!load_voxel(g, o, r, d, vc, &link_empty)) continue;
        ++cnt.n_active;

        bool vox_has_sample = false, vox_has_surf = false;
        for (int li = 0; li < g.level_set_num; ++li) {
            const double lv = (double)g.level_set_data[li];
            if (lv < vc.smin || lv > vc.smax) continue;
            if (!vox_has_surf) ++cnt.n_surf;
            vox_has_surf = true;
            double st[3] = {-1, -1, -1};
            solve_cubic_vieta(vc.fs[0] - lv, vc.fs[1], vc.fs[2], vc.fs[3], 1e-10, st);
            for (int j = 0; j < 3; ++j) {
                if (st[j] <= 0) continue;
                float pos[3];
                int32_t l[3];
                if (!root_position(g, r, d, vc, st[j], pos, l)) continue;
                vox_has_sample = true;
                if (o.only_outward_intersect) {  // :350-367
                    float sv[8], sg[3];
                    for (int i = 0; i < 8; ++i) sv[i] = g.surface_data[vc.lk[i]];
                    field_grad(sv, pos, sg);
                    const float ndd = -(sg[0] * r.dir[0] + sg[1] * r.dir[1] + sg[2] * r.dir[2]);
                    if (ndd >= 0.f) continue;
                }
                ++intersect_i;
                const float raw_alpha = trilerp(vc.dens, pos);
                if (raw_alpha > o.sigma_thresh) {
                    const float alpha = alpha_act(raw_alpha, o.alpha_activation_type);
                    const float rw = o.truncated_vol_render
                                         ? trunc_rw((float)intersect_i, g.truncated_vol_render_a,
                                                    o.trunc_vol_weight_min)
                                         : 1.f;
                    const float rwalpha = alpha * rw;
                    float color[3];
                    shade(vc.lk, pos, color);
                    const float pcnt = -1 * logf(1 - rwalpha);
                    const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
                    log_transmit -= pcnt;
                    for (int c = 0; c < 3; ++c) outv[c] += weight * fmaxf(color[c] + 0.5f, 0.f);
                    ++cnt.n_samples;
                    const float ts = d.t_close + (float)st[j];
                    if (dump) dump->push(ts, rwalpha, weight, d.voxel, 0);
                    if (s_weight != nullptr && sample_i < l_dist_max_sample) {
                        s_alpha[sample_i] = rwalpha;
                        s_weight[sample_i] = weight;
                        // t_close + st[j] is float + double -> double, stored as float
                        s_ts[sample_i] = (float)((double)d.t_close + st[j]);
                        sample_i += 1;
                    }
                }
            }
        }

        // fake sample, :423-541
        if (o.surf_fake_sample && !vox_has_sample && (!o.limited_fake_sample || vox_has_surf)) {
            if ((d.t_far - d.t_close) > o.surf_fake_sample_min_vox_len) {
                FakeCtx fc;
                fake_sample_geom(g, o, r, d, vc, fc);
                float alpha = trilerp(vc.dens, fc.pos);
                if (alpha > o.sigma_thresh) {
                    alpha = alpha_act(alpha, o.alpha_activation_type);
                    fake_sample_dist(g, o, vc, fc);
                    // alpha * _EXP(-.5 * _SQR(dist/std)) : -.5 is a double literal
                    alpha = alpha * expf((float)(-.5 * (double)sqrf(fc.dist / g.fake_sample_std)));
                    const float rw = o.truncated_vol_render
                                         ? trunc_rw((float)intersect_i, g.truncated_vol_render_a,
                                                    o.trunc_vol_weight_min)
                                         : 1.f;
                    alpha = alpha * rw;
                    float color[3];
                    shade(vc.lk, fc.pos, color);
                    const float pcnt = -1 * logf(1 - alpha);
                    const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
                    log_transmit -= pcnt;
                    for (int c = 0; c < 3; ++c) outv[c] += weight * fmaxf(color[c] + 0.5f, 0.f);
                    ++cnt.n_samples;
                    const float tm = (d.t_far + d.t_close) / 2.f;
                    if (dump) dump->push(tm, alpha, weight, d.voxel, 1);
                    if (s_weight != nullptr && sample_i < l_dist_max_sample && o.fake_sample_l_dist) {
                        s_alpha[sample_i] = alpha;
                        s_weight[sample_i] = weight;
                        s_ts[sample_i] = tm;
                        sample_i += 1;
                    }
                }
            }
        }

        if (expf(log_transmit) < o.stop_thresh) {  // :544-547
            log_transmit = -1e3f;
            break;
        }
    }
    const float bg = expf(log_transmit) * o.background_brightness;  // :553-555
    for (int c = 0; c < 3; ++c) out_rgb[c] = outv[c] + bg;
}

// ---------------------------------------------------------------------------------
// Backward: trace_ray_surf_trav_backward, surf_trav.cu:1710-2911
// ---------------------------------------------------------------------------------
struct LossParams {
    float sparsity_loss;
    float lambda_l_dist, lambda_l_entropy;
    bool no_norm_weight_l_entropy;
    float lambda_l_dist_a, lambda_l_entropy_a, lambda_l_samp_dist, lambda_l_di, l_di_alpha_thresh;
    float surf_sparse_alpha_thresh, lambda_inplace_surf_sparse, lambda_inwards_norm_loss;
    float lambda_conv_mode_samp;
    int l_dist_max_sample;
};

void scatter8(float* out, uint8_t* mask, const int32_t* lk, const float* w) {
    if (!out) return;
    for (int i = 0; i < 8; ++i) {
        atomic_addf(&out[lk[i]], w[i]);
        if (mask) mask[lk[i]] = 1;
    }
}

void trace_backward(const oracle_grid& g, const oracle_render_options& o, const Ray& r,
                    const float* grad_out, const float* color_cache, const LossParams& L,
                    const float* s_alpha, const float* s_weight, const float* s_ts,
                    const oracle_grads& G) {
    const int B = g.basis_dim;
    const int D = 3 * B;
    const int M = L.l_dist_max_sample;
    int sample_i = 0;
    int intersect_i = -1;

    // prologue over the sample cache, :1756-1793
    float sample_alpha_sum = 0.f, sample_weight_sum = 0.f;
    for (int i = 0; i < M; ++i) {
        sample_alpha_sum += s_alpha[i];
        sample_weight_sum += s_weight[i];
    }
    sample_alpha_sum = (float)std::fmax((double)sample_alpha_sum, 1e-8);
    sample_weight_sum = (float)std::fmax((double)sample_weight_sum, 1e-8);
    float Den_Dasum = 0.f, Den_Dwsum = 0.f, sample_t_mean = 0.f;
    for (int i = 0; i < M; ++i) {
        Den_Dasum += s_alpha[i] * (log_ratio(s_alpha[i], sample_alpha_sum) + 1.f) / sqrf(sample_alpha_sum);
        Den_Dwsum += s_weight[i] * (log_ratio(s_weight[i], sample_weight_sum) + 1.f) / sqrf(sample_weight_sum);
        sample_t_mean += s_weight[i] / sample_weight_sum * s_ts[i];
    }
    int valid_sample_n = 0;
    float shared_Dmeant_sign = 0.f;
    for (int i = 0; i < M; ++i) {
        if (s_ts[i] > 0.f) {
            valid_sample_n++;
            shared_Dmeant_sign +=
                (sample_t_mean > s_ts[i]) ? 1.f : ((sample_t_mean < s_ts[i]) ? -1.f : 0.f);
        }
    }
    int max_sample_id = 0;
    float max_weight = 0.f;
    for (int i = 0; i < M; ++i) {
        if (s_weight[i] > max_weight) {
            max_weight = s_weight[i];
            max_sample_id = i;
        }
    }
    // reads of the cache at a possibly out-of-range index when M == 0 (non-fused
    // backward) are guarded here; the reference only dereferences them when the
    // matching lambda is > 0, which never happens with M == 0.
    auto ts_at = [&](int i) { return (i < M) ? s_ts[i] : 0.f; };
    auto al_at = [&](int i) { return (i < M) ? s_alpha[i] : 0.f; };
    auto we_at = [&](int i) { return (i < M) ? s_weight[i] : 0.f; };

    float accum = fmaf(color_cache[0], grad_out[0],
                       fmaf(color_cache[1], grad_out[1], color_cache[2] * grad_out[2]));
    if (r.tmin > r.tmax) return;

    float log_transmit = 0.f;
    Dda d;
    dda_init(g, r, d);
    VoxelCtx vc;

    // colour of a sample and everything derived from it (the per-lane block
    // :2089-2139): returns total_color; fills curr_grad_color[D], lane_color[D].
    auto shade_bwd = [&](const int32_t* lk, const float* pos, float weight, float* cg,
                         float* total_color_out) {
        float tc[3], in01[3];
        for (int c = 0; c < 3; ++c) {
            float acc = 0.f;
            for (int k = 0; k < B; ++k) {
                float v[8];
                for (int i = 0; i < 8; ++i) v[i] = g.sh_data[(size_t)lk[i] * D + c * B + k];
                acc += trilerp(v, pos) * r.sph[k];
            }
            const float lct = acc + 0.5f;
            const float t = fmaxf(lct, 0.f);
            in01[c] = (t == lct) ? 1.f : 0.f;
            tc[c] = t * grad_out[c];
        }
        // (c0 + c2) + c1, :2112-2115
        *total_color_out = (tc[0] + tc[2]) + tc[1];
        for (int c = 0; c < 3; ++c) {
            const float grad_common = weight * in01[c] * grad_out[c];
            for (int k = 0; k < B; ++k) cg[c * B + k] = r.sph[k] * grad_common;
        }
    };
    auto scatter_sh = [&](const int32_t* lk, const float* pos, const float* cg) {
        if (!G.grad_sh_out) return;
        for (int e = 0; e < D; ++e) {
            float w[8];
            corner_weights(pos, cg[e], w);
            for (int i = 0; i < 8; ++i) atomic_addf(&G.grad_sh_out[(size_t)lk[i] * D + e], w[i]);
        }
    };
    // l_dist / l_entropy contributions to d/d(rwalpha), :2141-2210 and :2602-2666
    auto aux_rwalpha = [&](float rwalpha, float& curr) {
        const float denom = fminf(rwalpha - 1.f, -1e-8f);
        if (L.lambda_l_dist > 0.f) {
            float Dl = 0.f;
            float log_Tk = log_transmit;
            for (int k = sample_i; k < valid_sample_n; ++k) {
                float Dwk = 0.f;
                for (int j = 0; j < M; ++j) Dwk += s_weight[j] * std::fabs(ts_at(k) - s_ts[j]);
                if (k == sample_i) {
                    Dl += Dwk * expf(log_transmit);
                } else {
                    log_Tk += logf((float)std::fmax((double)(1.f - al_at(k - 1)), 1e-8));
                    Dl += Dwk * expf(log_Tk) * al_at(k) / denom;
                }
            }
            curr += L.lambda_l_dist * Dl;
        }
        if (L.lambda_l_entropy > 0.f) {
            auto Den_Dw = [&](int k) -> float {
                if (L.no_norm_weight_l_entropy)
                    return -(logf((float)std::fmax((double)we_at(k), 1e-8)) + 1.f);
                return -(log_ratio(we_at(k), sample_weight_sum) + 1.f) / sample_weight_sum + Den_Dwsum;
            };
            float Den_Dai = Den_Dw(sample_i) * expf(log_transmit);
            float log_Tk = log_transmit;
            for (int k = sample_i + 1; k < valid_sample_n; ++k) {
                const float Dwk = Den_Dw(k);
                log_Tk += logf((float)std::fmax((double)(1.f - al_at(k - 1)), 1e-8));
                Den_Dai += Dwk * expf(log_Tk) * al_at(k) / denom;
            }
            curr += L.lambda_l_entropy * Den_Dai;
        }
    };
    auto inwards_norm = [&](const float* pos, float& curr) {  // :2302-2323
        float sv[8], sg[3];
        for (int i = 0; i < 8; ++i) sv[i] = g.surface_data[vc.lk[i]];
        field_grad(sv, pos, sg);
        const float n = (float)std::fmax((double)sqrtf(sqrf(sg[0]) + sqrf(sg[1]) + sqrf(sg[2])), 1e-8);
        const float nd = (-sg[0] / n) * r.dir[0] + (-sg[1] / n) * r.dir[1] + (-sg[2] / n) * r.dir[2];
        if (nd > 0.f) curr += L.lambda_inwards_norm_loss * sqrf(nd);
    };

    while (d.t <= r.tmax) {
        dda_step(g, r, d);
        bool link_empty;
        if (!load_voxel(g, o, r, d, vc, &link_empty)) {
            if (link_empty) d.t += o.step_size;  // :1935-1936 (forward has a bare continue)
            continue;
        }
        const double fs0_original = vc.fs[0];
        bool vox_has_sample = false, vox_has_surf = false;

        for (int li = 0; li < g.level_set_num; ++li) {
            const double lv = (double)g.level_set_data[li];
            if (lv < vc.smin || lv > vc.smax) continue;
            vox_has_surf = true;
            double fs[4] = {fs0_original - lv, vc.fs[1], vc.fs[2], vc.fs[3]};
            double st[3] = {-1, -1, -1};
            const int root_type = solve_cubic_vieta(fs[0], fs[1], fs[2], fs[3], 1e-10, st);
            for (int st_id = 0; st_id < 3; ++st_id) {
                if (st[st_id] <= 0) continue;
                float pos[3];
                int32_t l[3];
                if (!root_position(g, r, d, vc, st[st_id], pos, l)) continue;
                vox_has_sample = true;
                if (o.only_outward_intersect) {
                    float sv[8], sg[3];
                    for (int i = 0; i < 8; ++i) sv[i] = g.surface_data[vc.lk[i]];
                    field_grad(sv, pos, sg);
                    const float ndd = -(sg[0] * r.dir[0] + sg[1] * r.dir[1] + sg[2] * r.dir[2]);
                    if (ndd >= 0.f) continue;
                }
                ++intersect_i;
                const float raw_alpha = trilerp(vc.dens, pos);
                if (!(raw_alpha > o.sigma_thresh)) continue;
                const float alpha = alpha_act(raw_alpha, o.alpha_activation_type);
                const float trunc_reweight =
                    o.truncated_vol_render ? trunc_rw((float)intersect_i, g.truncated_vol_render_a,
                                                      o.trunc_vol_weight_min)
                                           : 1.f;
                const float rwalpha = alpha * trunc_reweight;
                const float pcnt = -logf((float)std::fmax((double)(1.f - rwalpha), 1e-8));
                const float weight = expf(log_transmit) * (1.f - expf(-pcnt));

                float cg[3 * kMaxBasis];
                float total_color;
                shade_bwd(vc.lk, pos, weight, cg, &total_color);
                accum -= weight * total_color;
                float curr_grad_rwalpha =
                    accum / fminf(rwalpha - 1.f, -1e-8f) + total_color * expf(log_transmit);
                aux_rwalpha(rwalpha, curr_grad_rwalpha);
                log_transmit -= pcnt;

                scatter_sh(vc.lk, pos, cg);
                float grad_xyz[3] = {0, 0, 0};
                if (!o.no_surf_grad_from_sh) {  // :2250-2264
                    for (int e = 0; e < D; ++e) {
                        float v[8];
                        for (int i = 0; i < 8; ++i) v[i] = g.sh_data[(size_t)vc.lk[i] * D + e];
                        trilerp_dpos(v, pos, cg[e], grad_xyz);
                    }
                }
                // lane-0 block, :2266-2448
                if (L.lambda_l_dist_a > 0.f) {
                    float ga = 0.f;
                    for (int j = 0; j < M; ++j) ga += s_alpha[j] * std::fabs(ts_at(sample_i) - s_ts[j]);
                    curr_grad_rwalpha += L.lambda_l_dist_a * ga;
                }
                if (L.lambda_l_entropy_a > 0.f) {
                    const float Den_Dai = -(log_ratio(al_at(sample_i), sample_alpha_sum) + 1.f) / sample_alpha_sum;
                    curr_grad_rwalpha += L.lambda_l_entropy_a * (Den_Dai + Den_Dasum);
                }
                if (L.sparsity_loss > 0.f && raw_alpha > 0.f) {
                    curr_grad_rwalpha = sparsity_term(curr_grad_rwalpha, L.sparsity_loss, alpha, weight, sample_weight_sum);
                }
                if (L.lambda_inwards_norm_loss > 0.f) inwards_norm(pos, curr_grad_rwalpha);

                const float curr_grad_alpha = curr_grad_rwalpha * trunc_reweight;
                float curr_grad_raw_alpha = curr_grad_alpha * alpha_act_grad(alpha, o.alpha_activation_type);
                {
                    float w[8];
                    corner_weights(pos, curr_grad_raw_alpha, w);
                    scatter8(G.grad_density_out, G.mask_out, vc.lk, w);
                }
                if (L.lambda_l_di > 0.f && alpha < L.l_di_alpha_thresh)
                    curr_grad_raw_alpha += L.lambda_l_di * (-1.f) * alpha_act_grad(alpha, o.alpha_activation_type);
                trilerp_dpos(vc.dens, pos, curr_grad_raw_alpha, grad_xyz);
                float grad_st = grad_xyz[0] * r.dir[0] + grad_xyz[1] * r.dir[1] + grad_xyz[2] * r.dir[2];

                if (L.lambda_l_dist > 0.f || L.lambda_l_dist_a > 0.f) {
                    float gw = 0.f, ga = 0.f;
                    for (int j = 0; j < M; ++j) {
                        const float sgn = (ts_at(sample_i) > s_ts[j]) ? 1.f
                                          : (ts_at(sample_i) < s_ts[j]) ? -1.f : 0.f;
                        gw += sgn * we_at(sample_i) * s_weight[j];
                        ga += sgn * al_at(sample_i) * s_alpha[j];
                    }
                    grad_st += L.lambda_l_dist * gw + L.lambda_l_dist_a * ga;
                }
                if (L.lambda_l_samp_dist > 0.f) {
                    const float sgn = (sample_t_mean > ts_at(sample_i)) ? 1.f
                                      : (sample_t_mean < ts_at(sample_i)) ? -1.f : 0.f;
                    grad_st += L.lambda_l_samp_dist *
                               (shared_Dmeant_sign * al_at(sample_i) / sample_weight_sum + sgn * (-1.f));
                }
                if (L.lambda_conv_mode_samp > 0.f && trunc_reweight > o.trunc_vol_weight_min) {
                    const float gcm = (ts_at(sample_i) > ts_at(max_sample_id)) ? 1.f
                                      : (ts_at(sample_i) < ts_at(max_sample_id)) ? -1.f : 0.f;
                    grad_st += L.lambda_conv_mode_samp * gcm;
                }

                float grad_fs[4] = {grad_st, grad_st, grad_st, grad_st};
                cubic_root_grad_vieta(root_type, st_id, fs, grad_fs);
                float gsurf[8];
                const float nof[3] = {(float)vc.norm_origin[0], (float)vc.norm_origin[1],
                                      (float)vc.norm_origin[2]};
                surface_cubic_grad(nof, r.dir, grad_fs, gsurf);
                scatter8(G.grad_surface_out, G.mask_out, vc.lk, gsurf);

                if (alpha < L.surf_sparse_alpha_thresh) {  // :2436-2445
                    float w[8];
                    corner_weights(pos, L.lambda_inplace_surf_sparse, w);
                    scatter8(G.grad_surface_out, G.mask_out, vc.lk, w);
                }
                if (sample_i < M - 1) sample_i += 1;
            }
        }

        // fake sample gradient, :2460-2866
        if (o.surf_fake_sample && !vox_has_sample && (!o.limited_fake_sample || vox_has_surf)) {
            if ((d.t_far - d.t_close) > o.surf_fake_sample_min_vox_len) {
                FakeCtx fc;
                fake_sample_geom(g, o, r, d, vc, fc);
                const float raw_alpha = trilerp(vc.dens, fc.pos);
                if (raw_alpha > o.sigma_thresh) {
                    const float alpha = alpha_act(raw_alpha, o.alpha_activation_type);
                    fake_sample_dist(g, o, vc, fc);
                    const float reweight =
                        expf((float)(-.5 * (double)sqrf(fc.dist / g.fake_sample_std)));
                    const float trunc_reweight =
                        o.truncated_vol_render ? trunc_rw((float)intersect_i, g.truncated_vol_render_a,
                                                          o.trunc_vol_weight_min)
                                               : 1.f;
                    const float rw_alpha = alpha * reweight * trunc_reweight;
                    const float pcnt = -1 * logf((float)std::fmax((double)(1.f - rw_alpha), 1e-8));
                    const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
                    float cg[3 * kMaxBasis];
                    float total_color;
                    shade_bwd(vc.lk, fc.pos, weight, cg, &total_color);
                    accum -= weight * total_color;
                    float curr_grad_rwalpha =
                        accum / fminf(rw_alpha - 1.f, -1e-8f) + total_color * expf(log_transmit);
                    if (o.fake_sample_l_dist) aux_rwalpha(rw_alpha, curr_grad_rwalpha);
                    log_transmit -= pcnt;
                    scatter_sh(vc.lk, fc.pos, cg);

                    if (o.fake_sample_l_dist) {
                        if (L.lambda_l_dist_a > 0.f) {
                            float ga = 0.f;
                            for (int j = 0; j < M; ++j)
                                ga += s_alpha[j] * std::fabs(ts_at(sample_i) - s_ts[j]);
                            curr_grad_rwalpha += L.lambda_l_dist_a * ga;
                        }
                        if (L.lambda_l_entropy_a > 0.f) {
                            const float Den_Dai = -(log_ratio(al_at(sample_i), sample_alpha_sum) + 1.f) / sample_alpha_sum;
                            curr_grad_rwalpha += L.lambda_l_entropy_a * (Den_Dai + Den_Dasum);
                        }
                        if (sample_i < M - 1) sample_i += 1;
                    }
                    if (L.sparsity_loss > 0.f && raw_alpha > 0.f) {
                        curr_grad_rwalpha = sparsity_term(curr_grad_rwalpha, L.sparsity_loss, rw_alpha, weight, sample_weight_sum);
                    }
                    if (L.lambda_inwards_norm_loss > 0.f) inwards_norm(fc.pos, curr_grad_rwalpha);

                    const float curr_grad_alpha = curr_grad_rwalpha * reweight * trunc_reweight;
                    const float curr_grad_raw_alpha =
                        curr_grad_alpha * alpha_act_grad(alpha, o.alpha_activation_type);
                    {
                        float w[8];
                        corner_weights(fc.pos, curr_grad_raw_alpha, w);
                        scatter8(G.grad_density_out, G.mask_out, vc.lk, w);
                    }
                    // gradient to the surface through the gaussian re-weighting, :2792-2851
                    const float grad_fake_dist =
                        curr_grad_rwalpha * (-alpha * trunc_reweight * fc.dist * reweight / sqrf(g.fake_sample_std));
                    float grad_ns[8];
                    corner_weights(fc.pos, grad_fake_dist, grad_ns);
                    float gsurf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (!o.fake_sample_normalize_surf) {
                        for (int ks = 0; ks < 8; ++ks) gsurf[ks] = grad_ns[ks];
                    } else {
                        const double std3 = cubd(fc.std);
                        for (int ks = 0; ks < 8; ++ks) {
                            for (int kn = 0; kn < 8; ++kn) {
                                double jac = vc.surf[kn] * (fc.miu - vc.surf[ks]) / 8.f / std3;
                                if (ks == kn) jac += 1.f / fc.std;
                                gsurf[ks] = (float)((double)gsurf[ks] + (double)grad_ns[kn] * jac);
                            }
                        }
                    }
                    scatter8(G.grad_surface_out, G.mask_out, vc.lk, gsurf);
                    if (G.grad_fake_sample_std_out != nullptr) {
                        const float gstd = curr_grad_rwalpha * alpha * sqrf(fc.dist) * reweight *
                                           trunc_reweight /
                                           (g.fake_sample_std * g.fake_sample_std * g.fake_sample_std);
                        atomic_addf(G.grad_fake_sample_std_out, gstd);
                    }
                }
            }
        }
        if (expf(log_transmit) < o.stop_thresh) break;  // :2893-2895
    }
}

int check_common(const oracle_grid* g, const oracle_rays* r, const oracle_render_options* o) {
    if (!g || !r || !o) return -1;
    if (g->basis_type != 1) return -2;
    if (g->background_nlayers != 0) return -2;
    if (!(g->basis_dim == 1 || g->basis_dim == 4 || g->basis_dim == 9 || g->basis_dim == 16)) return -1;
    if (!g->links || !g->density_data || !g->sh_data || !g->surface_data) return -1;
    if (g->level_set_num > 0 && !g->level_set_data) return -1;
    return 0;
}

void attach_stats(const oracle_stats* st, int64_t q, const Counters& c, const Dump& dp) {
    if (!st) return;
    if (st->n_traversed) st->n_traversed[q] = c.n_trav;
    if (st->n_active) st->n_active[q] = c.n_active;
    if (st->n_samples) st->n_samples[q] = c.n_samples;
    if (st->n_surf) st->n_surf[q] = c.n_surf;
}

Dump make_dump(const oracle_stats* st, int64_t q) {
    Dump d;
    if (st && st->max_dump > 0) {
        d.max_dump = st->max_dump;
        const size_t off = (size_t)q * st->max_dump;
        d.t = st->dump_t ? st->dump_t + off : nullptr;
        d.alpha = st->dump_alpha ? st->dump_alpha + off : nullptr;
        d.weight = st->dump_weight ? st->dump_weight + off : nullptr;
        d.voxel = st->dump_voxel ? st->dump_voxel + 3 * off : nullptr;
        d.flags = st->dump_flags ? st->dump_flags + off : nullptr;
    }
    return d;
}

// ---------------------------------------------------------------------------------
// Legacy `surface` backend (render_lerp_kernel_surface.cu): the caller supplies each
// ray's voxel list; coefficients are taken about the ray ORIGIN (global t), the solver
// is the f/g/h form, alpha is always a sigmoid, one level set, no sigma gate.
// ---------------------------------------------------------------------------------
constexpr int CUBIC_ONE_R = 204;  // data_spec.hpp:24-31 (triple root)

// render_util.cuh:753-787
void surface_cubic_global(const double* s, const double* o, const double* d, const int32_t* l,
                          double* f) {
    double lo[3], hi[3];  // trilinear factors at t = 0: corner bit 0 / bit 1
    for (int k = 0; k < 3; ++k) {
        lo[k] = 1 - o[k] + l[k];
        hi[k] = o[k] - l[k];
    }
    double c[2], dd[2], e[2];
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

inline double cond_cbrt(double x) { return (x >= 0) ? cbrt(x) : -1. * cbrt(-x); }  // util.hpp:33
inline double d_cond_cbrt(double x) {                                             // util.hpp:36
    return (x >= 0) ? ((x > 0) ? (1. / 3. / sqrd(cbrt(x))) : 1.) : cbrt(-x) / -3. / x;
}

// render_util.cuh:940-1102
int solve_cubic_legacy(double f0, double f1, double f2, double f3, double eps, double* st) {
    auto tiny = [&](double x) { return std::abs(x) < eps; };
    if (tiny(f3)) {
        if (tiny(f2)) {
            if (tiny(f1)) return CUBIC_NO_ROOT;
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
            if (tiny(st[0] - st[1])) {
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
    if (tiny(f) & tiny(g) & tiny(h)) {
        st[0] = cond_cbrt(d);
        return CUBIC_ONE_R;
    }
    if (h <= 0) {
        if (h == 0) return CUBIC_NO_ROOT;
        const double i = sqrt((sqrd(g) / 4.) - h);
        const double j = cbrt(i);
        const double k = acos(-(g / (2 * i)));
        const double M = cos(k / 3.);
        const double N = sqrt(3) * sin(k / 3.);
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

// render_util.cuh:1418-1702: m[k] = d t / d f_k of root st_id (0 where the reference assigns 0)
void cubic_root_grad_legacy_mult(int type, int st_id, const double* fs, double* m) {
    for (int k = 0; k < 4; ++k) m[k] = 0.;
    if (type == CUBIC_LINEAR) {
        m[0] = -1. / fs[1];
        m[1] = fs[0] / sqrd(fs[1]);
        return;
    }
    if (type == CUBIC_POLY_ONE_R || type == CUBIC_POLY) {
        const double D = sqrd(fs[1]) - 4. * fs[2] * fs[0];
        const double sD = sqrt(D);
        // smaller root (st_id 0 of the two-root case) takes the "+sqrt" branch of :1447-1451
        const bool small = (type == CUBIC_POLY) && (st_id == 0);
        const double sgn = small ? -1. : 1.;
        const double dt_dD = sgn / (4. * fs[2] * sD);
        const double den = (type == CUBIC_POLY) ? 2. * sqrd(fs[2]) : 4. * sqrd(fs[2]);
        m[0] = -sgn / sD;
        m[1] = (-1) / (2 * fs[2]) + (dt_dD * 2 * fs[1]);
        m[2] = (fs[1] - sgn * sD) / den + (dt_dD * (-4) * fs[0]);
        return;
    }
    const double b = fs[2] / fs[3], c = fs[1] / fs[3], d = fs[0] / fs[3];
    // d(b,c,d)/d(f0..f3)
    const double Db[4] = {0., 0., 1. / fs[3], (-fs[2]) / fs[3] / fs[3]};
    const double Dc[4] = {0., 1. / fs[3], 0., (-fs[1]) / fs[3] / fs[3]};
    const double Dd[4] = {1. / fs[3], 0., 0., (-fs[0]) / fs[3] / fs[3]};
    if (type == CUBIC_ONE_R) {
        const double dc = d_cond_cbrt(d);
        m[0] = dc * Dd[0];
        m[3] = dc * Dd[3];
        return;
    }
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
        for (int q = 0; q < 4; ++q)
            m[q] = Dst_Dj * (Dj_Dg * Dg[q] + Dj_Dh * Dh[q]) + Dst_Dk * (Dk_Dg * Dg[q] + Dk_Dh * Dh[q]) +
                   Dst_Db * Db[q];
    } else {  // CUBIC_ONE_R_
        const double R = -(g / 2.) + sqrt(h);
        const double T = -(g / 2.) - sqrt(h);
        const double dR = d_cond_cbrt(R), dT = d_cond_cbrt(T);
        const double DR_Dh = 1. / (2. * sqrt(h)), DT_Dh = -1. / (2. * sqrt(h));
        for (int q = 0; q < 4; ++q)
            m[q] = dR * (-0.5 * Dg[q] + DR_Dh * Dh[q]) + dT * (-0.5 * Dg[q] + DT_Dh * Dh[q]) +
                   Dst_Db * Db[q];
    }
}

// grad[k] holds dL/dt on entry, dL/df_k on exit (float32 products, as the reference)
void cubic_root_grad_legacy(int type, int st_id, const double* fs, float* grad) {
    double m[4];
    cubic_root_grad_legacy_mult(type, st_id, fs, m);
    for (int k = 0; k < 4; ++k) grad[k] = (m[k] == 0.) ? 0.f : grad[k] * (float)m[k];
}

// render_util.cuh:1705-1775: d f_k / d s_corner of surface_cubic_global, float32.
// Each corner's trilinear weight along the ray is a product of three linear factors
// (A_x + B_x t)(A_y + B_y t)(A_z + B_z t); f_k picks the t^k coefficient.
template <typename T>
void surface_cubic_global_grad_t(const float* o, const float* d, const int32_t* l, const T* gf,
                                 float* gs) {
    for (int cr = 0; cr < 8; ++cr) {
        T A[3], Bc[3];
        for (int k = 0; k < 3; ++k) {
            const bool hi = (cr >> (2 - k)) & 1;
            A[k] = hi ? (-(T)l[k] + (T)o[k]) : ((T)l[k] - (T)o[k] + 1);
            Bc[k] = hi ? (T)d[k] : -(T)d[k];
        }
        const T t0 = A[0] * A[1] * A[2];
        const T t1 = Bc[0] * A[1] * A[2] + A[0] * (Bc[1] * A[2] + Bc[2] * A[1]);
        const T t2 = Bc[0] * (Bc[1] * A[2] + Bc[2] * A[1]) + Bc[1] * Bc[2] * A[0];
        const T t3 = Bc[0] * Bc[1] * Bc[2];
        gs[cr] = (float)((T)gf[0] * t0 + (T)gf[1] * t1 + (T)gf[2] * t2 + (T)gf[3] * t3);
    }
}

// Diagnostic switch (tests only): carry dL/df_k and the expression above in float64.  The
// reference works in float32, where the four terms cancel to ~1e-4 of their size for
// far-away origins.
int g_legacy_f64_surface_grad = 0;

void legacy_surface_grad(int type, int st_id, const double* fs, float grad_st, const float* o,
                         const float* d, const int32_t* l, float* gs) {
    if (g_legacy_f64_surface_grad) {
        double m[4];
        cubic_root_grad_legacy_mult(type, st_id, fs, m);
        for (int k = 0; k < 4; ++k) m[k] *= (double)grad_st;
        surface_cubic_global_grad_t<double>(o, d, l, m, gs);
    } else {
        float gf[4] = {grad_st, grad_st, grad_st, grad_st};
        cubic_root_grad_legacy(type, st_id, fs, gf);
        surface_cubic_global_grad_t<float>(o, d, l, gf, gs);
    }
}

inline float sigmoid_ref(float x) { return (float)(1. / (1. + (double)expf(-x))); }  // util.hpp:18

struct LegacyVoxel {
    int32_t l[3];
    int32_t lk[8];
    double fs[4];
    bool has_level;
};

// gates + coefficients of one listed voxel; surface.cu:66-126 == :385-416
bool legacy_load(const oracle_grid& g, const Ray& r, const int32_t* vox, LegacyVoxel& v) {
    for (int k = 0; k < 3; ++k) v.l[k] = vox[k];
    if ((v.l[0] + 1 >= g.size[0]) || (v.l[1] + 1 >= g.size[1]) || (v.l[2] + 1 >= g.size[2])) return false;
    if (v.l[0] < 0 || v.l[1] < 0 || v.l[2] < 0) return false;  // (the reference would read out of bounds)
    if (!voxel_links(g, v.l, v.lk)) return false;
    double s[8];
    double mn = 0, mx = 0;
    for (int i = 0; i < 8; ++i) {
        s[i] = (double)g.surface_data[v.lk[i]];
        mn = (i == 0) ? s[i] : std::fmin(mn, s[i]);
        mx = (i == 0) ? s[i] : std::fmax(mx, s[i]);
    }
    const double od[3] = {r.origin[0], r.origin[1], r.origin[2]};
    const double dd[3] = {r.dir[0], r.dir[1], r.dir[2]};
    surface_cubic_global(s, od, dd, v.l, v.fs);
    const double lv = g.level_set_data[0];
    v.has_level = !((lv < mn) || (lv > mx));
    v.fs[0] -= lv;
    return true;
}

bool legacy_position(const oracle_grid& g, const Ray& r, const LegacyVoxel& v, double st, float* pos) {
    for (int k = 0; k < 3; ++k) {
        pos[k] = fmaf((float)st, r.dir[k], r.origin[k]);
        pos[k] -= (float)std::min(v.l[k], g.size[k] - 2);
    }
    return !((pos[0] < 0) | (pos[0] > 1) | (pos[1] < 0) | (pos[1] > 1) | (pos[2] < 0) | (pos[2] > 1));
}

void legacy_shade(const oracle_grid& g, const Ray& r, const int32_t* lk, const float* pos, float* color) {
    const int B = g.basis_dim;
    for (int c = 0; c < 3; ++c) {
        float acc = 0.f;
        for (int k = 0; k < B; ++k) {
            float v[8];
            for (int i = 0; i < 8; ++i) v[i] = g.sh_data[(size_t)lk[i] * (3 * B) + c * B + k];
            acc += trilerp(v, pos) * r.sph[k];
        }
        color[c] = acc;
    }
}

// surface.cu:35-234
void legacy_forward(const oracle_grid& g, const oracle_render_options& o, const Ray& r,
                    const int32_t* voxel_ls, int32_t start, int32_t num, float* out_rgb) {
    const float bg = (g.background_nlayers == 0) ? o.background_brightness : 0.f;
    if (r.tmin > r.tmax) {
        for (int c = 0; c < 3; ++c) out_rgb[c] = bg;
        return;
    }
    float outv[3] = {0.f, 0.f, 0.f};
    float log_transmit = 0.f;
    for (int vi = 0; vi < num; ++vi) {
        LegacyVoxel v;
        if (!legacy_load(g, r, voxel_ls + 3 * ((size_t)start + vi), v)) continue;
        if (v.has_level) {
            double st[3] = {-1, -1, -1};
            solve_cubic_legacy(v.fs[0], v.fs[1], v.fs[2], v.fs[3], 1e-10, st);
            for (int j = 0; j < 3; ++j) {
                if (st[j] <= 0) continue;
                float pos[3];
                if (!legacy_position(g, r, v, st[j], pos)) continue;
                float dens[8];
                for (int i = 0; i < 8; ++i) dens[i] = g.density_data[v.lk[i]];
                const float alpha = sigmoid_ref(trilerp(dens, pos));
                float color[3];
                legacy_shade(g, r, v.lk, pos, color);
                const float pcnt = -1 * logf(1 - alpha);
                const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
                log_transmit -= pcnt;
                for (int c = 0; c < 3; ++c) outv[c] += weight * fmaxf(color[c] + 0.5f, 0.f);
            }
        }
        if (expf(log_transmit) < o.stop_thresh) {
            log_transmit = -1e3f;
            break;
        }
    }
    if (g.background_nlayers == 0)
        for (int c = 0; c < 3; ++c) outv[c] += expf(log_transmit) * o.background_brightness;
    for (int c = 0; c < 3; ++c) out_rgb[c] = outv[c];
}

// surface.cu:341-645
void legacy_backward(const oracle_grid& g, const oracle_render_options& o, const Ray& r,
                     const int32_t* voxel_ls, int32_t start, int32_t num, const float* gout,
                     const float* color_cache, float sparsity_loss, const oracle_grads& G) {
    const int B = g.basis_dim, D = 3 * B;
    float accum = fmaf(color_cache[0], gout[0], fmaf(color_cache[1], gout[1], color_cache[2] * gout[2]));
    if (r.tmin > r.tmax) return;
    float log_transmit = 0.f;
    for (int vi = 0; vi < num; ++vi) {
        LegacyVoxel v;
        if (!legacy_load(g, r, voxel_ls + 3 * ((size_t)start + vi), v)) continue;
        if (v.has_level) {
            double st[3] = {-1, -1, -1};
            const int type = solve_cubic_legacy(v.fs[0], v.fs[1], v.fs[2], v.fs[3], 1e-10, st);
            for (int j = 0; j < 3; ++j) {
                if (st[j] <= 0) continue;
                float pos[3];
                if (!legacy_position(g, r, v, st[j], pos)) continue;
                float dens[8];
                for (int i = 0; i < 8; ++i) dens[i] = g.density_data[v.lk[i]];
                const float raw_alpha = trilerp(dens, pos);
                const float alpha = sigmoid_ref(raw_alpha);
                float color[3];
                legacy_shade(g, r, v.lk, pos, color);
                const float pcnt = -1 * logf(1 - alpha);
                const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
                float total_ch[3], in01[3];
                for (int c = 0; c < 3; ++c) {
                    const float lc = color[c] + 0.5f;
                    const float cl = fmaxf(lc, 0.f);
                    in01[c] = (cl == lc) ? 1.f : 0.f;
                    total_ch[c] = cl * gout[c];
                }
                const float total_color = (total_ch[0] + total_ch[2]) + total_ch[1];  // :487-489
                accum -= weight * total_color;
                float curr_grad_alpha = accum / (alpha - 1) + total_color * expf(log_transmit);
                log_transmit -= pcnt;
                if (sparsity_loss > 0.f) curr_grad_alpha += sparsity_loss * (4 * alpha / (1 + 2 * (alpha * alpha)));
                // SH scatter (:508-513) and d/dpos through the colour (:516-528)
                float grad_xyz[3] = {0.f, 0.f, 0.f};
                float cw[8];
                corner_weights(pos, 1.f, cw);
                for (int c = 0; c < 3; ++c) {
                    for (int k = 0; k < B; ++k) {
                        const int e = c * B + k;
                        const float cg = r.sph[k] * (weight * in01[c] * gout[c]);
                        float vv[8];
                        for (int i = 0; i < 8; ++i) {
                            vv[i] = g.sh_data[(size_t)v.lk[i] * D + e];
                            if (G.grad_sh_out) atomic_addf(&G.grad_sh_out[(size_t)v.lk[i] * D + e], cw[i] * cg);
                        }
                        trilerp_dpos(vv, pos, cg, grad_xyz);
                    }
                }
                const double sg = 1. / (1. + (double)expf(-raw_alpha));  // _D_SIGMOID, util.hpp:29
                const float curr_grad_raw_alpha = (float)((double)curr_grad_alpha * (sg * (1 - sg)));
                {
                    float dw[8];
                    corner_weights(pos, curr_grad_raw_alpha, dw);
                    scatter8(G.grad_density_out, G.mask_out, v.lk, dw);
                }
                trilerp_dpos(dens, pos, curr_grad_raw_alpha, grad_xyz);
                const float grad_st = grad_xyz[0] * r.dir[0] + grad_xyz[1] * r.dir[1] + grad_xyz[2] * r.dir[2];
                float gs[8];
                int32_t lc[3];
                for (int k = 0; k < 3; ++k) lc[k] = std::min(v.l[k], g.size[k] - 2);
                legacy_surface_grad(type, j, v.fs, grad_st, r.origin, r.dir, lc, gs);
                scatter8(G.grad_surface_out, G.mask_out, v.lk, gs);
            }
        }
        if (expf(log_transmit) < o.stop_thresh) break;
    }
}

int legacy_check(const oracle_grid* g, const oracle_rays* r, const oracle_ray_vox* rv,
                 const oracle_render_options* o) {
    if (check_common(g, r, o)) return -1;
    if (!rv || !rv->vox_start_i || !rv->vox_num) return -1;
    if (rv->n_vox > 0 && !rv->voxel_ls) return -1;
    if (g->level_set_num < 1) return -1;
    return 0;
}


// ---------------------------------------------------------------------------------
// Evaluation renders: the traversal + cubic core without colour,
// sigma gate, window or fake samples.  surf_trav.cu:564-1705.
//   mode 0 expected termination depth (:564-794)      out0[Q]
//   mode 1 depth of the max-weight sample, 0 unless sum w > param (:796-1000)
//   mode 2 depth of the first sample with alpha > param (:1003-1168)
//   mode 3 alpha of the first sample with alpha > param (:1170-1335)
//   mode 4 surface-field gradient at the first sample with alpha > 0 (:1339-1531)  out0[Q,3]
//   mode 5 up to max_sample (depth, alpha) pairs with alpha > param (:1536-1705)   out0,out1[Q,max_sample]
// ---------------------------------------------------------------------------------
void eval_ray(const oracle_grid& g, const oracle_render_options& o_in, const Ray& r, int mode, float param,
              int max_sample, float* out0, float* out1) {
    oracle_render_options o = o_in;
    o.sigma_thresh = -INFINITY;  // these kernels have no density gate
    const int n_out = (mode == 4) ? 3 : (mode == 5 ? max_sample : 1);
    if (mode == 5) {  // the launcher hands over zeroed rows (:4066-4067); a missing ray writes one 0
        if (r.tmin > r.tmax) {
            if (max_sample > 0) out0[0] = 0.f;
            return;
        }
    } else {
        for (int i = 0; i < n_out; ++i) out0[i] = 0.f;
        if (r.tmin > r.tmax) return;
    }
    float outv = 0.f, weight_acc = 0.f, max_weight = -1.f, log_transmit = 0.f;
    int sample_id = 0;
    Dda d;
    dda_init(g, r, d);
    VoxelCtx vc;
    while (d.t <= r.tmax) {
        dda_step(g, r, d);
        bool link_empty;
        if (!load_voxel(g, o, r, d, vc, &link_empty)) continue;
        for (int li = 0; li < g.level_set_num; ++li) {
            const double lv = g.level_set_data[li];
            if (lv < vc.smin || lv > vc.smax) continue;
            double st[3] = {-1, -1, -1};
            solve_cubic_vieta(vc.fs[0] - lv, vc.fs[1], vc.fs[2], vc.fs[3], 1e-10, st);
            for (int j = 0; j < 3; ++j) {
                if (st[j] <= 0) continue;
                float pos[3];
                int32_t l[3];
                if (!root_position(g, r, d, vc, st[j], pos, l)) continue;
                const float alpha = alpha_act(trilerp(vc.dens, pos), o.alpha_activation_type);
                // (st + t_close) / step_size * world_step, evaluated in double as the reference's
                // mixed expression is
                const double depth = (st[j] + (double)d.t_close) / (double)o.step_size * (double)r.world_step;
                if (mode == 0 || mode == 1) {
                    const float pcnt = -1 * logf(1 - alpha);
                    const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
                    log_transmit -= pcnt;
                    if (mode == 0) {
                        outv = (float)((double)outv + (double)weight * depth);
                    } else {
                        weight_acc += weight;
                        if (weight > max_weight) {
                            max_weight = weight;
                            outv = (float)depth;
                        }
                    }
                } else if (mode == 2) {
                    if (alpha > param) {
                        out0[0] = (float)depth;
                        return;
                    }
                } else if (mode == 3) {
                    if (alpha > param) {
                        out0[0] = alpha;
                        return;
                    }
                } else if (mode == 4) {
                    if (alpha > 0) {
                        float sf[8];
                        for (int i = 0; i < 8; ++i) sf[i] = (float)vc.surf[i];
                        field_grad(sf, pos, out0);
                        return;
                    }
                } else {
                    if (alpha > param) {
                        out0[sample_id] = (float)depth;
                        out1[sample_id] = alpha;
                        if (++sample_id >= max_sample) return;
                    }
                }
            }
        }
        if ((mode == 0 || mode == 1) && expf(log_transmit) < o.stop_thresh) break;
    }
    if (mode == 0) out0[0] = outv;
    if (mode == 1) out0[0] = (weight_acc > param) ? outv : 0.f;
}


// =================================================================================
// The Plenoxels volume renderer (`cuvol` backend).  render_lerp_kernel_cuvol.cu
// =================================================================================
// compute_skip_dist, render_util.cuh:286-348: links below -1 hold the distance transform written by
// accelerate() (misc_kernel.cu accel_dist_*); the cell side of a link value -d is 2^(d-1).
float cuvol_skip_dist(const oracle_grid& g, const Ray& r, const int32_t* l, const float* pos) {
    const int64_t offx = (int64_t)g.size[1] * g.size[2], offy = g.size[2];
    const int32_t link_val = g.links[offx * l[0] + offy * l[1] + l[2]];
    if (link_val >= -1) return 0.f;
    const uint32_t dist = (uint32_t)(-link_val);
    const uint32_t shift = dist - 1;
    const uint32_t side = (uint32_t)((float)(1 << shift) - 1.f);
    float tmin = 0.f, tmax = 1e9f;
    for (int i = 0; i < 3; ++i) {
        int ul = ((l[i] >> shift) << shift);
        ul -= l[i];
        const float invdir = (float)(1.0 / (double)r.dir[i]);
        const float t1 = ((float)ul - pos[i]) * invdir;
        const float t2 = ((float)(ul + side) - pos[i]) * invdir;   // uint + int -> uint, then float
        if (r.dir[i] != 0.f) {
            tmin = fmaxf(tmin, fminf(t1, t2));
            if (fmaxf(t1, t2) < tmax) tmax = fmaxf(t1, t2);
        }
    }
    if (tmin > 0.f) return 0.f;
    return tmax;
}

// the sample cell and weights, render_lerp_kernel_cuvol.cu:57-62
void cuvol_cell(const oracle_grid& g, const Ray& r, float t, int32_t* l, float* pos) {
    for (int j = 0; j < 3; ++j) {
        pos[j] = fmaf(t, r.dir[j], r.origin[j]);
        pos[j] = fminf(fmaxf(pos[j], 0.f), (float)g.size[j] - 1.f);
        l[j] = std::min((int32_t)pos[j], g.size[j] - 2);
        pos[j] -= (float)l[j];
    }
}

// corner links of the cell and the MAYBE_READ_LINK gather (render_util.cuh:80-91) of column idx
void cuvol_links(const oracle_grid& g, const int32_t* l, int32_t* lk) {
    const int64_t offx = (int64_t)g.size[1] * g.size[2], offy = g.size[2];
    const int32_t* p = g.links + (offx * l[0] + offy * l[1] + l[2]);
    lk[0] = p[0];
    lk[1] = p[1];
    lk[2] = p[offy];
    lk[3] = p[offy + 1];
    lk[4] = p[offx];
    lk[5] = p[offx + 1];
    lk[6] = p[offx + offy];
    lk[7] = p[offx + offy + 1];
}
float cuvol_lerp(const float* data, int64_t stride, int idx, const int32_t* lk, const float* pos) {
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = lk[i] >= 0 ? data[(int64_t)lk[i] * stride + idx] : 0.f;
    return trilerp(v, pos);
}

// trace_ray_cuvol, render_lerp_kernel_cuvol.cu:30-125
void cuvol_forward(const oracle_grid& g, const oracle_render_options& o, Ray r, float* out, float* out_log_transmit) {
    if (r.tmin > r.tmax) {
        for (int c = 0; c < 3; ++c) out[c] = o.background_brightness;
        if (out_log_transmit) *out_log_transmit = 0.f;
        return;
    }
    const int B = g.basis_dim;
    float t = r.tmin, log_transmit = 0.f;
    float outv[3] = {0.f, 0.f, 0.f};
    while (t <= r.tmax) {
        int32_t l[3], lk[8];
        float pos[3];
        cuvol_cell(g, r, t, l, pos);
        const float skip = cuvol_skip_dist(g, r, l, pos);
        if (skip >= o.step_size) {
            t += ceilf(skip / o.step_size) * o.step_size;
            continue;
        }
        cuvol_links(g, l, lk);
        const float sigma = cuvol_lerp(g.density_data, 1, 0, lk, pos);
        if (o.last_sample_opaque && t + o.step_size > r.tmax) r.world_step = 1e9f;
        if (sigma > o.sigma_thresh) {
            const float pcnt = r.world_step * sigma;
            const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
            log_transmit -= pcnt;
            for (int c = 0; c < 3; ++c) {
                float tot = 0.f;
                for (int k = 0; k < B; ++k) tot += cuvol_lerp(g.sh_data, 3 * B, c * B + k, lk, pos) * r.sph[k];
                outv[c] += weight * fmaxf(tot + 0.5f, 0.f);
            }
            if (expf(log_transmit) < o.stop_thresh) {
                log_transmit = -1e3f;
                break;
            }
        }
        t += o.step_size;
    }
    for (int c = 0; c < 3; ++c) out[c] = outv[c] + expf(log_transmit) * o.background_brightness;
    if (out_log_transmit) *out_log_transmit = log_transmit;
}

// trace_ray_cuvol_backward, render_lerp_kernel_cuvol.cu:371-536
void cuvol_backward(const oracle_grid& g, const oracle_render_options& o, Ray r, const float* gout,
                    const float* color_cache, bool has_log_transmit, float log_transmit_in, float beta_loss,
                    float sparsity_loss, const oracle_grads& G) {
    float accum = fmaf(color_cache[0], gout[0], fmaf(color_cache[1], gout[1], color_cache[2] * gout[2]));
    if (beta_loss > 0.f) {
        const float transmit_in = expf(has_log_transmit ? log_transmit_in : 0.f);
        beta_loss = (float)((double)beta_loss * (1. - (double)transmit_in / (double)(1 - transmit_in + 1e-3)));
        accum += beta_loss;
    }
    if (r.tmin > r.tmax) return;
    const int B = g.basis_dim;
    float t = r.tmin, log_transmit = 0.f;
    while (t <= r.tmax) {
        int32_t l[3], lk[8];
        float pos[3];
        cuvol_cell(g, r, t, l, pos);
        const float skip = cuvol_skip_dist(g, r, l, pos);
        if (skip >= o.step_size) {
            t += ceilf(skip / o.step_size) * o.step_size;
            continue;
        }
        cuvol_links(g, l, lk);
        const float sigma = cuvol_lerp(g.density_data, 1, 0, lk, pos);
        if (o.last_sample_opaque && t + o.step_size > r.tmax) r.world_step = 1e9f;
        if (sigma > o.sigma_thresh) {
            const float pcnt = r.world_step * sigma;
            const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
            log_transmit -= pcnt;
            float total_color = 0.f;
            float ch[3], in01[3];
            for (int c = 0; c < 3; ++c) {
                float tot = 0.f;
                for (int k = 0; k < B; ++k) tot += cuvol_lerp(g.sh_data, 3 * B, c * B + k, lk, pos) * r.sph[k];
                tot += 0.5f;
                const float cl = fmaxf(tot, 0.f);
                in01[c] = (cl == tot) ? 1.f : 0.f;
                ch[c] = cl * gout[c];
            }
            total_color = ch[0] + ch[2];   // :469-471 (c0 + c2, then + c1)
            total_color += ch[1];
            float cw[8];
            corner_weights(pos, 1.f, cw);
            if (G.grad_sh_out) {
                for (int c = 0; c < 3; ++c) {
                    const float grad_common = weight * in01[c] * gout[c];
                    for (int k = 0; k < B; ++k) {
                        const float curr = r.sph[k] * grad_common;
                        float w8[8];
                        corner_weights(pos, curr, w8);
                        for (int i = 0; i < 8; ++i)
                            if (lk[i] >= 0) atomic_addf(&G.grad_sh_out[(int64_t)lk[i] * 3 * B + c * B + k], w8[i]);
                    }
                }
            }
            accum -= weight * total_color;
            float curr_grad_sigma = r.world_step * (total_color * expf(log_transmit) - accum);
            if (sparsity_loss > 0.f) curr_grad_sigma += sparsity_loss * (4 * sigma / (1 + 2 * (sigma * sigma)));
            if (G.grad_density_out) {
                float w8[8];
                corner_weights(pos, curr_grad_sigma, w8);
                for (int i = 0; i < 8; ++i)
                    if (lk[i] >= 0) {
                        atomic_addf(&G.grad_density_out[lk[i]], w8[i]);
                        if (G.mask_out) G.mask_out[lk[i]] = 1;
                    }
            }
            if (expf(log_transmit) < o.stop_thresh) break;
        }
        t += o.step_size;
    }
}

int cuvol_check(const oracle_grid* g, const oracle_rays* r, const oracle_render_options* o) {
    if (!g || !r || !o) return -1;
    if (g->basis_type != 1 || g->background_nlayers != 0) return -2;
    if (!(g->basis_dim == 1 || g->basis_dim == 4 || g->basis_dim == 9 || g->basis_dim == 16)) return -1;
    if (!g->links || !g->density_data || !g->sh_data) return -1;
    if (!(o->step_size > 0.f)) return -1;
    return 0;
}

}  // namespace

extern "C" {

void oracle_set_legacy_f64_surface_grad(int on) { g_legacy_f64_surface_grad = on; }
int oracle_num_threads(void) { return omp_get_max_threads(); }
void oracle_set_num_threads(int n) { omp_set_num_threads(n); }

void oracle_eval_sh(int32_t basis_dim, const float* dirs, int64_t n, float* out) {
    for (int64_t i = 0; i < n; ++i) eval_sh(basis_dim, dirs + 3 * i, out + (size_t)basis_dim * i);
}

int oracle_volume_render_surf_trav(const oracle_grid* grid, const oracle_rays* rays,
                                   const oracle_render_options* opt, float* rgb_out,
                                   const oracle_stats* stats) {
    if (int e = check_common(grid, rays, opt)) return e;
    const int64_t Q = rays->n_rays;
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t q = 0; q < Q; ++q) {
        Ray r;
        setup_ray(*grid, *opt, rays->origins + 3 * q, rays->dirs + 3 * q, r);
        Counters c;
        Dump dp = make_dump(stats, q);
        trace_forward(*grid, *opt, r, 0, nullptr, nullptr, nullptr, rgb_out + 3 * q, c,
                      dp.max_dump ? &dp : nullptr);
        attach_stats(stats, q, c, dp);
    }
    return 0;
}

int oracle_volume_render_surf_trav_backward(const oracle_grid* grid, const oracle_rays* rays,
                                            const oracle_render_options* opt,
                                            const float* grad_out, const float* color_cache,
                                            const oracle_grads* grads) {
    if (int e = check_common(grid, rays, opt)) return e;
    if (!grads || !grad_out || !color_cache) return -1;
    const int64_t Q = rays->n_rays;
    LossParams L{};
    L.l_dist_max_sample = 0;
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t q = 0; q < Q; ++q) {
        Ray r;
        setup_ray(*grid, *opt, rays->origins + 3 * q, rays->dirs + 3 * q, r);
        trace_backward(*grid, *opt, r, grad_out + 3 * q, color_cache + 3 * q, L, nullptr, nullptr,
                       nullptr, *grads);
    }
    return 0;
}

int oracle_volume_render_surf_trav_fused(const oracle_grid* grid, const oracle_rays* rays,
                                         const oracle_render_options* opt,
                                         const oracle_fused_params* fused, float* rgb_out,
                                         const oracle_grads* grads, const oracle_stats* stats) {
    if (int e = check_common(grid, rays, opt)) return e;
    if (!fused || !grads || !fused->rgb_gt || !rgb_out) return -1;
    if (fused->fused_surf_norm_reg_scale > 0.f) return -2;  // reference asserts, :2872-2875
    const int M = fused->l_dist_max_sample;
    if (M < 0 || M > kMaxCache) return -1;
    const int64_t Q = rays->n_rays;
    const int64_t Qn = fused->global_n_rays > 0 ? fused->global_n_rays : Q;
    LossParams L{};
    // launcher-side scaling, surf_trav.cu:3896-3915
    L.sparsity_loss = fused->sparsity_loss;
    L.lambda_l_dist = fused->lambda_l_dist / Qn;
    L.lambda_l_entropy = fused->lambda_l_entropy / Qn;
    L.no_norm_weight_l_entropy = fused->no_norm_weight_l_entropy != 0;
    L.lambda_l_dist_a = fused->lambda_l_dist_a / Qn;
    L.lambda_l_entropy_a = fused->lambda_l_entropy_a / Qn;
    L.lambda_l_samp_dist = fused->lambda_l_samp_dist / Qn;
    L.lambda_l_di = fused->lambda_l_di;
    L.l_di_alpha_thresh = fused->l_di_alpha_thresh;
    L.surf_sparse_alpha_thresh = fused->surf_sparse_alpha_thresh;
    L.lambda_inplace_surf_sparse = fused->lambda_inplace_surf_sparse;
    L.lambda_inwards_norm_loss = fused->lambda_inwards_norm_loss;
    L.lambda_conv_mode_samp = fused->lambda_conv_mode_samp;
    L.l_dist_max_sample = M;
    // gradient of the image loss, surf_trav.cu:3306-3316
    const float norm_l2 = 2.f / (3 * (int)Qn);
    const float norm_l1 = 1.f / (3 * (int)Qn);
#pragma omp parallel for schedule(dynamic, 16)
    for (int64_t q = 0; q < Q; ++q) {
        Ray r;
        setup_ray(*grid, *opt, rays->origins + 3 * q, rays->dirs + 3 * q, r);
        std::vector<float> cache(3 * (size_t)(M > 0 ? M : 1), 0.f);
        float* sa = cache.data();
        float* sw = sa + (M > 0 ? M : 1);
        float* sts = sw + (M > 0 ? M : 1);
        Counters c;
        Dump dp = make_dump(stats, q);
        trace_forward(*grid, *opt, r, M, sa, sw, sts, rgb_out + 3 * q, c, dp.max_dump ? &dp : nullptr);
        attach_stats(stats, q, c, dp);
        float gout[3];
        for (int i = 0; i < 3; ++i) {
            const float resid = rgb_out[3 * q + i] - fused->rgb_gt[3 * q + i];
            gout[i] = resid * norm_l2 * fused->lambda_l2;
            gout[i] += (resid > 0.f) ? (norm_l1 * fused->lambda_l1) : (-norm_l1 * fused->lambda_l1);
        }
        trace_backward(*grid, *opt, r, gout, rgb_out + 3 * q, L, sa, sw, sts, *grads);
    }
    return 0;
}

int64_t oracle_trace_voxel_lists(const oracle_grid* grid, const oracle_rays* rays,
                                 const oracle_render_options* opt, int32_t* voxel_ls,
                                 int64_t cap, int32_t* vox_start_i, int32_t* vox_num) {
    if (check_common(grid, rays, opt)) return -1;
    int64_t total = 0;
    for (int64_t q = 0; q < rays->n_rays; ++q) {
        Ray r;
        setup_ray(*grid, *opt, rays->origins + 3 * q, rays->dirs + 3 * q, r);
        int32_t n = 0;
        if (vox_start_i) vox_start_i[q] = (int32_t)total;
        if (!(r.tmin > r.tmax)) {
            Dda d;
            dda_init(*grid, r, d);
            while (d.t <= r.tmax) {
                dda_step(*grid, r, d);
                if (voxel_ls && total + n < cap) {
                    int32_t* p = voxel_ls + 3 * (total + n);
                    p[0] = d.voxel[0];
                    p[1] = d.voxel[1];
                    p[2] = d.voxel[2];
                }
                ++n;
            }
        }
        if (vox_num) vox_num[q] = n;
        total += n;
    }
    return total;
}

int oracle_setup_rays(const oracle_grid* grid, const oracle_rays* rays, const oracle_render_options* opt,
                      float* out) {
    if (check_common(grid, rays, opt)) return -1;
    for (int64_t q = 0; q < rays->n_rays; ++q) {
        Ray r;
        setup_ray(*grid, *opt, rays->origins + 3 * q, rays->dirs + 3 * q, r);
        for (int k = 0; k < 3; ++k) {
            out[8 * q + k] = r.origin[k];
            out[8 * q + 3 + k] = r.dir[k];
        }
        out[8 * q + 6] = r.tmin;
        out[8 * q + 7] = r.tmax;
    }
    return 0;
}

int oracle_test_cubic_root_grad(const double* fs, const int32_t* st_ids, float* grad_fs,
                                float* st, int64_t n, int32_t use_vieta) {
    if (!use_vieta) {  // test_cuda.cu:32-106: Vieta roots, gradient through the f/g/h chain
        for (int64_t i = 0; i < n; ++i) {
            double roots[3] = {-1, -1, -1};
            const int type = solve_cubic_vieta(fs[4 * i], fs[4 * i + 1], fs[4 * i + 2], fs[4 * i + 3], 1e-10, roots);
            for (int j = 0; j < 3; ++j) st[3 * i + j] = (float)roots[j];
            int root_num = 1;
            if (type == CUBIC_POLY) root_num = 2;
            else if (type == CUBIC_THREE_R) root_num = 3;
            if (type == CUBIC_NO_ROOT || st_ids[i] >= root_num) {
                for (int k = 0; k < 4; ++k) grad_fs[4 * i + k] = 0;
                continue;
            }
            cubic_root_grad_legacy(type, st_ids[i], fs + 4 * i, grad_fs + 4 * i);
        }
        return 0;
    }
    for (int64_t i = 0; i < n; ++i) {  // test_cuda.cu:32-106
        double roots[3] = {-1, -1, -1};
        const int type = solve_cubic_vieta(fs[4 * i], fs[4 * i + 1], fs[4 * i + 2], fs[4 * i + 3], 1e-10, roots);
        for (int j = 0; j < 3; ++j) st[3 * i + j] = (float)roots[j];
        int root_num = 1;
        if (type == CUBIC_POLY) root_num = 2;
        else if (type == CUBIC_THREE_R) root_num = 3;
        if (type == CUBIC_NO_ROOT || st_ids[i] >= root_num) {
            for (int k = 0; k < 4; ++k) grad_fs[4 * i + k] = 0;
            continue;
        }
        cubic_root_grad_vieta(type, st_ids[i], fs + 4 * i, grad_fs + 4 * i);
    }
    return 0;
}

int oracle_volume_render_surface(const oracle_grid* grid, const oracle_rays* rays,
                                 const oracle_ray_vox* ray_vox, const oracle_render_options* opt,
                                 float* rgb_out) {
    if (int rc = legacy_check(grid, rays, ray_vox, opt)) return rc;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t q = 0; q < rays->n_rays; ++q) {
        Ray r;
        setup_ray(*grid, *opt, rays->origins + 3 * q, rays->dirs + 3 * q, r);
        legacy_forward(*grid, *opt, r, ray_vox->voxel_ls, ray_vox->vox_start_i[q], ray_vox->vox_num[q],
                       rgb_out + 3 * q);
    }
    return 0;
}

int oracle_volume_render_surface_backward(const oracle_grid* grid, const oracle_rays* rays,
                                          const oracle_ray_vox* ray_vox,
                                          const oracle_render_options* opt, const float* grad_out,
                                          const float* color_cache, const oracle_grads* grads) {
    if (int rc = legacy_check(grid, rays, ray_vox, opt)) return rc;
    if (!grads || !grad_out || !color_cache) return -1;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t q = 0; q < rays->n_rays; ++q) {
        Ray r;
        setup_ray(*grid, *opt, rays->origins + 3 * q, rays->dirs + 3 * q, r);
        legacy_backward(*grid, *opt, r, ray_vox->voxel_ls, ray_vox->vox_start_i[q], ray_vox->vox_num[q],
                        grad_out + 3 * q, color_cache + 3 * q, 0.f, *grads);
    }
    return 0;
}

// surface.cu:1325-1439: forward, then backward with gout = (rgb - gt) * 2 / (3 Q) (:1013-1018)
int oracle_volume_render_surface_fused(const oracle_grid* grid, const oracle_rays* rays,
                                       const oracle_ray_vox* ray_vox,
                                       const oracle_render_options* opt, const float* rgb_gt,
                                       float beta_loss, float sparsity_loss, float* rgb_out,
                                       const oracle_grads* grads) {
    (void)beta_loss;  // only reaches the background model (:636-640), which is not rendered here
    if (int rc = legacy_check(grid, rays, ray_vox, opt)) return rc;
    if (!grads || !rgb_gt || !rgb_out) return -1;
    const float norm = 2.f / (3 * (int)rays->n_rays);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t q = 0; q < rays->n_rays; ++q) {
        Ray r;
        setup_ray(*grid, *opt, rays->origins + 3 * q, rays->dirs + 3 * q, r);
        legacy_forward(*grid, *opt, r, ray_vox->voxel_ls, ray_vox->vox_start_i[q], ray_vox->vox_num[q],
                       rgb_out + 3 * q);
        float gout[3];
        for (int c = 0; c < 3; ++c) gout[c] = (rgb_out[3 * q + c] - rgb_gt[3 * q + c]) * norm;
        legacy_backward(*grid, *opt, r, ray_vox->voxel_ls, ray_vox->vox_start_i[q], ray_vox->vox_num[q],
                        gout, rgb_out + 3 * q, sparsity_loss, *grads);
    }
    return 0;
}

int oracle_render_eval(const oracle_grid* grid, const oracle_rays* rays, const oracle_render_options* opt,
                       int32_t mode, float param, int32_t max_sample, float* out0, float* out1) {
    if (check_common(grid, rays, opt)) return -1;
    if (mode < 0 || mode > 5 || !out0 || (mode == 5 && (!out1 || max_sample < 1))) return -1;
    const int stride = (mode == 4) ? 3 : (mode == 5 ? max_sample : 1);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t q = 0; q < rays->n_rays; ++q) {
        Ray r;
        setup_ray(*grid, *opt, rays->origins + 3 * q, rays->dirs + 3 * q, r);
        eval_ray(*grid, *opt, r, mode, param, max_sample, out0 + (size_t)stride * q,
                 out1 ? out1 + (size_t)stride * q : nullptr);
    }
    return 0;
}

/* render_lerp_kernel_cuvol.cu:1120-1165 */
int oracle_volume_render_cuvol(const oracle_grid* grid, const oracle_rays* rays, const oracle_render_options* opt,
                               float* rgb_out) {
    if (int rc = cuvol_check(grid, rays, opt)) return rc;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t q = 0; q < rays->n_rays; ++q) {
        Ray r;
        setup_ray(*grid, *opt, rays->origins + 3 * q, rays->dirs + 3 * q, r);
        cuvol_forward(*grid, *opt, r, rgb_out + 3 * q, nullptr);
    }
    return 0;
}

/* render_lerp_kernel_cuvol.cu:1209-1270 */
int oracle_volume_render_cuvol_backward(const oracle_grid* grid, const oracle_rays* rays,
                                        const oracle_render_options* opt, const float* grad_out,
                                        const float* color_cache, const oracle_grads* grads) {
    if (int rc = cuvol_check(grid, rays, opt)) return rc;
    if (!grads || !grad_out || !color_cache) return -1;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t q = 0; q < rays->n_rays; ++q) {
        Ray r;
        setup_ray(*grid, *opt, rays->origins + 3 * q, rays->dirs + 3 * q, r);
        cuvol_backward(*grid, *opt, r, grad_out + 3 * q, color_cache + 3 * q, false, 0.f, 0.f, 0.f, *grads);
    }
    return 0;
}

/* render_lerp_kernel_cuvol.cu:1272-1354: beta_loss / Q (:1323), gout = (rgb - gt) * 2 / (3 Q) (:889-895) */
int oracle_volume_render_cuvol_fused(const oracle_grid* grid, const oracle_rays* rays,
                                     const oracle_render_options* opt, const float* rgb_gt, float beta_loss,
                                     float sparsity_loss, float* rgb_out, const oracle_grads* grads) {
    if (int rc = cuvol_check(grid, rays, opt)) return rc;
    if (!grads || !rgb_gt || !rgb_out) return -1;
    const float norm = 2.f / (3 * (int)rays->n_rays);
    const float beta = beta_loss / (float)rays->n_rays;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t q = 0; q < rays->n_rays; ++q) {
        Ray r;
        setup_ray(*grid, *opt, rays->origins + 3 * q, rays->dirs + 3 * q, r);
        float log_t = 0.f;
        cuvol_forward(*grid, *opt, r, rgb_out + 3 * q, &log_t);
        float gout[3];
        for (int c = 0; c < 3; ++c) gout[c] = (rgb_out[3 * q + c] - rgb_gt[3 * q + c]) * norm;
        cuvol_backward(*grid, *opt, r, gout, rgb_out + 3 * q, beta_loss > 0.f, log_t, beta, sparsity_loss, *grads);
    }
    return 0;
}

/* grid_weight_render / sparse_grid_weight_render, misc_kernel.cu:186-400, 888-937 (cam2world_ray:
 * render_util.cuh:599-617).  links == NULL: `data` is the dense [X,Y,Z] volume and the vertex keeps the largest
 * weight; otherwise data = density rows and the vertex keeps the largest transmittance in front of a sample.
 * PARITY UNPINNED: the reference has no second implementation of these two kernels to take vectors from. */
int oracle_weight_render(const float* data, const int32_t* links, const int32_t* size, const float* c2w, float fx,
                         float fy, float cx, float cy, int32_t width, int32_t height, float step_size,
                         float stop_thresh, int32_t last_sample_opaque, const float* offset, const float* scaling,
                         float* out) {
    if (!data || !size || !c2w || !offset || !scaling || !out || !(step_size > 0.f)) return -1;
    const int st0 = size[1] * size[2], st1 = size[2];
    const bool sparse = links != nullptr;
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t tid = 0; tid < (int64_t)width * height; ++tid) {
        const int iy = (int)(tid / width), ix = (int)(tid % width);
        float x = ((float)ix + 0.5f - cx) / fx;
        float y = ((float)iy + 0.5f - cy) / fy;
        float z = sqrtf((float)((double)(x * x + y * y) + 1.0));
        x /= z;
        y /= z;
        z = 1.0f / z;
        float d[3], o[3];
        for (int i = 0; i < 3; ++i) {
            d[i] = c2w[4 * i] * x + c2w[4 * i + 1] * y + c2w[4 * i + 2] * z;
            o[i] = fmaf(c2w[4 * i + 3], scaling[i], offset[i]);
        }
        for (int i = 0; i < 3; ++i) d[i] *= scaling[i];
        const float rn = 1.f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        for (int i = 0; i < 3; ++i) d[i] *= rn;
        const float world_step = rn * step_size;
        float t = 0.f, tmax = 2e3f;
        for (int i = 0; i < 3; ++i) {
            const float invdir = (float)(1.0 / (double)d[i]);
            const float t1 = (-0.5f - o[i]) * invdir;
            const float t2 = ((float)size[i] - 0.5f - o[i]) * invdir;
            if (d[i] != 0.f) {
                t = fmaxf(t, fminf(t1, t2));
                tmax = fminf(tmax, fmaxf(t1, t2));
            }
        }
        if (t > tmax) continue;
        float lli = 0.f;
        while (t <= tmax) {
            float pos[3];
            int32_t l[3];
            for (int j = 0; j < 3; ++j) {
                pos[j] = o[j] + t * d[j];
                pos[j] = fminf(fmaxf(pos[j], 0.f), (float)size[j] - 1.f);
                l[j] = std::min((int32_t)pos[j], size[j] - 2);
                pos[j] -= (float)l[j];
            }
            const int idx = l[0] * st0 + l[1] * st1 + l[2];
            const int offs[8] = {0, 1, st1, st1 + 1, st0, st0 + 1, st0 + st1, st0 + st1 + 1};
            float v[8];
            for (int i = 0; i < 8; ++i) {
                if (sparse) {
                    const int32_t lk = links[idx + offs[i]];
                    v[i] = lk >= 0 ? data[lk] : 0.f;
                } else {
                    v[i] = data[idx + offs[i]];
                }
            }
            float sigma = trilerp(v, pos);
            if (!sparse && last_sample_opaque && t + step_size > tmax) {
                sigma += 1e9f;
                lli = 0.f;
            }
            if (sigma > 1e-8f) {
                const float log_att = -world_step * sigma;
                const float tr = expf(lli);
                const float val = sparse ? tr : tr * (1.f - expf(log_att));
                lli += log_att;
                for (int i = 0; i < 8; ++i) {
                    float* q = out + idx + offs[i];
#pragma omp critical(oracle_weight_max)
                    if (val > *q) *q = val;
                }
                if (expf(lli) < stop_thresh) break;
            }
            t += step_size;
        }
    }
    return 0;
}

/* render_lerp_kernel_cuvol.cu:127-370, 1356-1442: depth renders of the cuvol march.  mode 0 expected term,
 * 1 mode term (param = weight_thresh), 2 med term (out0 depths / out1 sigmas, [Q, max_sample], zeroed here),
 * 3 first sample above param. */
int oracle_render_eval_cuvol(const oracle_grid* grid, const oracle_rays* rays, const oracle_render_options* opt,
                             int32_t mode, float param, int32_t max_sample, float* out0, float* out1) {
    if (int rc = cuvol_check(grid, rays, opt)) return rc;
    if (mode < 0 || mode > 3 || !out0 || (mode == 2 && (!out1 || max_sample < 1))) return -1;
    const oracle_grid& g = *grid;
    const oracle_render_options& o = *opt;
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t q = 0; q < rays->n_rays; ++q) {
        Ray r;
        setup_ray(g, o, rays->origins + 3 * q, rays->dirs + 3 * q, r);
        if (mode == 2)
            for (int i = 0; i < max_sample; ++i) out0[q * max_sample + i] = out1[q * max_sample + i] = 0.f;
        else
            out0[q] = 0.f;
        if (r.tmin > r.tmax) continue;
        float t = r.tmin, outv = 0.f, weight_acc = 0.f, max_weight = -1.f, log_transmit = 0.f;
        int sample_i = 0;
        const float thresh = mode == 3 ? param : o.sigma_thresh;
        while (t <= r.tmax) {
            int32_t l[3], lk[8];
            float pos[3];
            cuvol_cell(g, r, t, l, pos);
            const float skip = cuvol_skip_dist(g, r, l, pos);
            if (skip >= o.step_size) {
                t += ceilf(skip / o.step_size) * o.step_size;
                continue;
            }
            cuvol_links(g, l, lk);
            const float sigma = cuvol_lerp(g.density_data, 1, 0, lk, pos);
            if (sigma > thresh) {
                const float depth = (t / o.step_size) * r.world_step;
                if (mode == 3) {
                    outv = depth;
                    break;
                }
                const float pcnt = r.world_step * sigma;
                const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
                log_transmit -= pcnt;
                if (mode == 0) {
                    outv += weight * depth;
                    weight_acc += weight;
                } else if (mode == 1) {
                    weight_acc += weight;
                    if (weight > max_weight) {
                        max_weight = weight;
                        outv = depth;
                    }
                } else if (sample_i < max_sample) {
                    out0[q * max_sample + sample_i] = depth;
                    out1[q * max_sample + sample_i] = sigma;
                    ++sample_i;
                }
                if (expf(log_transmit) < o.stop_thresh) break;
            }
            t += o.step_size;
        }
        if (mode == 3) out0[q] = outv;
        if (mode == 0 || mode == 1) out0[q] = weight_acc > param ? outv : 0.f;
    }
    return 0;
}

// cubic_extract_iso_pts (svox2_kernel.cu:248-372, 542-583): per listed cell, n_sample x n_sample axis-parallel
// lines in each of the three directions; the first root of the level field in [0,1] whose trilinear mask value
// reaches density_thresh is written as a grid-space point.  `out` [n_cells, 3 n^2, 3] is zeroed by the caller.
int oracle_cubic_extract_iso_pts(const int32_t* links, const int32_t* size, const float* level_data,
                                 const float* mask_data, const int32_t* cell_ids, int64_t n_cells,
                                 int32_t n_sample, float density_thresh, float* out) {
    const int X = size[0], Y = size[1], Z = size[2];
    const float step = 1.f / (float)(n_sample - 1);
    for (int64_t tid = 0; tid < n_cells; ++tid) {
        const int xyz = cell_ids[tid];
        const int z = xyz % Z, xy = xyz / Z, y = xy % Y, x = xy / Y;
        if (x >= X - 1 || y >= Y - 1 || z >= Z - 1) continue;
        int lk[8];
        bool ok = true;
        for (int c = 0; c < 8; ++c) {
            lk[c] = links[((int64_t)(x + ((c >> 2) & 1)) * Y + (y + ((c >> 1) & 1))) * Z + (z + (c & 1))];
            ok = ok && lk[c] >= 0;
        }
        if (!ok) continue;
        double surf[8];
        float mask[8];
        for (int c = 0; c < 8; ++c) {
            surf[c] = level_data[lk[c]];
            mask[c] = mask_data[lk[c]];
        }
        for (int i = 0; i < n_sample; ++i) {
            const float pos1 = i * step;
            for (int j = 0; j < n_sample; ++j) {
                const float pos2 = j * step;
                for (int dir_id = 0; dir_id < 3; ++dir_id) {
                    double d[3] = {0., 0., 0.}, o[3] = {0., 0., 0.};
                    d[dir_id] = 1.;
                    o[dir_id == 0 ? 1 : 0] = pos1;
                    o[dir_id == 2 ? 1 : 2] = pos2;
                    double fs[4], st[3] = {-1, -1, -1};
                    surface_cubic(surf, o, d, fs);
                    solve_cubic_vieta(fs[0], fs[1], fs[2], fs[3], 1e-10, st);
                    for (int k = 0; k < 3; ++k) {
                        if (!(st[k] >= 0. && st[k] <= 1.)) continue;
                        const float pt[3] = {(float)(o[0] + d[0] * st[k]), (float)(o[1] + d[1] * st[k]),
                                             (float)(o[2] + d[2] * st[k])};
                        if (trilerp(mask, pt) >= density_thresh) {
                            float* dst = out + ((int64_t)tid * 3 * n_sample * n_sample + (i * n_sample * 3 + j * 3 + dir_id)) * 3;
                            dst[0] = pt[0] + x;
                            dst[1] = pt[1] + y;
                            dst[2] = pt[2] + z;
                            break;
                        }
                    }
                }
            }
        }
    }
    return 0;
}


}  // extern "C"
