// The Plenoxels volume renderer (`cuvol` backend):
// render_lerp_kernel_cuvol.cu:30-125 (forward), :371-536 (backward), kernels :764-927,
// launchers :1120-1354.  Fixed-step ray marching: at t_k = tmin + k * step_size the raw density is
// interpolated trilinearly (a missing corner reads as 0); samples with density > sigma_thresh are
// shaded with the SH rows of the 8 corners and composited front to back.
//
// One ray per lane marches; lanes that reached a counting sample are served one after another by
// the whole wave (lanes = corner x SH coefficient: row-contiguous loads and atomics,
// coop_shade_fwd / coop_shade_bwd of asurf_kernels.h).  The march tests one bit per cell of the
// per-call occupancy pyramid ("some corner density exceeds sigma_thresh": a trilinear value
// cannot exceed its corners) from register-cached words and leaves an empty 8^3 brick in one move.
// t advances by repeated float addition, t += step_size, also across a skipped brick (a short
// register loop), so the sample parameters are bit for bit those of the reference's march over
// un-accelerated links (:56, :116) and of its PyTorch restatement (svox2.py:1270-1351).  With
// accelerate()d links the reference itself moves by ceilf(skip / step) * step (:75) and its samples
// drift from its own un-accelerated ones by float rounding; that variant is not reproduced.
#pragma once
#include "asurf_kernels.h"

namespace asurf {

struct March {
    float ox, oy, oz, dx, dy, dz;
    float t, tmax, step;
    int lim0, lim1, lim2;  // size - 1
    int cur_key, cur_brick, cur_lx;
    unsigned cur_flag;
    unsigned long long cur_word;
};

__device__ __forceinline__ void march_init(const KGrid& g, const KOpt& o, const Ray& r, March& m) {
    m.ox = r.o[0];
    m.oy = r.o[1];
    m.oz = r.o[2];
    m.dx = r.d[0];
    m.dy = r.d[1];
    m.dz = r.d[2];
    m.t = r.tmin;
    m.tmax = r.tmax;
    m.step = o.step_size;
    m.lim0 = g.size[0] - 1;
    m.lim1 = g.size[1] - 1;
    m.lim2 = g.size[2] - 1;
    m.cur_key = -1;
    m.cur_brick = 0;
    m.cur_lx = -1;
    m.cur_flag = 0;
    m.cur_word = 0;
}

// Cell and interpolation weights at parameter t (render_lerp_kernel_cuvol.cu:54-60)
__device__ __forceinline__ void march_cell(const March& m, float t, int* l, float* pos) {
    const float p0 = fminf(fmaxf(fmaf(t, m.dx, m.ox), 0.f), (float)m.lim0);
    const float p1 = fminf(fmaxf(fmaf(t, m.dy, m.oy), 0.f), (float)m.lim1);
    const float p2 = fminf(fmaxf(fmaf(t, m.dz, m.oz), 0.f), (float)m.lim2);
    l[0] = min((int)p0, m.lim0 - 1);
    l[1] = min((int)p1, m.lim1 - 1);
    l[2] = min((int)p2, m.lim2 - 1);
    pos[0] = p0 - (float)l[0];
    pos[1] = p1 - (float)l[1];
    pos[2] = p2 - (float)l[2];
}

// Advances m.t to the next sample parameter whose cell is flagged; returns false when the ray ends.
// On success *t_out, l, pos describe that point (the caller then adds one step).
template <bool LDS>
__device__ __forceinline__ bool march_next(const KGrid& g, const unsigned int* lds_coarse, March& m,
                                           float* t_out, int* l, float* pos) {
    // the guard only exists so that no defect can keep a wave alive (every iteration advances t)
    for (int guard = 0; guard < (1 << 22); ++guard) {
        const float t = m.t;
        if (!(t <= m.tmax)) return false;
        march_cell(m, t, l, pos);
        const int key = ((l[0] >> 3) << 20) | ((l[1] >> 3) << 10) | (l[2] >> 3);
        if (key != m.cur_key) {
            m.cur_key = key;
            const int bz = l[2] >> 3;
            const int bxy = (l[0] >> 3) * g.nb[1] + (l[1] >> 3);
            m.cur_brick = bxy * g.nb[2] + bz;
            const int wi = bxy * g.nwz + (bz >> 5);
            unsigned word;
            if constexpr (LDS) word = lds_coarse[wi];
            else word = g.coarse[wi];
            m.cur_flag = (word >> (bz & 31)) & 1u;
            m.cur_lx = -1;
        }
        if (!m.cur_flag) {
            // leave the empty brick: the samples up to its exit lie in it.  Conservative (floor with
            // a margin, and only while no coordinate is being clamped), never past the exit.
            int adv = 1;
            const float px = fmaf(t, m.dx, m.ox), py = fmaf(t, m.dy, m.oy), pz = fmaf(t, m.dz, m.oz);
            const bool inside = px >= 0.f && px <= (float)m.lim0 && py >= 0.f && py <= (float)m.lim1 &&
                                pz >= 0.f && pz <= (float)m.lim2;
            if (inside) {
                const float bx0 = (float)(l[0] & ~7), by0 = (float)(l[1] & ~7), bz0 = (float)(l[2] & ~7);
                const float ex = (m.dx > 0.f) ? fminf(bx0 + 8.f, (float)m.lim0) : bx0;
                const float ey = (m.dy > 0.f) ? fminf(by0 + 8.f, (float)m.lim1) : by0;
                const float ez = (m.dz > 0.f) ? fminf(bz0 + 8.f, (float)m.lim2) : bz0;
                const float tx = (m.dx != 0.f) ? (ex - m.ox) / m.dx : 1e30f;
                const float ty = (m.dy != 0.f) ? (ey - m.oy) / m.dy : 1e30f;
                const float tz = (m.dz != 0.f) ? (ez - m.oz) / m.dz : 1e30f;
                const float t_exit = fminf(fminf(tx, ty), tz);
                const float n = floorf((t_exit - t) / m.step - 0.01f);
                if (n > 1.f) adv = (int)fminf(n, 65536.f);
            }
            for (int i = 0; i < adv; ++i) m.t += m.step;
            continue;
        }
        const int lx = l[0] & 7;
        if (lx != m.cur_lx) {
            m.cur_lx = lx;
            m.cur_word = g.fine[(size_t)m.cur_brick * 8 + lx];
        }
        if ((m.cur_word >> (((l[1] & 7) << 3) | (l[2] & 7))) & 1ull) {
            *t_out = t;
            return true;
        }
        m.t += m.step;
    }
    return false;
}

// 8 corner links of cell l and the trilinear raw density (missing corner = 0)
__device__ __forceinline__ float cuvol_sigma(const KGrid& g, const int* l, const float* pos, int* lk) {
    const int32_t* p = g.links + ((size_t)l[0] * g.offx + (size_t)l[1] * g.offy + l[2]);
    lk[0] = p[0];
    lk[1] = p[1];
    lk[2] = p[g.offy];
    lk[3] = p[g.offy + 1];
    lk[4] = p[g.offx];
    lk[5] = p[g.offx + 1];
    lk[6] = p[g.offx + g.offy];
    lk[7] = p[g.offx + g.offy + 1];
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = lk[i] >= 0 ? g.density[lk[i]] : 0.f;
    return trilerp8(v, pos);
}

__device__ __forceinline__ void cuvol_ray(const KGrid& g, const KOpt& o, const float* origins, const float* dirs,
                                          long long q, bool valid, Ray& r, float* world_step) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.o[k] = valid ? origins[3 * q + k] : 0.f;
        r.d[k] = valid ? dirs[3 * q + k] : 1.f;
    }
    const float ds0 = r.d[0] * g.scaling[0], ds1 = r.d[1] * g.scaling[1], ds2 = r.d[2] * g.scaling[2];
    *world_step = (1.f / sqrtf(ds0 * ds0 + ds1 * ds1 + ds2 * ds2)) * o.step_size;
}

// ------------------------------------------------------------------------------------
template <int B, bool CLDS>
__global__ void __launch_bounds__(kBlock)
k_cuvol_fwd(KGrid g, KOpt o, int last_sample_opaque, const float* __restrict__ origins,
            const float* __restrict__ dirs, long long Q, float* __restrict__ rgb_out,
            float* __restrict__ log_transmit_out) {
    extern __shared__ unsigned int s_coarse[];
    __shared__ float s_sph[kBlock * kSphPad];
    __shared__ LaneRec s_rec[kBlock];
    const int tid = threadIdx.x;
    if constexpr (CLDS) {
        for (int i = tid; i < g.coarse_words; i += kBlock) s_coarse[i] = g.coarse[i];
        __syncthreads();
    }
    const long long q = (long long)blockIdx.x * kBlock + tid;
    const bool valid = q < Q;
    Ray r;
    float world_step;
    cuvol_ray(g, o, origins, dirs, q, valid, r, &world_step);
    {
        float sph[B];
        eval_sh<B>(r.d, sph);
#pragma unroll
        for (int k = 0; k < B; ++k) s_sph[tid * kSphPad + k] = sph[k];
    }
    setup_ray(g, o, r);
    bool active = valid && !(r.tmin > r.tmax);
    const bool misses = valid && !active;
    March m;
    march_init(g, o, r, m);
    float outv[3] = {0.f, 0.f, 0.f};
    float log_transmit = 0.f;
    const float* sph_wave = s_sph + (tid & ~63) * kSphPad;
    const LaneRec* rec_wave = s_rec + (tid & ~63);
    while (__any(active)) {
        float pos[3] = {0.f, 0.f, 0.f};
        float sigma = 0.f, t = 0.f;
        bool hit = false;
        int lk[8];
        while (active && !hit) {
            int l[3];
            if (!march_next<CLDS>(g, s_coarse, m, &t, l, pos)) {
                active = false;
                break;
            }
            sigma = cuvol_sigma(g, l, pos, lk);
            hit = sigma > o.sigma_thresh;
            m.t += m.step;
        }
        const unsigned long long need = __ballot(hit);
        if (!need) continue;
        if (hit) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s_rec[tid].link[i] = lk[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float color[3] = {0.f, 0.f, 0.f};
        coop_shade_fwd<B>(g, need, rec_wave, sph_wave, pos, color);
        if (hit) {
            const float ws = (last_sample_opaque && t + o.step_size > r.tmax) ? 1e9f : world_step;
            const float pcnt = ws * sigma;
            const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
            log_transmit -= pcnt;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) outv[ch] += weight * fmaxf(color[ch] + 0.5f, 0.f);
            if (expf(log_transmit) < o.stop_thresh) {
                log_transmit = -1e3f;
                active = false;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (valid) {
        const float bg = o.background_brightness;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) rgb_out[3 * q + ch] = misses ? bg : outv[ch] + expf(log_transmit) * bg;
        if (log_transmit_out != nullptr) log_transmit_out[q] = misses ? 0.f : log_transmit;
    }
}

// ------------------------------------------------------------------------------------
template <int B, bool CLDS>
__global__ void __launch_bounds__(kBlock)
k_cuvol_bwd(KGrid g, KOpt o, int last_sample_opaque, KGrads G, const float* __restrict__ origins,
            const float* __restrict__ dirs, long long Q, const float* __restrict__ grad_in,
            const float* __restrict__ color_cache, int grad_in_is_rgb,
            const float* __restrict__ log_transmit_in, float beta_loss, float sparsity_loss) {
    extern __shared__ unsigned int s_coarse[];
    __shared__ float s_sph[kBlock * kSphPad];
    __shared__ LaneRec s_rec[kBlock];
    const int tid = threadIdx.x;
    if constexpr (CLDS) {
        for (int i = tid; i < g.coarse_words; i += kBlock) s_coarse[i] = g.coarse[i];
        __syncthreads();
    }
    const long long q = (long long)blockIdx.x * kBlock + tid;
    const bool valid = q < Q;
    Ray r;
    float world_step;
    cuvol_ray(g, o, origins, dirs, q, valid, r, &world_step);
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
        const float norm = 2.f / (3 * (int)Q);  // :889-895
        float cc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            cc[i] = color_cache[3 * q + i];
            gout[i] = grad_in_is_rgb ? (cc[i] - grad_in[3 * q + i]) * norm : grad_in[3 * q + i];
        }
        accum = fmaf(cc[0], gout[0], fmaf(cc[1], gout[1], cc[2] * gout[2]));
        if (beta_loss > 0.f && log_transmit_in != nullptr) {  // :396-400
            const float tr = expf(log_transmit_in[q]);
            accum += beta_loss * (1 - tr / (1 - tr + 1e-3f));
        }
    }
    March m;
    march_init(g, o, r, m);
    float log_transmit = 0.f;
    const float* sph_wave = s_sph + (tid & ~63) * kSphPad;
    const LaneRec* rec_wave = s_rec + (tid & ~63);
    while (__any(active)) {
        float pos[3] = {0.f, 0.f, 0.f};
        float sigma = 0.f, t = 0.f;
        bool hit = false;
        int lk[8];
        while (active && !hit) {
            int l[3];
            if (!march_next<CLDS>(g, s_coarse, m, &t, l, pos)) {
                active = false;
                break;
            }
            sigma = cuvol_sigma(g, l, pos, lk);
            hit = sigma > o.sigma_thresh;
            m.t += m.step;
        }
        const unsigned long long need = __ballot(hit);
        if (!need) continue;
        float ws = world_step, weight = 0.f;
        if (hit) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s_rec[tid].link[i] = lk[i];
            ws = (last_sample_opaque && t + o.step_size > r.tmax) ? 1e9f : world_step;
            const float pcnt = ws * sigma;
            weight = expf(log_transmit) * (1.f - expf(-pcnt));
            log_transmit -= pcnt;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float total_color = 0.f;
        float unused[3] = {0.f, 0.f, 0.f};
        coop_shade_bwd<B>(g, G, false, need, rec_wave, sph_wave, pos, weight, gout, &total_color, unused);
        if (hit) {
            accum -= weight * total_color;
            float grad_sigma = ws * (total_color * expf(log_transmit) - accum);  // :486-487
            if (sparsity_loss > 0.f) grad_sigma += sparsity_loss * (4 * sigma / (1 + 2 * (sigma * sigma)));
            if (G.grad_density != nullptr) {
                float cw[8];
                corner_weights(pos, grad_sigma, cw);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (lk[i] >= 0) {
                        atomicAdd(&G.grad_density[lk[i]], cw[i]);
                        if (G.mask != nullptr) G.mask[lk[i]] = 1;
                    }
                }
            }
            if (expf(log_transmit) < o.stop_thresh) active = false;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

}  // namespace asurf
