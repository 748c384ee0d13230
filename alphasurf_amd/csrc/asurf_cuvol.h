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

// Advances m.t to the next sample parameter whose cell is flagged; on success *t_out, l, pos describe
// that point (the caller then adds one step).
// `budget` (lattice points this call may still visit) bounds the divergence between the lanes of a
// wave: 0 = ray ended, 1 = found, 2 = budget spent (call again).
template <bool LDS>
__device__ __forceinline__ int march_next(const KGrid& g, const unsigned int* lds_coarse, March& m,
                                          float* t_out, int* l, float* pos, int& budget) {
    // every iteration advances t and spends budget: no defect can keep a wave alive
    while (budget > 0) {
        --budget;
        const float t = m.t;
        if (!(t <= m.tmax)) return 0;
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
            return 1;
        }
        m.t += m.step;
    }
    return 2;
}

// ------------------------------------------------------------------------------------
// Records.  A lane gathers up to kCuvolK consecutive counting samples of its ray that lie in ONE
// cell (at step 0.5 a ray spends ~1.7 samples per cell) and the wave then serves that record: the
// cell's 8 SH rows are loaded once (and stay in registers while the next lanes' records name the
// same cell), every sample costs only its weights and three reductions, and in the backward pass
// the SH gradient of all samples of all consecutive records of one cell leaves as ONE set of
// row-contiguous atomics.  Weights and the stopping rule depend on the densities alone
// (render_lerp_kernel_cuvol.cu:99-111), so the lane knows them before any colour exists.
// ------------------------------------------------------------------------------------
constexpr int kCuvolK = 4;
#ifndef ASURF_CUVOL_BUDGET
#define ASURF_CUVOL_BUDGET 4
#endif
constexpr int kCuvolBudget = ASURF_CUVOL_BUDGET;  // lattice points a lane may visit per round

struct CuvolCell {  // the cell whose corner data the lane holds in registers
    int cid;        // flat index of its first corner in `links`; -1 = none
    int lk0, lk1, lk2, lk3, lk4, lk5, lk6, lk7;
    float d0, d1, d2, d3, d4, d5, d6, d7;
};

struct CuvolSmp {
    float px, py, pz;  // position in the cell
    float w;           // compositing weight
    float ws;          // world step (1e9 for an opaque last sample)
    float ta;          // transmittance after the sample
    float sig;
};

// Next sample whose density exceeds sigma_thresh; m.t is left one step past it.
template <bool CLDS>
__device__ __forceinline__ int cuvol_next(const KGrid& g, const KOpt& o, const unsigned int* lds_coarse,
                                          March& m, CuvolCell& c, float* t, float* pos, float* sigma, int& budget) {
    while (true) {
        int l[3];
        const int got = march_next<CLDS>(g, lds_coarse, m, t, l, pos, budget);
        if (got != 1) return got;
        const int cid = l[0] * g.offx + l[1] * g.offy + l[2];
        if (cid != c.cid) {
            c.cid = cid;
            const int32_t* p = g.links + cid;
            c.lk0 = p[0];
            c.lk1 = p[1];
            c.lk2 = p[g.offy];
            c.lk3 = p[g.offy + 1];
            c.lk4 = p[g.offx];
            c.lk5 = p[g.offx + 1];
            c.lk6 = p[g.offx + g.offy];
            c.lk7 = p[g.offx + g.offy + 1];
            c.d0 = c.lk0 >= 0 ? g.density[c.lk0] : 0.f;
            c.d1 = c.lk1 >= 0 ? g.density[c.lk1] : 0.f;
            c.d2 = c.lk2 >= 0 ? g.density[c.lk2] : 0.f;
            c.d3 = c.lk3 >= 0 ? g.density[c.lk3] : 0.f;
            c.d4 = c.lk4 >= 0 ? g.density[c.lk4] : 0.f;
            c.d5 = c.lk5 >= 0 ? g.density[c.lk5] : 0.f;
            c.d6 = c.lk6 >= 0 ? g.density[c.lk6] : 0.f;
            c.d7 = c.lk7 >= 0 ? g.density[c.lk7] : 0.f;
        }
        const float v[8] = {c.d0, c.d1, c.d2, c.d3, c.d4, c.d5, c.d6, c.d7};
        *sigma = trilerp8(v, pos);
        m.t += m.step;
        if (*sigma > o.sigma_thresh) return 1;
    }
}

// Per-lane gathering state that survives from one record to the next
struct CuvolGather {
    CuvolCell cell;
    bool pend;  // a counting sample of the NEXT cell was already found (cell holds its data)
    float pend_t, pend_px, pend_py, pend_pz, pend_sig;
    float log_transmit;
};

// Fills one record (<= kCuvolK samples of one cell) for this lane: returns the sample count, writes
// the cell's links to *rec.  `active` turns false when the ray has ended or reached stop_thresh.
template <bool CLDS>
__device__ __forceinline__ int cuvol_gather(const KGrid& g, const KOpt& o, const unsigned int* lds_coarse,
                                            March& m, CuvolGather& st, float tmax, float world_step,
                                            int last_sample_opaque, bool& active, LaneRec* rec, int* rec_cid,
                                            CuvolSmp* smp) {
    int n = 0;
    bool collecting = active;
    int budget = kCuvolBudget;
#pragma unroll
    for (int j = 0; j < kCuvolK; ++j) {
        smp[j].px = smp[j].py = smp[j].pz = 0.f;
        smp[j].w = 0.f;
        smp[j].ws = world_step;
        smp[j].ta = 0.f;
        smp[j].sig = 0.f;
        if (collecting) {
            float t = st.pend_t, pos[3] = {st.pend_px, st.pend_py, st.pend_pz}, sigma = st.pend_sig;
            bool have = st.pend;
            if (!have) {
                const int got = cuvol_next<CLDS>(g, o, lds_coarse, m, st.cell, &t, pos, &sigma, budget);
                have = got == 1;
                if (!have) {
                    active = active && got != 0;
                    collecting = false;
                } else if (j > 0 && st.cell.cid != *rec_cid) {
                    // first sample of the next cell: keep it for the next record
                    st.pend = true;
                    st.pend_t = t;
                    st.pend_px = pos[0];
                    st.pend_py = pos[1];
                    st.pend_pz = pos[2];
                    st.pend_sig = sigma;
                    have = false;
                    collecting = false;
                }
            }
            if (have) {
                st.pend = false;
                if (j == 0) {
                    *rec_cid = st.cell.cid;
                    rec->link[0] = st.cell.lk0;
                    rec->link[1] = st.cell.lk1;
                    rec->link[2] = st.cell.lk2;
                    rec->link[3] = st.cell.lk3;
                    rec->link[4] = st.cell.lk4;
                    rec->link[5] = st.cell.lk5;
                    rec->link[6] = st.cell.lk6;
                    rec->link[7] = st.cell.lk7;
                }
                const float ws = (last_sample_opaque && t + o.step_size > tmax) ? 1e9f : world_step;  // :93-95
                const float pcnt = ws * sigma;
                const float weight = expf(st.log_transmit) * (1.f - expf(-pcnt));
                st.log_transmit -= pcnt;
                const float ta = expf(st.log_transmit);
                smp[j].px = pos[0];
                smp[j].py = pos[1];
                smp[j].pz = pos[2];
                smp[j].w = weight;
                smp[j].ws = ws;
                smp[j].ta = ta;
                smp[j].sig = sigma;
                n = j + 1;
                if (ta < o.stop_thresh) {  // :107-110
                    st.log_transmit = -1e3f;
                    active = false;
                    collecting = false;
                }
            }
        }
    }
    return n;
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

__device__ __forceinline__ void cuvol_gather_init(CuvolGather& st) {
    st.cell.cid = -1;
    st.cell.lk0 = st.cell.lk1 = st.cell.lk2 = st.cell.lk3 = -1;
    st.cell.lk4 = st.cell.lk5 = st.cell.lk6 = st.cell.lk7 = -1;
    st.cell.d0 = st.cell.d1 = st.cell.d2 = st.cell.d3 = 0.f;
    st.cell.d4 = st.cell.d5 = st.cell.d6 = st.cell.d7 = 0.f;
    st.pend = false;
    st.pend_t = st.pend_px = st.pend_py = st.pend_pz = st.pend_sig = 0.f;
    st.log_transmit = 0.f;
}

// ------------------------------------------------------------------------------------
// Serving.  Two records are served at a time, one per half wave.  In a half, lane (ch, c) =
// (channel 0..2, corner 0..7; the fourth octet idles) holds the B coefficients of its channel in
// its corner's SH row (one contiguous 4B-byte load), reduces them against the ray's SH basis ONCE
// per record (the sample position does not enter), and a sample then costs one trilinear weight,
// one multiply and a 3-step DPP sum over the 8 corners.  Sample positions come in and colours go
// back through the record's LDS slots.
// ------------------------------------------------------------------------------------
struct alignas(16) CuvolRec {
    int link[8];
    int cid, n;
    float g0, g1, g2;  // d(loss)/d(rgb) of the ray (backward)
    int pad;
};

// The forward's: no loss gradient, 48 bytes -- with the samples as three floats the block's LDS is 36 KB + the coarse
// bitmap, four blocks a CU instead of three (the kernel waits for memory 56 % of its wave cycles).
struct alignas(16) CuvolRecF {
    int link[8];
    int cid, n, pad0, pad1;
};
struct CuvolF3 {
    float x, y, z;
};

template <int B>
struct CuvolLay {
    static constexpr int SP = (B + 3) / 4 * 4;  // LDS floats per ray for its SH basis
};

template <int B>
__device__ __forceinline__ void cuvol_load_rows(const KGrid& g, int link, int ch, float* val) {
    if (link >= 0) {
        const float* row = g.sh + ((size_t)link * (3 * B) + ch * B);
#pragma unroll
        for (int k = 0; k < B; ++k) val[k] = row[k];
    } else {
#pragma unroll
        for (int k = 0; k < B; ++k) val[k] = 0.f;
    }
}

// The ray of a lane: rpw rays a wave in batch order, or -- an image batch of width tile_w, 64 rays a wave (the host has
// checked that 8 x 8 tiles partition the image) -- the wave's 8 x 8 pixel tile.
__device__ __forceinline__ long long cuvol_lane_ray(int rpw, int tile_w) {
    const int tid = threadIdx.x;
    const long long wg = (long long)blockIdx.x * (kBlock / 64) + (tid >> 6);
    if (tile_w > 0) {
        // (the tile's width in the top byte: 4, 8 or 16 pixels; its height 64 / that)
        const int tw = tile_w >> 24, W = tile_w & 0xffffff, th = 64 / tw;
        const int tiles_x = W / tw, lane = tid & 63;
        const long long ty = wg / tiles_x, tx = wg - ty * tiles_x;
        return (ty * th + lane / tw) * W + tx * tw + (lane % tw);
    }
    return wg * rpw + (tid & 63);
}

// The lanes that hold a record this round, in the order of their cells (lane i: the i-th of them; wave_sort64,
// asurf_bwd_pipeline.h): records of one cell are then served one after the other -- its SH rows fetched once, its
// gradient rows written once -- wherever the rays that hold them sit in the wave.
__device__ __forceinline__ int cuvol_serve_order(bool has, int cid) {
    const int lane = threadIdx.x & 63;
    const unsigned long long key = has ? (((unsigned long long)(unsigned)cid << 6) | (unsigned)lane) : ~0ull;
    return (int)(wave_sort64(key) & 63ull);
}

template <int B, bool CLDS>
__global__ void __launch_bounds__(kBlock, (B <= 9 ? 4 : 2))
k_cuvol_fwd(KGrid g, KOpt o, int last_sample_opaque, int rpw, const float* __restrict__ origins,
            const float* __restrict__ dirs, long long Q, float* __restrict__ rgb_out,
            float* __restrict__ log_transmit_out, int tile_w) {
    constexpr int SP = CuvolLay<B>::SP;
    extern __shared__ unsigned int s_coarse[];
    __shared__ __attribute__((aligned(16))) float s_sph[kBlock * SP];
    __shared__ CuvolRecF s_rec[kBlock];
    __shared__ CuvolF3 s_smp[kBlock * kCuvolK];
    const int tid = threadIdx.x;
    if constexpr (CLDS) {
        for (int i = tid; i < g.coarse_words; i += kBlock) s_coarse[i] = g.coarse[i];
        __syncthreads();
    }
    // rpw rays per wave (their lanes 0 .. rpw-1 march; all 64 lanes serve): a wave works its records off one pair
    // after the other, so the length of that chain -- 64 rays x ~60 samples / 2 at ~1.5 us -- is the latency of a call
    // that does not fill the chip; small batches spread their rays over more, shorter-lived waves
    const long long q = cuvol_lane_ray(rpw, tile_w);
    const bool valid = (tid & 63) < rpw && q < Q;
    Ray r;
    float world_step;
    cuvol_ray(g, o, origins, dirs, q, valid, r, &world_step);
    {
        float sph[B];
        eval_sh<B>(r.d, sph);
#pragma unroll
        for (int k = 0; k < SP; ++k) s_sph[tid * SP + k] = k < B ? sph[k < B ? k : 0] : 0.f;
    }
    setup_ray(g, o, r);
    bool active = valid && !(r.tmin > r.tmax);
    const bool misses = valid && !active;
    March m;
    march_init(g, o, r, m);
    CuvolGather st;
    cuvol_gather_init(st);
    float outv[3] = {0.f, 0.f, 0.f};
    const int lane = tid & 63, wbase = tid & ~63;
    const int half = lane >> 5, ch = (lane >> 3) & 3, c = lane & 7;
    int held = -1;
    float val[B];
#pragma unroll
    for (int k = 0; k < B; ++k) val[k] = 0.f;
    while (__any(active)) {
        CuvolSmp smp[kCuvolK];
        LaneRec lr;
        int rec_cid = -1;
        const int n = cuvol_gather<CLDS>(g, o, s_coarse, m, st, r.tmax, world_step, last_sample_opaque, active,
                                         &lr, &rec_cid, smp);
        unsigned long long need = __ballot(n > 0);
        if (!need) continue;
        if (n > 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s_rec[tid].link[i] = lr.link[i];
            s_rec[tid].cid = rec_cid;
            s_rec[tid].n = n;
#pragma unroll
            for (int j = 0; j < kCuvolK; ++j) {
                CuvolF3 p3;
                p3.x = smp[j].px;
                p3.y = smp[j].py;
                p3.z = smp[j].pz;
                s_smp[tid * kCuvolK + j] = p3;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#ifdef ASURF_CUVOL_NOSERVE
        need = 0;
#endif
        // (in cell order for a tile of rays; a row of 64 rays meets its cells in lane order as it is -- there the sort only
        // costs: forward 7.9 -> 8.3 ms on the bench scene)
        const int order = tile_w > 0 ? cuvol_serve_order(n > 0, rec_cid) : 0;
        const int n_rec = __popcll(need);
        for (int t = 0; t < n_rec; t += 2) {
            bool two = t + 1 < n_rec;
            int r0, r1;
            if (tile_w > 0) {
                r0 = __builtin_amdgcn_readlane(order, t);
                r1 = two ? __builtin_amdgcn_readlane(order, t + 1) : r0;
            } else {
                next_pair(need, &r0, &r1, &two);
            }
            const int rr = wbase + (half ? r1 : r0);
            const bool mine = ch < 3 && (half == 0 || two);
            const int cid = s_rec[rr].cid, nr = mine ? s_rec[rr].n : 0;
            if (mine && cid != held) {
                held = cid;
                cuvol_load_rows<B>(g, s_rec[rr].link[c], ch, val);
            }
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < B; ++k) d = fmaf(s_sph[rr * SP + k], val[k], d);
            const int jmax = max(bcast_i(n, r0), bcast_i(n, r1));
#pragma unroll
            for (int j = 0; j < kCuvolK; ++j) {
                if (j < jmax) {
                    const CuvolF3 p = s_smp[rr * kCuvolK + j];
                    float a = j < nr ? corner_w(c << 0, p.x, p.y, p.z) * d : 0.f;
                    a = dpp_sum8(a);
                    if (c == 7 && j < nr) reinterpret_cast<float*>(&s_smp[rr * kCuvolK + j])[ch] = a;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int j = 0; j < kCuvolK; ++j) {
            if (j < n) {
                const CuvolF3 col = s_smp[tid * kCuvolK + j];
                outv[0] += smp[j].w * fmaxf(col.x + 0.5f, 0.f);
                outv[1] += smp[j].w * fmaxf(col.y + 0.5f, 0.f);
                outv[2] += smp[j].w * fmaxf(col.z + 0.5f, 0.f);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (valid) {
        const float bg = o.background_brightness;
        const float tr = expf(st.log_transmit);
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) rgb_out[3 * q + c3] = misses ? bg : outv[c3] + tr * bg;
        if (log_transmit_out != nullptr) log_transmit_out[q] = misses ? 0.f : st.log_transmit;
    }
}

// ------------------------------------------------------------------------------------
// Backward.  Phase A (per pair, the layout above): colours, clamp masks, and per lane (ch, c)
// A = sum_j w_c(pos_j) * weight_j * in01_j[ch] * gout[ch].  Phase B (per record, lanes = corner x
// SH coefficient as in coop_shade_bwd): grad_sh[c][e] += A[ch(e)][c] * sph[k(e)], accumulated in
// registers while consecutive records name the same cell and flushed as row-contiguous atomics.
// ------------------------------------------------------------------------------------
// (three waves a SIMD: left to itself the compiler takes 172 registers, four over what three waves allow -- the kernel
// waits for memory 57 % of its wave cycles, every wave more hides some of it)
template <int B, bool CLDS>
__global__ void __launch_bounds__(kBlock, (B <= 9 ? 3 : 2))
k_cuvol_bwd(KGrid g, KOpt o, int last_sample_opaque, int rpw, KGrads G, const float* __restrict__ origins,
            const float* __restrict__ dirs, long long Q, const float* __restrict__ grad_in,
            const float* __restrict__ color_cache, int grad_in_is_rgb,
            const float* __restrict__ log_transmit_in, float beta_loss, float sparsity_loss, int tile_w) {
    using C = Coop<B>;
    constexpr int SP = CuvolLay<B>::SP;
    extern __shared__ unsigned int s_coarse[];
    __shared__ __attribute__((aligned(16))) float s_sph[kBlock * SP];
    __shared__ CuvolRec s_rec[kBlock];
    __shared__ float4 s_smp[kBlock * kCuvolK];
    __shared__ float s_carry[(kBlock / 64) * 8 * C::D];  // per wave: SH sums changing corner slots (below)
    const int tid = threadIdx.x;
    if constexpr (CLDS) {
        for (int i = tid; i < g.coarse_words; i += kBlock) s_coarse[i] = g.coarse[i];
        __syncthreads();
    }
    // rpw rays per wave (their lanes 0 .. rpw-1 march; all 64 lanes serve): a wave works its records off one pair
    // after the other, so the length of that chain -- 64 rays x ~60 samples / 2 at ~1.5 us -- is the latency of a call
    // that does not fill the chip; small batches spread their rays over more, shorter-lived waves
    const long long q = cuvol_lane_ray(rpw, tile_w);
    const bool valid = (tid & 63) < rpw && q < Q;
    Ray r;
    float world_step;
    cuvol_ray(g, o, origins, dirs, q, valid, r, &world_step);
    {
        float sph[B];
        eval_sh<B>(r.d, sph);
#pragma unroll
        for (int k = 0; k < SP; ++k) s_sph[tid * SP + k] = k < B ? sph[k < B ? k : 0] : 0.f;
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
    s_rec[tid].g0 = gout[0];
    s_rec[tid].g1 = gout[1];
    s_rec[tid].g2 = gout[2];
    March m;
    march_init(g, o, r, m);
    CuvolGather st;
    cuvol_gather_init(st);
    const int lane = tid & 63, wbase = tid & ~63;
    const int half = lane >> 5, ch = (lane >> 3) & 3, c = lane & 7;
    // phase B layout
    const int cs = lane / C::D;
    const int e = lane - cs * C::D;
    const int chB = e / B;
    const int kB = e - chB * B;
    const bool onB = cs < C::CPR;
    // density gradient of the previous record of this lane's ray, not yet written (see below)
    bool pd_valid = false;
    int pd_cid = 0;
    int pd_link[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    float pd_acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float pd_keep[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int held = -1;
    float val[B];
#pragma unroll
    for (int k = 0; k < B; ++k) val[k] = 0.f;
    while (__any(active)) {
        CuvolSmp smp[kCuvolK];
        LaneRec lr;
        int rec_cid = -1;
        const int n = cuvol_gather<CLDS>(g, o, s_coarse, m, st, r.tmax, world_step, last_sample_opaque, active,
                                         &lr, &rec_cid, smp);
        unsigned long long need = __ballot(n > 0);
        if (!need) continue;
        if (n > 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) s_rec[tid].link[i] = lr.link[i];
            s_rec[tid].cid = rec_cid;
            s_rec[tid].n = n;
#pragma unroll
            for (int j = 0; j < kCuvolK; ++j)
                s_smp[tid * kCuvolK + j] = make_float4(smp[j].px, smp[j].py, smp[j].pz, smp[j].w);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int heldB = -1;
        float gacc[C::ROUNDS];
        int link_h[C::ROUNDS];
#pragma unroll
        for (int rd = 0; rd < C::ROUNDS; ++rd) {
            gacc[rd] = 0.f;
            link_h[rd] = -1;
        }
        auto flush = [&]() {
            if (G.mask != nullptr && e == 0) {  // every corner row of a served cell is touched (:120-135)
#pragma unroll
                for (int rd = 0; rd < C::ROUNDS; ++rd)
                    if (link_h[rd] >= 0) {
                        G.mask[link_h[rd]] = 1;
                        if (G.mask2 != nullptr) G.mask2[link_h[rd]] = 1;
                    }
            }
            if (G.grad_sh != nullptr) {
#pragma unroll
                for (int rd = 0; rd < C::ROUNDS; ++rd) {
                    if (link_h[rd] >= 0 && gacc[rd] != 0.f)
                        atomicAdd(&G.grad_sh[(size_t)link_h[rd] * C::D + e], gacc[rd]);
                }
            }
        };
        // (in cell order for a tile of rays; a row of 64 rays meets its cells in lane order as it is -- there the sort only
        // costs: forward 7.9 -> 8.3 ms on the bench scene)
        const int order = tile_w > 0 ? cuvol_serve_order(n > 0, rec_cid) : 0;
        const int n_rec = __popcll(need);
        for (int t = 0; t < n_rec; t += 2) {
            bool two = t + 1 < n_rec;
            int r0, r1;
            if (tile_w > 0) {
                r0 = __builtin_amdgcn_readlane(order, t);
                r1 = two ? __builtin_amdgcn_readlane(order, t + 1) : r0;
            } else {
                next_pair(need, &r0, &r1, &two);
            }
            // ---- phase A ----
            const int rr = wbase + (half ? r1 : r0);
            const bool mine = ch < 3 && (half == 0 || two);
            const int cid = s_rec[rr].cid, nr = mine ? s_rec[rr].n : 0;
            if (mine && cid != held) {
                held = cid;
                cuvol_load_rows<B>(g, s_rec[rr].link[c], ch, val);
            }
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < B; ++k) d = fmaf(s_sph[rr * SP + k], val[k], d);
            const float gch = ch == 0 ? s_rec[rr].g0 : ch == 1 ? s_rec[rr].g1 : s_rec[rr].g2;
            const int jmax = max(bcast_i(n, r0), bcast_i(n, r1));
            float A = 0.f;
#pragma unroll
            for (int j = 0; j < kCuvolK; ++j) {
                if (j < jmax) {
                    const float4 p = s_smp[rr * kCuvolK + j];
                    const float w = j < nr ? corner_w(c, p.x, p.y, p.z) : 0.f;
                    float a = dpp_sum8(w * d);
                    // every lane of the octet needs the channel sum (held by its lane 7)
                    a = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane | 7) << 2, __builtin_bit_cast(int, a)));
                    const float sc = a + 0.5f;
                    const float tcl = fmaxf(sc, 0.f);
                    const float in01 = tcl == sc ? 1.f : 0.f;
                    A = fmaf(w, p.w * in01 * gch, A);
                    if (c == 7 && j < nr) reinterpret_cast<float*>(&s_smp[rr * kCuvolK + j])[ch] = tcl * gch;
                }
            }
            // ---- phase B ----
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (h == 0 || two) {
                    const int rb = wbase + (h ? r1 : r0);
                    const int cidb = s_rec[rb].cid;  // wave-uniform
                    if (cidb != heldB) {
                        // a cell that shares a face with the held one: the four rows of that face keep accumulating
                        // (they are corners c ^ b of the new cell), only the far face is written out (as k_bwd_sh)
                        const int diff = cidb - heldB;
                        const int ad = diff < 0 ? -diff : diff;
                        const int b = heldB < 0 ? 0 : (ad == 1 ? 1 : (ad == g.offy ? 2 : (ad == g.offx ? 4 : 0)));
                        if (b != 0) {
                            const bool high = diff > 0;
                            float* carry = s_carry + (tid >> 6) * (8 * C::D);
#pragma unroll
                            for (int rd = 0; rd < C::ROUNDS; ++rd) {
                                const int cb = rd * C::CPR + cs;
                                if (onB && cb < 8) {
                                    const bool shared = ((cb & b) != 0) == high;
                                    if (!shared && link_h[rd] >= 0) {
                                        if (G.mask != nullptr && e == 0) {
                                            G.mask[link_h[rd]] = 1;
                                            if (G.mask2 != nullptr) G.mask2[link_h[rd]] = 1;
                                        }
                                        if (G.grad_sh != nullptr && gacc[rd] != 0.f)
                                            atomicAdd(&G.grad_sh[(size_t)link_h[rd] * C::D + e], gacc[rd]);
                                    }
                                    carry[cb * C::D + e] = shared ? gacc[rd] : 0.f;
                                }
                            }
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                            for (int rd = 0; rd < C::ROUNDS; ++rd) {
                                const int cb = rd * C::CPR + cs;
                                gacc[rd] = (onB && cb < 8) ? carry[(cb ^ b) * C::D + e] : 0.f;
                            }
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        } else {
                            flush();
#pragma unroll
                            for (int rd = 0; rd < C::ROUNDS; ++rd) gacc[rd] = 0.f;
                        }
                        heldB = cidb;
#pragma unroll
                        for (int rd = 0; rd < C::ROUNDS; ++rd) {
                            const int cb = rd * C::CPR + cs;
                            link_h[rd] = (onB && cb < 8) ? s_rec[rb].link[cb] : -1;
                        }
                    }
                    const float sphk = onB ? s_sph[rb * SP + kB] : 0.f;
#pragma unroll
                    for (int rd = 0; rd < C::ROUNDS; ++rd) {
                        const int cb = rd * C::CPR + cs;
                        const int src = h * 32 + chB * 8 + (cb & 7);
                        const float Av = __builtin_bit_cast(
                            float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, A)));
                        if (onB && cb < 8) gacc[rd] = fmaf(Av, sphk, gacc[rd]);
                    }
                }
            }
        }
        flush();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (n > 0) {
            float cw[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < kCuvolK; ++j) {
                if (j < n) {
                    const float4 tq = s_smp[tid * kCuvolK + j];
                    const float tc = (tq.x + tq.z) + tq.y;  // (c0 + c2) + c1, :469-471
                    accum -= smp[j].w * tc;
                    float grad_sigma = smp[j].ws * (tc * smp[j].ta - accum);  // :486-487
                    if (sparsity_loss > 0.f)
                        grad_sigma += sparsity_loss * (4 * smp[j].sig / (1 + 2 * (smp[j].sig * smp[j].sig)));
                    const float pos[3] = {smp[j].px, smp[j].py, smp[j].pz};
                    float w8[8];
                    corner_weights(pos, grad_sigma, w8);
#pragma unroll
                    for (int i = 0; i < 8; ++i) cw[i] += w8[i];
                }
            }
            if (G.grad_density != nullptr) {
                // A ray's next record usually lies in a cell that shares a face with this one: the four rows of
                // that face receive both records' terms in one atomic.  The previous record's sums wait in
                // registers (pd_*); its far face is written out now, its near face is folded into this record.
                const int diff = rec_cid - pd_cid;
                const bool px = diff == g.offx, mx = diff == -g.offx, py = diff == g.offy, my = diff == -g.offy;
                const bool pz = diff == 1, mz = diff == -1;
                const bool adjacent = pd_valid && (px || mx || py || my || pz || mz);
                // bit of the axis moved along, and whether the old cell's shared face is its high one
                const int b = (px || mx) ? 4 : ((py || my) ? 2 : 1);
                const bool high = px || py || pz;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    // old corner i lies on the shared face iff its bit b equals `high`; it is the new corner i ^ b
                    const bool shared = adjacent && (((i & 4) && b == 4) || ((i & 2) && b == 2) || ((i & 1) && b == 1)) == high;
                    if (pd_valid && !shared && pd_link[i] >= 0) atomicAdd(&G.grad_density[pd_link[i]], pd_acc[i]);
                    pd_keep[i] = shared ? pd_acc[i] : 0.f;
                }
                // new corner j receives the old corner j ^ b when that one was shared
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float carry = b == 4 ? pd_keep[j ^ 4] : (b == 2 ? pd_keep[j ^ 2] : pd_keep[j ^ 1]);
                    pd_acc[j] = cw[j] + (adjacent ? carry : 0.f);
                    pd_link[j] = lr.link[j];
                }
                pd_cid = rec_cid;
                pd_valid = true;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    if (pd_valid && G.grad_density != nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (pd_link[i] >= 0) atomicAdd(&G.grad_density[pd_link[i]], pd_acc[i]);
    }
}

// ------------------------------------------------------------------------------------
// Depth renders of the cuvol march (render_lerp_kernel_cuvol.cu:127-370, kernels :929-1020): one ray
// per lane, no colours.  MODE 0 expected termination, 1 depth of the heaviest sample (both 0 unless
// the accumulated weight exceeds `param`), 2 the first max_sample (depth, sigma) pairs, 3 the first
// sample whose density exceeds `param` (the host passes it in o.sigma_thresh, so the occupancy masks
// are built for it).
// ------------------------------------------------------------------------------------
template <int MODE, bool CLDS>
__global__ void __launch_bounds__(kBlock)
k_cuvol_eval(KGrid g, KOpt o, const float* __restrict__ origins, const float* __restrict__ dirs, long long Q,
             float param, int max_sample, float* __restrict__ out0, float* __restrict__ out1) {
    extern __shared__ unsigned int s_coarse[];
    const int tid = threadIdx.x;
    if constexpr (CLDS) {
        for (int i = tid; i < g.coarse_words; i += kBlock) s_coarse[i] = g.coarse[i];
        __syncthreads();
    }
    const long long q = (long long)blockIdx.x * kBlock + tid;
    if (q >= Q) return;
    Ray r;
    float world_step;
    cuvol_ray(g, o, origins, dirs, q, true, r, &world_step);
    setup_ray(g, o, r);
    if (r.tmin > r.tmax) {
        if (MODE != 2) out0[q] = 0.f;
        return;
    }
    March m;
    march_init(g, o, r, m);
    CuvolGather st;
    cuvol_gather_init(st);
    float outv = 0.f, weight_acc = 0.f, max_weight = -1.f, log_transmit = 0.f;
    int sample_i = 0;
    int budget = 1 << 24;  // lattice points: tmax <= 2e3 bounds the march, this bounds a defect
    while (true) {
        float t, pos[3], sigma;
        if (cuvol_next<CLDS>(g, o, s_coarse, m, st.cell, &t, pos, &sigma, budget) != 1) break;
        const float depth = (t / o.step_size) * world_step;
        if (MODE == 3) {
            outv = depth;
            break;
        }
        const float pcnt = world_step * sigma;
        const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
        log_transmit -= pcnt;
        if (MODE == 0) {
            outv += weight * depth;
            weight_acc += weight;
        } else if (MODE == 1) {
            weight_acc += weight;
            if (weight > max_weight) {
                max_weight = weight;
                outv = depth;
            }
        } else if (sample_i < max_sample) {
            out0[(size_t)q * max_sample + sample_i] = depth;
            out1[(size_t)q * max_sample + sample_i] = sigma;
            ++sample_i;
        }
        if (expf(log_transmit) < o.stop_thresh) break;
    }
    if (MODE == 3) out0[q] = outv;
    if (MODE == 0 || MODE == 1) out0[q] = weight_acc > param ? outv : 0.f;
}

}  // namespace asurf
