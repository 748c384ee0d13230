// asurf_sweep.h -- k_sweep: which flagged voxels lie along a ray, without stepping through the grid.
//
// The reference's traversal (surf_trav.cu:86-197) is a serial chain: a few hundred dependent steps per ray,
// most of them through voxels that hold nothing.  k_walk (asurf_pipeline.h) reproduces that chain step by
// step (jumping over empty bricks); its duration is the chain's, however few rays a call carries.  Here the
// chain is gone.  Along its dominant axis a (|d_a| largest) a ray crosses every voxel slab a = i exactly
// once, and WHERE it crosses is closed form: between the planes a = i and a = i + 1 the other two
// coordinates move by at most one voxel each, so the voxels the ray can touch in that slab are the <= 2 x 2
// block around the two plane intersections.  Slabs are independent of each other, so they are spread over
// lanes; the occupancy pyramid is consulted per 8-slab brick layer first, and the per-voxel bit tests are
// 64-bit mask operations on one word per slab (the fine plane is stored once per orientation, word = slab
// index along a, so that the block is a rows x columns rectangle of one word whatever a is).
//
// Exactness.  The block is computed in float arithmetic about the TRUE ray and widened by 1/256 voxel (float
// error of the plane intersections is < 1e-3 voxel at grid sizes <= 1024): it contains every flagged voxel
// the reference's float walk visits, and a few it does not (7 % on the bench scene).  Whether the walk visits
// a voxel is then decided exactly and in closed form (walk_visits, asurf_kernels.h: the walk is a merge of
// three increasing sequences of plane-crossing times, so "visited" is a comparison of the voxel's own six
// crossing times, the reference's float expressions); the rest is dropped, so the tables come out identical
// to k_walk's.  Order: the walk is monotone along every axis, so any order that extends the component-wise
// one is the walk's order on the voxels it visits: entries are ranked by (a, b, c) index in travel direction.
// The backward pass's early stop (k_walk's `cut` ordinal, surf_trav.cu:1925-1937) is closed form too
// (walk_cut_before) and only concerns voxels within step_size of the ray's end.
//
//   block = rays_per_block rays x LPR lanes (LPR = brick layers along the longest axis, rounded to 2^k)
//   phase 1  lane = (ray, brick layer): footprint of the ray over the layer's 8 slabs -> the <= 3 x 3 bricks
//            it can touch -> one task per brick whose coarse bit is set (LDS list)
//   phase 2  lane = task: the brick's 8 words (one 64-byte line), per slab the rectangle mask of the cells
//            between the ray's positions at the two slab planes, AND; hits are appended to the ray's LDS
//            buffer as sort keys
//   phase 3  lane = entry (packed over the block's rays): exact membership test, rejected entries marked
//   phase 4  per ray: counts, spill slice, the block's cells of the dense list (one atomic per block)
//   phase 5  lane = entry: rank among the ray's surviving keys = its slot; table / spill slice / dense list
//            written as k_walk does
// Rays that are not regular (non-finite origin or direction) or that collect more entries than their share K of the
// block's key buffer (at least 128) are handed to the one-kernel path (P.flagged), as k_walk does for rays it
// cannot place.
#pragma once
#include "asurf_pipeline.h"

namespace asurf {

constexpr int kSweepLanes = 512;    // (ray, brick layer) pairs per block: rays per block = kSweepLanes / LPR
constexpr int kSweepThreads = 256;  // threads per block (the pairs are strided over them)
constexpr int kSweepMaxThreads = 512;
constexpr int kSweepKeys = 2048;    // entries of all rays of a block kept in LDS, at least (K = keys / rays per block)
constexpr int kSweepMaxRays = 32;   // rays per block
constexpr unsigned kSweepCutBit = 1u << 30;   // key flag: the reference's backward pass stops before this voxel
constexpr unsigned kSweepDropped = 0xffffffffu;  // key of an entry the exact test rejected
// tasks per (ray, brick layer) pair a block has room for, x 1/2: a pair touches 2 x 2 bricks at most unless its
// footprint straddles a third one, and rarely are all of them occupied; a block that runs out takes its rays in
// four passes (9 / 4 <= 5 / 2)
constexpr int kSweepTasksPerLane2 = 5;

struct SweepRay {  // 144 bytes
    float o[3], tmin;           // the ray in grid coordinates (setup_ray), as the reference's walk sees it
    float d[3], tmax;
    int n0[3], pad2;            // first voxel of the walk
    float rd[3];                // correctly rounded 1 / d: the plane times of phase 3 come from quotient() when
    int fast_div;               // ... it is usable (Walk::fast_div, asurf_kernels.h)
    float oa, ob, oc, sb;
    float sc, ulo, uhi;
    int flags;                  // bits 0-1 a, bit 2/3/4 direction negative along a/b/c, bit 5 swept
    int size_a, size_b, size_c, na;   // na: brick layers that hold voxels along a
    int cs_a, cs_b, cs_c, state;      // coarse-bitmap strides; state 0 nothing to do, 1 swept, 2 handed back
    int bs_a, bs_b, bs_c, pad;        // brick strides of the fine planes
};

static_assert(kPipeC1 <= 64, "phase 4 holds one table slot per lane of a wave");
__device__ __forceinline__ int sweep_clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// diagnostics (phase timing; instrumented build only): ASURF_DEBUG = 256 * n makes the blocks leave after phase n - 1, their rays marked empty
__device__ __forceinline__ bool sweep_debug_leaves(const KOpt& o, int n, int rpb, long long q0, long long Q,
                                                   const PipeBuf& P) {
#ifndef ASURF_INSTRUMENT
    return false;
#endif
    if (((o.debug >> 8) & 7) != n) return false;
    const long long q = q0 + threadIdx.x;
    if ((int)threadIdx.x < rpb && q < Q) {
        P.n_items[q] = 0;
        P.cut[q] = 0x7fffffff;
        P.sp_base[q] = 0;
        P.cand_mask[q] = 0ull;
    }
    return true;
}

// LDS is what limits the number of resident blocks (every phase waits on memory or on a barrier, so the
// kernel's speed is the number of blocks a CU interleaves): everything sized per launch, the coarse plane is
// read from global memory (4 KB at 256^3: it lives in L1).
__global__ void __launch_bounds__(kSweepMaxThreads)
k_sweep(KGrid g, KOpt o, const float* __restrict__ origins, const float* __restrict__ dirs, long long Q,
        PipeBuf P, int lpr_log2, int rpb, int K, int want_cut, int task_cap, int n_layers) {
    extern __shared__ uint4 s_dyn4[];   // [rpb rays] [task_cap tasks] [rpb * K keys]
    SweepRay* s_ray = reinterpret_cast<SweepRay*>(s_dyn4);
    uint2* s_tasks = reinterpret_cast<uint2*>(s_ray + rpb);  // x: ray | layer | brick b | brick c, y: brick index
    unsigned* s_key = reinterpret_cast<unsigned*>(s_tasks + task_cap);
    __shared__ int s_cnt[kSweepMaxRays];      // entries found per ray
    __shared__ int s_n[kSweepMaxRays];        // entries placed per ray (0 when handed back)
    __shared__ int s_eend[kSweepMaxRays];     // inclusive prefix of the rays' found counts (entries in s_key)
    __shared__ int s_slot0[64];               // first list cell of each table slot, within the block's segment
    __shared__ int s_spb[kSweepMaxRays];      // spill base of the ray
    __shared__ int s_ntask, s_lbase;
    const TlScope tl(2);
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const long long q0 = (long long)blockIdx.x * rpb;
    const float kEps = 1.f / 256.f;
    Ray r0;
#pragma unroll
    for (int k = 0; k < 3; ++k) r0.o[k] = r0.d[k] = 0.f;
    if (tid < rpb && q0 + tid < Q) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r0.o[k] = origins[3 * (q0 + tid) + k];
            r0.d[k] = dirs[3 * (q0 + tid) + k];
        }
    }
    // ---- phase 0: one thread per ray ----
    if (tid == 0) s_ntask = 0;
    if (tid < rpb) {
        SweepRay R;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            R.o[k] = R.d[k] = 0.f;
            R.n0[k] = 0;
        }
        R.tmin = R.tmax = 0.f;
        R.pad2 = 0;
        R.rd[0] = R.rd[1] = R.rd[2] = 0.f;
        R.fast_div = 0;
        R.oa = R.ob = R.oc = R.sb = R.sc = 0.f;
        R.ulo = 1.f;
        R.uhi = 0.f;
        R.flags = 0;
        R.size_a = R.size_b = R.size_c = 2;
        R.na = 0;
        R.cs_a = R.cs_b = R.cs_c = R.bs_a = R.bs_b = R.bs_c = 0;
        R.state = 0;
        R.pad = 0;
        const long long q = q0 + tid;
        if (q < Q) {
            Ray r = r0;
            setup_ray(g, o, r);
            const bool hits_grid = !(r.tmin > r.tmax);
            if (hits_grid && !ray_is_regular(r)) {
                R.state = 2;
            } else if (hits_grid) {
                R.state = 1;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    R.o[k] = r.o[k];
                    R.d[k] = r.d[k];
                }
                R.tmin = r.tmin;
                R.tmax = r.tmax;
                walk_start(g, r, R.n0);
#pragma unroll
                for (int k = 0; k < 3; ++k) R.rd[k] = 1.f / r.d[k];
                R.fast_div = fmaxf(fmaxf(fabsf(R.rd[0]), fabsf(R.rd[1])), fabsf(R.rd[2])) < 1e30f ? 1 : 0;
                const float ax = fabsf(r.d[0]), ay = fabsf(r.d[1]), az = fabsf(r.d[2]);
                const int a = (ax >= ay && ax >= az) ? 0 : (ay >= az ? 1 : 2);
                // cyclic order: (a, b, c) = (0,1,2), (1,2,0), (2,0,1)
                const float da = a == 0 ? r.d[0] : (a == 1 ? r.d[1] : r.d[2]);
                const float db = a == 0 ? r.d[1] : (a == 1 ? r.d[2] : r.d[0]);
                const float dc = a == 0 ? r.d[2] : (a == 1 ? r.d[0] : r.d[1]);
                R.oa = a == 0 ? r.o[0] : (a == 1 ? r.o[1] : r.o[2]);
                R.ob = a == 0 ? r.o[1] : (a == 1 ? r.o[2] : r.o[0]);
                R.oc = a == 0 ? r.o[2] : (a == 1 ? r.o[0] : r.o[1]);
                R.sb = db / da;
                R.sc = dc / da;
                const float u0 = fmaf(r.tmin, da, R.oa), u1 = fmaf(r.tmax, da, R.oa);
                R.ulo = fminf(u0, u1) - 2.f * kEps;
                R.uhi = fmaxf(u0, u1) + 2.f * kEps;
                R.flags = a | (da < 0.f ? 4 : 0) | (db < 0.f ? 8 : 0) | (dc < 0.f ? 16 : 0) | 32;
                const int sx = g.size[0], sy = g.size[1], sz = g.size[2];
                R.size_a = a == 0 ? sx : (a == 1 ? sy : sz);
                R.size_b = a == 0 ? sy : (a == 1 ? sz : sx);
                R.size_c = a == 0 ? sz : (a == 1 ? sx : sy);
                R.na = ((R.size_a - 2) >> 3) + 1;
                // coarse bitmap: bit index = (bx * nb1 + by) * (nwz * 32) + bz
                const int cX = g.nb[1] * g.nwz * 32, cY = g.nwz * 32, cZ = 1;
                R.cs_a = a == 0 ? cX : (a == 1 ? cY : cZ);
                R.cs_b = a == 0 ? cY : (a == 1 ? cZ : cX);
                R.cs_c = a == 0 ? cZ : (a == 1 ? cX : cY);
                const int bX = g.nb[1] * g.nb[2], bY = g.nb[2], bZ = 1;
                R.bs_a = a == 0 ? bX : (a == 1 ? bY : bZ);
                R.bs_b = a == 0 ? bY : (a == 1 ? bZ : bX);
                R.bs_c = a == 0 ? bZ : (a == 1 ? bX : bY);
            }
        }
        s_ray[tid] = R;
        s_cnt[tid] = 0;
    }
    __syncthreads();
    if (sweep_debug_leaves(o, 1, rpb, q0, Q, P)) return;

    // Phases 1 and 2 run once over all rays of the block; if the task list would not hold them (scenes where
    // most bricks hold a flagged voxel) they run again in four passes over a quarter of the rays each (one ray a pass
    // where the brick layers are so many that a quarter could still overflow), which
    // fits whatever the scene: (pairs per block / 4) x 9 bricks <= task_cap.
    int passes = 1;
    for (int pass = 0; pass < passes; ++pass) {
    const int r_lo = (pass * rpb) / passes, r_hi = ((pass + 1) * rpb) / passes;
    // ---- phase 1: (ray, brick layer) -> tasks ----
    // (pairs = rays x the layers of the longest axis, not rounded up to a power of two: 96 at 768^3, 48 at 384^3;
    // the quotient through a float reciprocal is exact for these sizes, pairs < 2^12 and layers <= 128)
    const float inv_layers = 1.f / (float)n_layers;
    for (int lt = tid; lt < rpb * n_layers; lt += nthreads) {
        const int rl = (int)(((float)lt + 0.5f) * inv_layers), A = lt - rl * n_layers;
        if (rl >= r_lo && rl < r_hi) {
            const SweepRay R = s_ray[rl];
            if (R.state == 1 && A < R.na) {
                float L = (A == 0) ? -1e30f : (float)(8 * A) - kEps;
                float U = (A == R.na - 1) ? 1e30f : (float)(8 * A + 8) + kEps;
                L = fmaxf(L, R.ulo);
                U = fminf(U, R.uhi);
                if (L <= U) {
                    const float bL = fmaf(L - R.oa, R.sb, R.ob), bU = fmaf(U - R.oa, R.sb, R.ob);
                    const float cL = fmaf(L - R.oa, R.sc, R.oc), cU = fmaf(U - R.oa, R.sc, R.oc);
                    const int bl = sweep_clampi((int)floorf(fminf(bL, bU) - kEps), 0, R.size_b - 2) >> 3;
                    const int bh = sweep_clampi((int)floorf(fmaxf(bL, bU) + kEps), 0, R.size_b - 2) >> 3;
                    const int cl = sweep_clampi((int)floorf(fminf(cL, cU) - kEps), 0, R.size_c - 2) >> 3;
                    const int ch = sweep_clampi((int)floorf(fmaxf(cL, cU) + kEps), 0, R.size_c - 2) >> 3;
                    for (int Bb = bl; Bb <= bh; ++Bb) {
                        for (int Bc = cl; Bc <= ch; ++Bc) {
                            // (24-bit multiplies: full rate; every product is below 2^24 at grid sizes <= 1024)
                            const int lin = __mul24(A, R.cs_a) + __mul24(Bb, R.cs_b) + __mul24(Bc, R.cs_c);
                            if ((g.coarse[lin >> 5] >> (lin & 31)) & 1u) {
                                const int pos = atomicAdd(&s_ntask, 1);
                                if (pos < task_cap)
                                    s_tasks[pos] = make_uint2(
                                        ((unsigned)rl << 27) | ((unsigned)A << 18) | ((unsigned)Bb << 9) | (unsigned)Bc,
                                        (unsigned)(__mul24(A, R.bs_a) + __mul24(Bb, R.bs_b) + __mul24(Bc, R.bs_c)));
                            }
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    if (passes == 1 && rpb >= 2 && s_ntask > task_cap) {  // (block-uniform) start over, a quarter of the rays at a time
        __syncthreads();
        if (tid == 0) s_ntask = 0;
        __syncthreads();
        passes = (((rpb / 4) << lpr_log2) * 9 <= task_cap) ? min(4, rpb) : rpb;  // (a pass always fits: 9 bricks a pair)
        pass = -1;
        continue;
    }

    if (sweep_debug_leaves(o, 2, rpb, q0, Q, P)) return;
    // ---- phase 2: lane = task.  The brick's 8 words (one 64-byte line); the ray's (b, c) position at the 9 slab
    // planes, each shared by two slabs; per slab the rectangle mask of the cells between two planes ----
    const int ntask = min(s_ntask, task_cap);
    for (int ti = tid; ti < ntask; ti += nthreads) {
        const uint2 task2 = s_tasks[ti];
        const unsigned task = task2.x;
        const int rl = (int)(task >> 27), A = (int)((task >> 18) & 511), Bb = (int)((task >> 9) & 511),
                  Bc = (int)(task & 511);
        const SweepRay R = s_ray[rl];
        const int a = R.flags & 3;
        const unsigned long long* plane = a == 0 ? g.fine : (a == 1 ? g.fine_y : g.fine_z);
        const ulonglong2* wp = reinterpret_cast<const ulonglong2*>(plane + (size_t)task2.y * 8);
        const ulonglong2 w01 = wp[0], w23 = wp[1], w45 = wp[2], w67 = wp[3];
        const unsigned long long words[8] = {w01.x, w01.y, w23.x, w23.y, w45.x, w45.y, w67.x, w67.y};
        const int i0 = 8 * A, last = R.size_a - 2;
        // brick-local coordinates; the lattice's clamp [0, size - 2] in them
        const float bo = R.ob - (float)(8 * Bb), co = R.oc - (float)(8 * Bc);
        const int bmin = -8 * Bb, bmax = R.size_b - 2 - 8 * Bb, cmin = -8 * Bc, cmax = R.size_c - 2 - 8 * Bc;
        const float kEps2 = 2.f * kEps;  // kEps on the plane's position along a (slopes <= 1) + kEps on b / c
        const int rka = (R.flags & 4) ? 1023 : 0, rkb = (R.flags & 8) ? 1023 : 0, rkc = (R.flags & 16) ? 1023 : 0;
        // plane 0 (the lattice's first slab reaches back to the ray's start, its last one on to the ray's end)
        float praw = (i0 == 0) ? -1e30f : (float)i0;
        float pu = fminf(fmaxf(praw, R.ulo), R.uhi);
        float pb = fmaf(pu - R.oa, R.sb, bo), pc = fmaf(pu - R.oa, R.sc, co);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k;
            const float qraw = (i >= last) ? 1e30f : (float)(i + 1);
            const float qu = fminf(fmaxf(qraw, R.ulo), R.uhi);
            const float qb = fmaf(qu - R.oa, R.sb, bo), qc = fmaf(qu - R.oa, R.sc, co);
            const bool live = (i <= last) && (words[k] != 0ull) && (praw <= R.uhi) && (qraw >= R.ulo);
            const int lb0 = max(sweep_clampi((int)floorf(fminf(pb, qb) - kEps2), bmin, bmax), 0);
            const int lb1 = min(sweep_clampi((int)floorf(fmaxf(pb, qb) + kEps2), bmin, bmax), 7);
            const int lc0 = max(sweep_clampi((int)floorf(fminf(pc, qc) - kEps2), cmin, cmax), 0);
            const int lc1 = min(sweep_clampi((int)floorf(fmaxf(pc, qc) + kEps2), cmin, cmax), 7);
            praw = qraw;
            pb = qb;
            pc = qc;
            if (!live || lb0 > lb1 || lc0 > lc1) continue;
            const unsigned long long rows = (2ull << (8 * lb1 + 7)) - (1ull << (8 * lb0));
            const unsigned pat = ((2u << lc1) - (1u << lc0)) & 0xffu;
            const unsigned rep = pat * 0x01010101u;
            unsigned long long h = words[k] & rows & (((unsigned long long)rep << 32) | rep);
            while (h) {
                const int bit = __ffsll((long long)h) - 1;
                h &= h - 1ull;
                const int ib = 8 * Bb + (bit >> 3), ic = 8 * Bc + (bit & 7);
                const unsigned key = ((unsigned)(rka ^ i) << 20) | ((unsigned)(rkb ^ ib) << 10) | (unsigned)(rkc ^ ic);
                const int slot = atomicAdd(&s_cnt[rl], 1);
                if (slot < K) s_key[rl * K + slot] = key;
            }
        }
    }
    __syncthreads();
    if (pass + 1 < passes) {
        if (tid == 0) s_ntask = 0;
        __syncthreads();
    }
    }  // passes

    if (sweep_debug_leaves(o, 3, rpb, q0, Q, P)) return;
    // ---- phase 3: exact membership, lane = entry.  s_eend: inclusive prefix of the rays' FOUND counts ----
    if (tid < 64) {
        const int found = (tid < rpb) ? min(s_cnt[tid], K) : 0;
        int incl = found;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (tid >= d) incl += t;
        }
        if (tid < rpb) {
            s_eend[tid] = incl;
            s_n[tid] = 0;
        }
    }
    __syncthreads();
    const int n_found = s_eend[rpb - 1];
    for (int t = tid; t < n_found; t += nthreads) {
        int rl = 0;  // first ray whose inclusive prefix exceeds t
#pragma unroll
        for (int step = kSweepMaxRays / 2; step >= 1; step >>= 1)
            if (rl + step <= rpb && s_eend[rl + step - 1] <= t) rl += step;
        const int found = min(s_cnt[rl], K);
        const int slot = t - (s_eend[rl] - found);
        unsigned key = s_key[rl * K + slot];
        const SweepRay R = s_ray[rl];
        const int f = R.flags, a = f & 3;
        const int ra = (int)(key >> 20), rb = (int)((key >> 10) & 1023), rc = (int)(key & 1023);
        const int ia = (f & 4) ? 1023 - ra : ra, ib = (f & 8) ? 1023 - rb : rb, ic = (f & 16) ? 1023 - rc : rc;
        // (a, b, c) -> (x, y, z)
        const int vv[3] = {a == 0 ? ia : (a == 1 ? ic : ib), a == 0 ? ib : (a == 1 ? ia : ic),
                           a == 0 ? ic : (a == 1 ? ib : ia)};
        Ray r;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r.o[k] = R.o[k];
            r.d[k] = R.d[k];
        }
        r.tmin = R.tmin;
        r.tmax = R.tmax;
        float tc[3], tf[3], t_entry;
        if (R.fast_div) {  // the IEEE quotients, three instructions each (quotient(), asurf_kernels.h)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int up = r.d[k] > 0.f ? 1 : 0;
                tc[k] = quotient((float)(vv[k] + 1 - up) - r.o[k], r.d[k], R.rd[k]);
                tf[k] = quotient((float)(vv[k] + up) - r.o[k], r.d[k], R.rd[k]);
            }
        } else {
            voxel_times(r, vv[0], vv[1], vv[2], tc, tf);
        }
        if (walk_visits(r, R.n0, vv, tc, tf, &t_entry)) {
            atomicAdd(&s_n[rl], 1);
            if (want_cut && (t_entry + o.step_size > r.tmax) &&
                walk_cut_before(g, r, o.step_size, R.n0, vv, tc))
                key |= kSweepCutBit;
        } else {
            key = kSweepDropped;
        }
        s_key[rl * K + slot] = key;
    }
    __syncthreads();

    if (sweep_debug_leaves(o, 4, rpb, q0, Q, P)) return;
    // ---- phase 4: per ray, counts and allocations ----
    if (tid < rpb) {
        const long long q = q0 + tid;
        const int state = s_ray[tid].state;
        int n = 0, spb = 0;
        if (q < Q) {
            bool handback = (state == 2) || s_cnt[tid] > K || s_ntask > task_cap;
            n = (state == 1 && !handback) ? s_n[tid] : 0;
            if (n > P.C1) {
                const int extra = n - P.C1;
                spb = atomicAdd(&P.counters[1], extra);
                if (spb + extra > P.sp_cap) {  // spill list exhausted
                    for (int e = spb; e < min(spb + extra, P.sp_cap); ++e) P.sp_ray[e] = -1;
                    handback = true;
                    n = 0;
                }
            }
            if (handback) P.flagged[atomicAdd(&P.counters[0], 1)] = (int)q;
            P.n_items[q] = handback ? -1 : n;
            P.cut[q] = 0x7fffffff;
            P.sp_base[q] = spb;
            P.cand_mask[q] = 0ull;
        }
        s_n[tid] = n;
        s_spb[tid] = spb;
    }
    __syncthreads();
    // (first wave) the block's cells of the dense list are laid out SLOT-major -- all rays' slot 0, then slot 1,
    // ... -- so that neighbouring lanes of k_geom hold neighbouring rays at the same depth, i.e. the same or
    // adjacent voxels (s_slot0[s]: first cell of slot s)
    if (tid < 64) {
        int c = 0;  // rays that fill slot `tid`
        if (tid < P.C1)
            for (int r2 = 0; r2 < rpb; ++r2) c += (s_n[r2] > tid) ? 1 : 0;
        int ci = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(ci, d);
            if (tid >= d) ci += t;
        }
        s_slot0[tid] = ci - c;
        const int cells = __shfl(ci, 63);
        if (tid == 0) {
            const int shard = list_segment(P, rpb);
            s_lbase = shard * P.list_cap +
                      (cells > 0 ? atomicAdd(&P.counters[kCtrListShard0 + shard * kCtrShardStride], cells) : 0);
        }
    }
    __syncthreads();

    if (sweep_debug_leaves(o, 5, rpb, q0, Q, P)) return;
    // ---- phase 5: per surviving entry, its rank among the ray's keys is its slot ----
    for (int t = tid; t < n_found; t += nthreads) {
        int rl = 0;
#pragma unroll
        for (int step = kSweepMaxRays / 2; step >= 1; step >>= 1)
            if (rl + step <= rpb && s_eend[rl + step - 1] <= t) rl += step;
        const int found = min(s_cnt[rl], K);
        const int slot = t - (s_eend[rl] - found);
        const unsigned* keys = s_key + rl * K;
        const unsigned key = keys[slot];
        if (key == kSweepDropped || s_n[rl] == 0) continue;  // s_n == 0: ray handed back
        const unsigned kk = key & ~kSweepCutBit;
        int ord = 0;  // (a dropped key has bit 31 set: larger than every kept one)
        for (int j = 0; j < found; ++j) ord += ((keys[j] & ~kSweepCutBit) < kk) ? 1 : 0;
        const int f = s_ray[rl].flags, a = f & 3;
        const int ra = (int)((kk >> 20) & 1023), rb = (int)((kk >> 10) & 1023), rc = (int)(kk & 1023);
        const int ia = (f & 4) ? 1023 - ra : ra, ib = (f & 8) ? 1023 - rb : rb, ic = (f & 16) ? 1023 - rc : rc;
        const int vx = a == 0 ? ia : (a == 1 ? ic : ib);
        const int vy = a == 0 ? ib : (a == 1 ? ia : ic);
        const int vz = a == 0 ? ic : (a == 1 ? ib : ia);
        const int voxel = pack_voxel(vx, vy, vz);
        const long long q = q0 + rl;
        if (key & kSweepCutBit) atomicMin(&P.cut[q], ord);  // rare
        if (ord < P.C1) {
            int before = 0;  // rays in front of this one that fill slot `ord` too
            for (int r2 = 0; r2 < rl; ++r2) before += (s_n[r2] > ord) ? 1 : 0;
            P.items[(size_t)ord * Q + q] = voxel;
            P.list[s_lbase + s_slot0[ord] + before] = ((unsigned)ord << kListRayBits) | (unsigned)q;
        } else {
            const int at = s_spb[rl] + ord - P.C1;
            P.sp_items[at] = voxel;
            P.sp_ray[at] = (int)q;
        }
    }
    tl.end();
}

}  // namespace asurf
