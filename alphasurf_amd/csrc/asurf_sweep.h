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


// ------------------------------------------------------------------------------------------------------------------
// k_sweep, wave-autonomous form (round 3).  Same search, same closed forms, same tables as the block form above; what
// changes is who waits for whom.  The block form strides the (ray, layer) pairs, the tasks and the hits of 16 rays over
// 256 threads with a barrier between phases: a hundred tasks occupy four waves at 40 % of their lanes, phases 0 and 4
// run on 16 of 256 threads while the rest wait, and every task and every hit takes a same-address LDS atomic with return.
// Here a WAVE owns its rays (16 of them) from the first load to the last store: its own slice of LDS, no block barrier
// anywhere (waves of a block never meet), and every list is built by ordered compaction -- per-lane counts, one DPP
// prefix scan, each lane writes at its offset:
//   * a hundred tasks are two full wave iterations instead of four partial ones (phase 2 and 3 halve);
//   * tasks come out in (ray, layer) order and hits in task order, so a ray's hits are CONTIGUOUS: no per-ray key
//     buffers (the rays share one buffer), no atomics with return;
//   * waves drift apart in time, so one wave's set-up or allocation round trip runs under another's bit tests.
// A wave whose tasks or hits do not fit takes its rays in smaller groups (a quarter, then one at a time); a single ray
// that does not fit on its own is handed to the one-kernel path.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kSweepWaveRays = 16;       // rays per wave, at most (fewer when the batch is small: sweep_wave_rays)
constexpr int kSweepWavesPerBlock = 1;   // (1 / 2 / 4 waves a block: 0.305 / 0.313 / 0.322 ms at config 2 -- a block's LDS is held until its last
                                         // wave ends; at 6 waves per SIMD -- 79 registers, 12 bytes of scratch -- 0.311)

struct SweepWaveMisc {
    int cnt[kSweepWaveRays];        // hits found per ray
    int start[kSweepWaveRays + 1];  // first hit of the ray in the wave's key buffer
    int n[kSweepWaveRays];          // hits that pass the exact test (0 when handed back)
    int spb[kSweepWaveRays];        // spill base
    int hand[kSweepWaveRays];       // 1: handed back
    int pad[3];
};
// (phases 4 and 5 keep two more tables of 64 words -- the first list cell of each table slot, the rays that fill it --
// where the task list was: tasks are dead once phase 2 has run, and task_cap >= 128)
static_assert(sizeof(SweepWaveMisc) % 16 == 0, "the waves' LDS slices stay 16-byte aligned");

// cross-lane hand-over through the wave's own LDS slice: LDS operations of a wave execute in order; the compiler must
// not move them across this point, and the data must have landed
__device__ __forceinline__ void sweep_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// inclusive prefix sum over the 64 lanes (all active): DPP row shifts + the two row broadcasts (wave_sum's sequence
// IS the scan: after the four shifts a lane holds the prefix within its row of 16, the broadcasts add the rows in front)
__device__ __forceinline__ int wave_incl_scan(int v) {
#define ASURF_DPP_IADD(ctrl, row_mask) v += __builtin_amdgcn_update_dpp(0, v, ctrl, row_mask, 0xf, false)
    ASURF_DPP_IADD(0x111, 0xf);
    ASURF_DPP_IADD(0x112, 0xf);
    ASURF_DPP_IADD(0x114, 0xf);
    ASURF_DPP_IADD(0x118, 0xf);
    ASURF_DPP_IADD(0x142, 0xa);
    ASURF_DPP_IADD(0x143, 0xc);
#undef ASURF_DPP_IADD
    return v;
}

__device__ __forceinline__ int list_segment_unit(const PipeBuf& P, unsigned unit, int rays_per_unit) {
    const unsigned nbb = (unsigned)(P.band_rays / rays_per_unit);
    const unsigned band = unit / nbb, in_band = unit - band * nbb;
    return (int)(band % kListGroups) * kListShards + (int)((in_band + band / kListGroups) % kListShards);
}

// 5 waves a SIMD: 89 registers, no scratch (left to itself the compiler takes 102 and 4 fit), and the wave's LDS slice at
// 256^3 -- 8 016 bytes -- lets 20 waves share a CU.  The kernel is half VALU issue and half the latency of dependent
// instructions with few waves to switch to: 4 / 5 / 6 (36 bytes of scratch) / 8 (96 bytes) waves: 0.294 / 0.281 / 0.284 /
// 0.322 ms at config 2.
#ifndef ASURF_SWEEP_EU
#define ASURF_SWEEP_EU 5
#endif
// The value gates a voxel passes before it is a work voxel (surf_trav.cu:230-239, 265-277), decided the way
// k_build_masks2 decides them for the per-call planes (asurf_masks2.h: live = not all 8 densities below the gate;
// work = live and some level neither above all 8 surface values nor below all of them) -- for the calls whose planes
// hold the links' test only (KOpt::lazy_gates).
__device__ __forceinline__ bool lazy_gates_pass(const KGrid& g, float sigma_thresh, const int32_t* lk) {
    bool ok = true, all_below = true;
    float sv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ok = ok && lk[i] >= 0;
    if (!ok) return false;   // (cannot happen under planes that match `links`: memory safety under a broken promise)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        all_below = all_below && (g.density[lk[i]] < sigma_thresh);
        sv[i] = g.surface[lk[i]];
    }
    if (all_below) return false;
    bool any = false;
    for (int s = 0; s < g.level_set_num; ++s) {
        const float lv = g.level_set[s];
        bool all_g = true, all_h = true;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            all_g = all_g && (sv[i] > lv);
            all_h = all_h && (sv[i] < lv);
        }
        any = any || (!all_g && !all_h);
    }
    return any;
}

// (the structs BY VALUE: bound by reference to the kernels' arguments they were kept in private memory -- 48 bytes of
// scratch a lane in k_sweep_wave where the kernel written out had 12; 0.2555 -> 0.2524 ms)
template <bool LAZY>
__device__ __forceinline__ void
sweep_wave_body(const KGrid g, const KOpt o, const float* __restrict__ origins, const float* __restrict__ dirs, long long Q,
                const PipeBuf P, int n_layers, int want_cut, int task_cap, int key_cap, int wave_bytes, int rpw) {
    extern __shared__ uint4 s_dyn4[];
    const TlScope tl(2);
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const unsigned unit = blockIdx.x * (blockDim.x >> 6) + wib;   // this wave's number
    const long long q0 = (long long)unit * rpw;
    if (q0 >= Q) return;   // (no barrier anywhere below: a wave may leave on its own)
    char* const base = reinterpret_cast<char*>(s_dyn4) + (size_t)wib * wave_bytes;
    SweepRay* const s_ray = reinterpret_cast<SweepRay*>(base);
    unsigned* const s_task = reinterpret_cast<unsigned*>(s_ray + rpw);   // ray << 21 | layer << 14 | brick b << 7 | brick c
    unsigned* const s_key = s_task + task_cap;
    unsigned char* const s_krl = reinterpret_cast<unsigned char*>(s_key + key_cap);   // the ray a key belongs to
    SweepWaveMisc* const M = reinterpret_cast<SweepWaveMisc*>(s_krl + key_cap);       // (key_cap: a multiple of 16)
    int* const s_slot0 = reinterpret_cast<int*>(s_task);   // phases 4-5: first list cell of each table slot
    unsigned* const s_fill = s_task + 64;                   // ... bit r: ray r fills the slot
    const float kEps = 1.f / 256.f;

    // ---- phase 0: lane = ray ----
    if (lane < rpw) {
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
        const long long q = q0 + lane;
        if (q < Q) {
            Ray r;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                r.o[k] = origins[3 * q + k];
                r.d[k] = dirs[3 * q + k];
            }
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
        s_ray[lane] = R;
        M->hand[lane] = (q < Q && R.state == 2) ? 1 : 0;
    }
    sweep_wave_sync();

#ifdef ASURF_INSTRUMENT
    // phase timing (instrumented build): ASURF_DEBUG = 256 * n makes every wave leave after phase n - 1, results invalid
#define ASURF_SWEEP_LEAVE(n_)                                  \
    if (((o.debug >> 8) & 7) == (n_)) {                        \
        if (lane < rpw && q0 + lane < Q) {                     \
            P.n_items[q0 + lane] = 0;                          \
            P.cut[q0 + lane] = 0x7fffffff;                     \
            P.sp_base[q0 + lane] = 0;                          \
            P.cand_mask[q0 + lane] = 0ull;                     \
        }                                                      \
        return;                                                \
    }
#else
#define ASURF_SWEEP_LEAVE(n_)
#endif
    ASURF_SWEEP_LEAVE(1)
    const char* const cbase = reinterpret_cast<const char*>(g.coarse);
#define ASURF_COARSE_WORD(lin_) (*reinterpret_cast<const unsigned*>(cbase + (((unsigned)(lin_) >> 5) << 2)))
    const float inv_layers = 1.f / (float)n_layers;
    const int layers_log2 = (n_layers & (n_layers - 1)) == 0 ? 31 - __clz(n_layers) : -1;
    int group = rpw;   // rays taken together; shrinks when the wave's lists overflow
    for (int r_lo = 0; r_lo < rpw;) {
        const int r_hi = min(r_lo + group, rpw);
        // ---- phase 1: lane = (ray, brick layer) -> tasks, in (ray, layer) order ----
        int ntask = 0;   // (wave-uniform)
        // one turn of 64 pairs: lane's pair is (ray rl, layer A), `in_range` false for lanes beyond the last pair
        auto pair_turn = [&](const int rl, const int A, const bool in_range) {
            unsigned m9 = 0u;   // which of the <= 3 x 3 bricks of the footprint hold a flagged voxel
            int bl = 0, cl = 0;
            {
                // Everything the pair needs of its ray is asked of LDS at once and every lane does the arithmetic (rl is
                // a ray of the wave whatever the lane): nested conditions made five dependent LDS round trips of this.
                const SweepRay& R = s_ray[rl];
                const int state = R.state, hand = M->hand[rl], na = R.na;
                const float ulo = R.ulo, uhi = R.uhi, oa = R.oa;
                const float sb = R.sb, sc = R.sc, ob = R.ob, oc = R.oc;
                const int size_b = R.size_b, size_c = R.size_c;
                const int cs_a = R.cs_a, cs_b = R.cs_b, cs_c = R.cs_c;
                float L = (A == 0) ? -1e30f : (float)(8 * A) - kEps;
                float U = (A == na - 1) ? 1e30f : (float)(8 * A + 8) + kEps;
                L = fmaxf(L, ulo);
                U = fminf(U, uhi);
                const bool on = in_range && state == 1 && hand == 0 && A < na && L <= U;
                const float bL = fmaf(L - oa, sb, ob), bU = fmaf(U - oa, sb, ob);
                const float cL = fmaf(L - oa, sc, oc), cU = fmaf(U - oa, sc, oc);
                bl = sweep_clampi((int)floorf(fminf(bL, bU) - kEps), 0, size_b - 2) >> 3;
                const int bh = sweep_clampi((int)floorf(fmaxf(bL, bU) + kEps), 0, size_b - 2) >> 3;
                cl = sweep_clampi((int)floorf(fminf(cL, cU) - kEps), 0, size_c - 2) >> 3;
                const int ch = sweep_clampi((int)floorf(fmaxf(cL, cU) + kEps), 0, size_c - 2) >> 3;
                if (on) {
                    const int lin0 = __mul24(A, cs_a) + __mul24(bl, cs_b) + __mul24(cl, cs_c);
                    // 0 .. 2: the footprint of one layer is at most 8 slabs + the box's half voxel + the margins long
                    // and the slopes are <= 1, so it spans less than two bricks' lengths
                    const int nb_ = min(bh - bl, 2), nc_ = min(ch - cl, 2);
                    // the 2 x 2 corner first, its four bits asked for together; a third row or column is rare.  (Byte
                    // offsets in 32 bits beside the bitmap's base: one address instruction a load instead of three.)
                    const int l00 = lin0, l01 = lin0 + cs_c, l10 = lin0 + cs_b, l11 = lin0 + cs_b + cs_c;
                    const unsigned w00 = ASURF_COARSE_WORD(l00);
                    const unsigned w01 = nc_ >= 1 ? ASURF_COARSE_WORD(l01) : 0u;
                    const unsigned w10 = nb_ >= 1 ? ASURF_COARSE_WORD(l10) : 0u;
                    const unsigned w11 = (nb_ >= 1 && nc_ >= 1) ? ASURF_COARSE_WORD(l11) : 0u;
                    m9 = ((w00 >> (l00 & 31)) & 1u) | (((w01 >> (l01 & 31)) & 1u) << 1) |
                         (((w10 >> (l10 & 31)) & 1u) << 3) | (((w11 >> (l11 & 31)) & 1u) << 4);
                    if (nb_ == 2 || nc_ == 2) {
                        for (int ib = 0; ib <= nb_; ++ib)
                            for (int ic = 0; ic <= nc_; ++ic) {
                                if (ib < 2 && ic < 2) continue;
                                const int lin = lin0 + __mul24(ib, cs_b) + __mul24(ic, cs_c);
                                m9 |= ((ASURF_COARSE_WORD(lin) >> (lin & 31)) & 1u) << (3 * ib + ic);
                            }
                    }
                }
            }
            const int cnt = __popc(m9);
            const int incl = wave_incl_scan(cnt);
            const int total = bcast_i(incl, 63);
            if (total == 0) return;   // (wave-uniform)
            int at = ntask + incl - cnt;
            if (ntask + total <= task_cap) {
                unsigned m = m9;
                while (m) {
                    const int bit = __ffs((int)m) - 1;
                    m &= m - 1u;
                    const int ib = bit / 3, ic = bit - 3 * ib;
                    s_task[at++] = ((unsigned)rl << 21) | ((unsigned)A << 14) | ((unsigned)(bl + ib) << 7) | (unsigned)(cl + ic);
                }
            }
            ntask += total;
        };
        // The rays of a wave are neighbours in the batch, and for a coherent batch neighbours in space: over a layer
        // their footprints share a brick or two, and most layers hold nothing for any of them (26 layers a ray on the
        // bench scene, 4 with a flagged brick).  So first lane = LAYER: the box around the footprints of all rays of the
        // group over that layer (a ray's own footprint is inside it: floor and the clamps are monotone) against the
        // bitmap -- a sixteenth of the pairs' arithmetic per (ray, layer) -- and then the pairs of the layers that are
        // left only, in (ray, layer) order as before.  Needs one dominant axis for the whole group; a box wider than 3 x 3
        // bricks (an incoherent batch) is not looked up, its layer counts as occupied.
        bool bundled = false;
        int nk = 0;   // layers left (wave-uniform)
        if (r_hi - r_lo >= 8 && n_layers <= key_cap) {
            int my_a = -1;
            if (lane >= r_lo && lane < r_hi && s_ray[lane].state == 1) my_a = s_ray[lane].flags & 3;
            const unsigned long long act = __ballot(my_a >= 0);
            if (act != 0ull) {
                const int a0 = bcast_i(my_a, __ffsll((long long)act) - 1);
                bundled = __ballot(my_a >= 0 && my_a != a0) == 0ull;
            }
            if (bundled) {
                const SweepRay& R0 = s_ray[__ffsll((long long)act) - 1];
                const int na = R0.na, size_b = R0.size_b, size_c = R0.size_c, cs_a = R0.cs_a, cs_b = R0.cs_b, cs_c = R0.cs_c;
                int* const s_layers = reinterpret_cast<int*>(s_key);   // (no keys yet)
                // lanes per layer: the rays of the group are dealt to `sub` lanes, combined by xor shuffles
                const int sub_log2 = n_layers <= 16 ? 2 : (n_layers <= 32 ? 1 : 0), sub = 1 << sub_log2;
                const int part = lane & (sub - 1);
                for (int A0 = 0; A0 < na; A0 += 64 >> sub_log2) {
                    const int A = A0 + (lane >> sub_log2);
                    const float Lr = (A == 0) ? -1e30f : (float)(8 * A) - kEps;
                    const float Ur = (A == na - 1) ? 1e30f : (float)(8 * A + 8) + kEps;
                    float bmin = 3e38f, bmax = -3e38f, cmin = 3e38f, cmax = -3e38f;
                    // (a ray that is not swept has ulo > uhi -- phase 0 -- and drops out by itself; one that was handed
                    // back only widens the box)
                    for (int r = r_lo + part; r < r_hi; r += sub) {
                        const SweepRay& R = s_ray[r];
                        const float L = fmaxf(Lr, R.ulo), U = fminf(Ur, R.uhi);
                        const bool have = L <= U;
                        const float oa = R.oa, sb = R.sb, sc = R.sc, ob = R.ob, oc = R.oc;
                        const float bL = fmaf(L - oa, sb, ob), bU = fmaf(U - oa, sb, ob);
                        const float cL = fmaf(L - oa, sc, oc), cU = fmaf(U - oa, sc, oc);
                        const float b0 = fminf(bmin, fminf(bL, bU)), b1 = fmaxf(bmax, fmaxf(bL, bU));
                        const float c0 = fminf(cmin, fminf(cL, cU)), c1 = fmaxf(cmax, fmaxf(cL, cU));
                        bmin = have ? b0 : bmin;
                        bmax = have ? b1 : bmax;
                        cmin = have ? c0 : cmin;
                        cmax = have ? c1 : cmax;
                    }
                    for (int m = 1; m < sub; m <<= 1) {
                        bmin = fminf(bmin, __shfl_xor(bmin, m));
                        bmax = fmaxf(bmax, __shfl_xor(bmax, m));
                        cmin = fminf(cmin, __shfl_xor(cmin, m));
                        cmax = fmaxf(cmax, __shfl_xor(cmax, m));
                    }
                    bool occupied = false;
                    if (A < na && bmin <= bmax) {
                        const int bl = sweep_clampi((int)floorf(bmin - kEps), 0, size_b - 2) >> 3;
                        const int bh = sweep_clampi((int)floorf(bmax + kEps), 0, size_b - 2) >> 3;
                        const int cl = sweep_clampi((int)floorf(cmin - kEps), 0, size_c - 2) >> 3;
                        const int ch = sweep_clampi((int)floorf(cmax + kEps), 0, size_c - 2) >> 3;
                        const int nb_ = bh - bl, nc_ = ch - cl;
                        if (nb_ > 2 || nc_ > 2) {
                            occupied = true;
                        } else {
                            // the 2 x 2 corner together, a third row or column when the box has one
                            const int lin0 = __mul24(A, cs_a) + __mul24(bl, cs_b) + __mul24(cl, cs_c);
                            const int l00 = lin0, l01 = lin0 + cs_c, l10 = lin0 + cs_b, l11 = lin0 + cs_b + cs_c;
                            const unsigned w00 = ASURF_COARSE_WORD(l00);
                            const unsigned w01 = nc_ >= 1 ? ASURF_COARSE_WORD(l01) : 0u;
                            const unsigned w10 = nb_ >= 1 ? ASURF_COARSE_WORD(l10) : 0u;
                            const unsigned w11 = (nb_ >= 1 && nc_ >= 1) ? ASURF_COARSE_WORD(l11) : 0u;
                            unsigned any = ((w00 >> (l00 & 31)) | (w01 >> (l01 & 31)) | (w10 >> (l10 & 31)) | (w11 >> (l11 & 31))) & 1u;
                            if (nb_ == 2 || nc_ == 2) {
                                for (int ib = 0; ib <= nb_; ++ib)
                                    for (int ic = 0; ic <= nc_; ++ic) {
                                        if (ib < 2 && ic < 2) continue;
                                        const int lin = lin0 + __mul24(ib, cs_b) + __mul24(ic, cs_c);
                                        any |= (ASURF_COARSE_WORD(lin) >> (lin & 31)) & 1u;
                                    }
                            }
                            occupied = any != 0u;
                        }
                    }
                    const unsigned long long mo = __ballot(occupied && part == 0);
                    if (occupied && part == 0) s_layers[nk + __popcll(mo & ((1ull << lane) - 1ull))] = A;
                    nk += __popcll(mo);
                }
                sweep_wave_sync();
                if (nk > 0) {
                    const int nr = r_hi - r_lo, npairs = nr * nk;
                    const float inv_nk = 1.f / (float)nk;
                    for (int lt0 = 0; lt0 < npairs; lt0 += 64) {
                        const int lt = lt0 + lane;
                        const int ri = min((int)(((float)lt + 0.5f) * inv_nk), nr - 1);
                        const int j = min(lt - ri * nk, nk - 1);
                        pair_turn(r_lo + ri, s_layers[j], lt < npairs);
                    }
                }
            }
        }
        if (!bundled) {
            const int pair_lo = r_lo * n_layers, pair_hi = r_hi * n_layers;
            for (int lt0 = pair_lo; lt0 < pair_hi; lt0 += 64) {
                const int lt = lt0 + lane;
                // (ray, layer) of the pair: a shift when the layers are a power of two (the product is a quarter-rate
                // instruction)
                int rl, A;
                if (layers_log2 >= 0) {
                    rl = min(lt >> layers_log2, rpw - 1);
                    A = lt - (rl << layers_log2);
                } else {
                    rl = min((int)(((float)lt + 0.5f) * inv_layers), rpw - 1);
                    A = lt - rl * n_layers;
                }
                pair_turn(rl, A, lt < pair_hi);
            }
        }
        if (ntask > task_cap) {   // (wave-uniform)
            if (group > 1) {
                group = max(1, group >> 2);
                continue;
            }
            if (lane == 0) M->hand[r_lo] = 1;   // one ray alone does not fit: the one-kernel path takes it
            sweep_wave_sync();
            ntask = 0;
        }
        sweep_wave_sync();

        ASURF_SWEEP_LEAVE(2)
        // ---- phase 1b: lane = task.  Two tasks in three find nothing: the brick holds flagged voxels, but not where the
        // ray passes.  One rectangle -- the ray's footprint over ALL of the brick's slabs, a superset of the slabs'
        // rectangles: the positions are monotone along the ray, so the planes at the brick's two ends bound the ones in
        // between -- against the brick's projection along a (the OR of its 8 words) removes half of the tasks for a
        // sixth of what a task costs; the survivors are compacted in place (ordered: ballot + prefix), so that phase 2
        // spends its 8 slabs per lane on tasks that can yield something. ----
        {
            int nkeep = 0;   // (wave-uniform)
            for (int t0 = 0; t0 < ntask; t0 += 64) {
                const int ti = t0 + lane;
                bool keep = false;
                unsigned task = 0u;
                if (ti < ntask) {
                    task = s_task[ti];
                    const int rl = (int)(task >> 21), A = (int)((task >> 14) & 127), Bb = (int)((task >> 7) & 127),
                              Bc = (int)(task & 127);
                    const SweepRay& R = s_ray[rl];
                    const int a = R.flags & 3;
                    const unsigned long long* plane = a == 0 ? g.fine : (a == 1 ? g.fine_y : g.fine_z);
                    const int brick = __mul24(A, R.bs_a) + __mul24(Bb, R.bs_b) + __mul24(Bc, R.bs_c);
                    const ulonglong2* wp = reinterpret_cast<const ulonglong2*>(plane + (size_t)brick * 8);
                    const ulonglong2 w01 = wp[0], w23 = wp[1], w45 = wp[2], w67 = wp[3];
                    const unsigned long long proj = (w01.x | w01.y) | (w23.x | w23.y) | (w45.x | w45.y) | (w67.x | w67.y);
                    const float ulo = R.ulo, uhi = R.uhi, oa = R.oa;
                    const int i0 = 8 * A, last = R.size_a - 2;
                    const float bo = R.ob - (float)(8 * Bb), co = R.oc - (float)(8 * Bc);
                    const int bmin = -8 * Bb, bmax = R.size_b - 2 - 8 * Bb, cmin = -8 * Bc, cmax = R.size_c - 2 - 8 * Bc;
                    const float kEps2 = 2.f * kEps;
                    // the planes at the brick's two ends, exactly as phase 2 forms them for its first and last slab
                    const float praw = (i0 == 0) ? -1e30f : (float)i0;
                    const float qraw = (i0 + 7 >= last) ? 1e30f : (float)(i0 + 8);
                    const float pu = fminf(fmaxf(praw, ulo), uhi), qu = fminf(fmaxf(qraw, ulo), uhi);
                    const float pb = fmaf(pu - oa, R.sb, bo), pc = fmaf(pu - oa, R.sc, co);
                    const float qb = fmaf(qu - oa, R.sb, bo), qc = fmaf(qu - oa, R.sc, co);
                    const int lb0 = max(sweep_clampi((int)floorf(fminf(pb, qb) - kEps2), bmin, bmax), 0);
                    const int lb1 = min(sweep_clampi((int)floorf(fmaxf(pb, qb) + kEps2), bmin, bmax), 7);
                    const int lc0 = max(sweep_clampi((int)floorf(fminf(pc, qc) - kEps2), cmin, cmax), 0);
                    const int lc1 = min(sweep_clampi((int)floorf(fmaxf(pc, qc) + kEps2), cmin, cmax), 7);
                    if (lb0 <= lb1 && lc0 <= lc1 && (praw <= uhi) && (qraw >= ulo)) {
                        const unsigned long long rows = (2ull << (8 * lb1 + 7)) - (1ull << (8 * lb0));
                        const unsigned pat = ((2u << lc1) - (1u << lc0)) & 0xffu;
                        const unsigned rep = pat * 0x01010101u;
                        keep = (proj & rows & (((unsigned long long)rep << 32) | rep)) != 0ull;
                    }
                }
                const unsigned long long mk = __ballot(keep);
                if (keep) s_task[nkeep + __popcll(mk & ((1ull << lane) - 1ull))] = task;
                nkeep += __popcll(mk);
            }
            ntask = nkeep;
        }
        sweep_wave_sync();

        ASURF_SWEEP_LEAVE(3)
        // ---- phase 2: lane = task.  The brick's 8 words; per slab the rectangle mask of the cells between the ray's
        // positions at the slab's two planes; the hits of all tasks in task order (= ray order) ----
        if (lane < rpw) M->cnt[lane] = 0;
        sweep_wave_sync();
        int nkey = 0;   // (wave-uniform)
        bool key_overflow = false;
        for (int t0 = 0; t0 < ntask; t0 += 64) {
            const int ti = t0 + lane;
            unsigned long long h[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) h[k] = 0ull;
            int rl = 0, A = 0, Bb = 0, Bc = 0;
            int fl = 0;
            if (ti < ntask) {
                const unsigned task = s_task[ti];
                rl = (int)(task >> 21);
                A = (int)((task >> 14) & 127);
                Bb = (int)((task >> 7) & 127);
                Bc = (int)(task & 127);
                const SweepRay& R = s_ray[rl];
                fl = R.flags;
                const int a = fl & 3;
                const unsigned long long* plane = a == 0 ? g.fine : (a == 1 ? g.fine_y : g.fine_z);
                const int brick = __mul24(A, R.bs_a) + __mul24(Bb, R.bs_b) + __mul24(Bc, R.bs_c);
                const ulonglong2* wp = reinterpret_cast<const ulonglong2*>(plane + (size_t)brick * 8);
                const ulonglong2 w01 = wp[0], w23 = wp[1], w45 = wp[2], w67 = wp[3];
                const unsigned long long words[8] = {w01.x, w01.y, w23.x, w23.y, w45.x, w45.y, w67.x, w67.y};
                const float ulo = R.ulo, uhi = R.uhi, oa = R.oa, sb = R.sb, sc = R.sc;
                const int i0 = 8 * A, last = R.size_a - 2;
                // brick-local coordinates; the lattice's clamp [0, size - 2] in them
                const float bo = R.ob - (float)(8 * Bb), co = R.oc - (float)(8 * Bc);
                const int bmin = -8 * Bb, bmax = R.size_b - 2 - 8 * Bb, cmin = -8 * Bc, cmax = R.size_c - 2 - 8 * Bc;
                const float kEps2 = 2.f * kEps;  // kEps on the plane's position along a (slopes <= 1) + kEps on b / c
                // plane 0 (the lattice's first slab reaches back to the ray's start, its last one on to the ray's end)
                float praw = (i0 == 0) ? -1e30f : (float)i0;
                float pu = fminf(fmaxf(praw, ulo), uhi);
                float pb = fmaf(pu - oa, sb, bo), pc = fmaf(pu - oa, sc, co);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int i = i0 + k;
                    const float qraw = (i >= last) ? 1e30f : (float)(i + 1);
                    const float qu = fminf(fmaxf(qraw, ulo), uhi);
                    const float qb = fmaf(qu - oa, sb, bo), qc = fmaf(qu - oa, sc, co);
                    const bool live = (i <= last) && (words[k] != 0ull) && (praw <= uhi) && (qraw >= ulo);
                    const int lb0 = max(sweep_clampi((int)floorf(fminf(pb, qb) - kEps2), bmin, bmax), 0);
                    const int lb1 = min(sweep_clampi((int)floorf(fmaxf(pb, qb) + kEps2), bmin, bmax), 7);
                    const int lc0 = max(sweep_clampi((int)floorf(fminf(pc, qc) - kEps2), cmin, cmax), 0);
                    const int lc1 = min(sweep_clampi((int)floorf(fmaxf(pc, qc) + kEps2), cmin, cmax), 7);
                    praw = qraw;
                    pb = qb;
                    pc = qc;
                    if (live && lb0 <= lb1 && lc0 <= lc1) {
                        const unsigned long long rows = (2ull << (8 * lb1 + 7)) - (1ull << (8 * lb0));
                        const unsigned pat = ((2u << lc1) - (1u << lc0)) & 0xffu;
                        const unsigned rep = pat * 0x01010101u;
                        h[k] = words[k] & rows & (((unsigned long long)rep << 32) | rep);
                    }
                }
            }
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) cnt += __popcll(h[k]);
            const int incl = wave_incl_scan(cnt);
            const int total = bcast_i(incl, 63);
            if (total == 0) continue;   // (wave-uniform)
            if (nkey + total > key_cap) {
                key_overflow = true;
                break;
            }
            if (cnt > 0) {
                atomicAdd(&M->cnt[rl], cnt);   // (no return value: ds_add_u32)
                int at = nkey + incl - cnt;
                const int rka = (fl & 4) ? 1023 : 0, rkb = (fl & 8) ? 1023 : 0, rkc = (fl & 16) ? 1023 : 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    unsigned long long hk = h[k];
                    const int i = 8 * A + k;
                    while (hk) {
                        const int bit = __ffsll((long long)hk) - 1;
                        hk &= hk - 1ull;
                        const int ib = 8 * Bb + (bit >> 3), ic = 8 * Bc + (bit & 7);
                        s_krl[at] = (unsigned char)rl;
                        s_key[at++] = ((unsigned)(rka ^ i) << 20) | ((unsigned)(rkb ^ ib) << 10) | (unsigned)(rkc ^ ic);
                    }
                }
            }
            nkey += total;
        }
        if (key_overflow) {   // (wave-uniform)
            if (group > 1) {
                group = max(1, group >> 2);
                continue;
            }
            if (lane == 0) M->hand[r_lo] = 1;
            nkey = 0;
            if (lane < rpw) M->cnt[lane] = 0;
        }
        sweep_wave_sync();

        ASURF_SWEEP_LEAVE(4)
        // ---- the rays' slices of the key buffer ----
        {
            const int c = (lane < rpw) ? M->cnt[lane] : 0;
            const int incl = wave_incl_scan(c);
            if (lane < rpw) {
                M->start[lane] = incl - c;
                M->n[lane] = 0;
            }
            if (lane == rpw - 1) M->start[rpw] = incl;
        }
        sweep_wave_sync();

        // ---- phase 3: exact membership, lane = hit ----
        for (int t0 = 0; t0 < nkey; t0 += 64) {
            const int t = t0 + lane;
            if (t < nkey) {
                const int rl = s_krl[t];
                unsigned key = s_key[t];
                const SweepRay& R = s_ray[rl];
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
                const int n0[3] = {R.n0[0], R.n0[1], R.n0[2]};
                int32_t lk[8];
                if constexpr (LAZY) {   // asked for before the exact test's arithmetic: one round trip hidden behind it
                    const bool in = vv[0] + 1 < g.size[0] && vv[1] + 1 < g.size[1] && vv[2] + 1 < g.size[2];
                    const int32_t* lp = g.links + ((size_t)vv[0] * g.offx + (size_t)vv[1] * g.offy + vv[2]);
                    lk[0] = in ? lp[0] : -1;
                    lk[1] = in ? lp[1] : -1;
                    lk[2] = in ? lp[g.offy] : -1;
                    lk[3] = in ? lp[g.offy + 1] : -1;
                    lk[4] = in ? lp[g.offx] : -1;
                    lk[5] = in ? lp[g.offx + 1] : -1;
                    lk[6] = in ? lp[g.offx + g.offy] : -1;
                    lk[7] = in ? lp[g.offx + g.offy + 1] : -1;
                }
                float tc[3], tf[3], t_entry;
                if (R.fast_div) {  // the IEEE quotients, three instructions each (quotient(), asurf_kernels.h)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int up = r.d[k] > 0.f ? 1 : 0;
                        const float rdk = R.rd[k];
                        tc[k] = quotient((float)(vv[k] + 1 - up) - r.o[k], r.d[k], rdk);
                        tf[k] = quotient((float)(vv[k] + up) - r.o[k], r.d[k], rdk);
                    }
                } else {
                    voxel_times(r, vv[0], vv[1], vv[2], tc, tf);
                }
                bool keep = walk_visits(r, n0, vv, tc, tf, &t_entry);
                if constexpr (LAZY) keep = keep && lazy_gates_pass(g, o.sigma_thresh, lk);
                if (keep) {
                    atomicAdd(&M->n[rl], 1);
                    if (want_cut && (t_entry + o.step_size > r.tmax) &&
                        walk_cut_before(g, r, o.step_size, n0, vv, tc))
                        key |= kSweepCutBit;
                } else {
                    key = kSweepDropped;
                }
                s_key[t] = key;
            }
        }
        sweep_wave_sync();

        ASURF_SWEEP_LEAVE(5)
        // ---- phase 4: per ray, counts and allocations ----
        if (lane >= r_lo && lane < r_hi) {
            const long long q = q0 + lane;
            int n = 0, spb = 0;
            if (q < Q) {
                bool handback = M->hand[lane] != 0;
                n = handback ? 0 : M->n[lane];
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
            M->n[lane] = n;
            M->spb[lane] = spb;
        }
        sweep_wave_sync();
        // the group's cells of the dense list are laid out SLOT-major -- all rays' slot 0, then slot 1, ... -- so that
        // neighbouring lanes of k_geom hold neighbouring rays at the same depth (slot0[s]: first cell of slot s)
        int lbase;
        {
            unsigned fm = 0u;  // rays that fill slot `lane`
            // (ray r2's count sits in lane r2: read across the wave, no LDS)
            const int n_mine = (lane >= r_lo && lane < r_hi) ? M->n[lane] : 0;
#pragma unroll
            for (int r2 = 0; r2 < kSweepWaveRays; ++r2)
                fm |= (__builtin_amdgcn_readlane(n_mine, r2) > lane ? 1u : 0u) << r2;
            if (lane >= P.C1) fm = 0u;
            const int c = __popc(fm);
            const int ci = wave_incl_scan(c);
            s_slot0[lane] = ci - c;
            s_fill[lane] = fm;
            const int cells = bcast_i(ci, 63);
            int lb = 0;
            const int shard = list_segment_unit(P, unit, rpw);
            if (ASURF_AID(o, 1)) {   // (instrumented build: what the allocation's round trip costs -- results invalid)
                lb = (int)((unit * 640u) % (unsigned)max(P.list_cap - 1024, 1));
            } else
            if (lane == 0 && cells > 0) lb = atomicAdd(&P.counters[kCtrListShard0 + shard * kCtrShardStride], cells);
            lbase = shard * P.list_cap + bcast_i(lb, 0);
        }
        // (P.cut was stored by other lanes of this wave than the ones that lower it below: the stores must have left)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        sweep_wave_sync();

        // ---- phase 5: per surviving entry, its rank among the ray's keys is its slot ----
        for (int t0 = 0; t0 < nkey; t0 += 64) {
            const int t = t0 + lane;
            if (t >= nkey) continue;
            const int rl = s_krl[t];
            const unsigned key = s_key[t];
            if (key == kSweepDropped || M->n[rl] == 0) continue;  // n == 0: ray handed back
            const unsigned kk = key & ~kSweepCutBit;
            const int k_lo = M->start[rl], k_hi = M->start[rl + 1];
            int ord = 0;  // (a dropped key has bit 31 set: larger than every kept one)
            for (int j = k_lo; j < k_hi; ++j) ord += ((s_key[j] & ~kSweepCutBit) < kk) ? 1 : 0;
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
                const int before = __popc(s_fill[ord] & ((1u << rl) - 1u));  // rays in front that fill slot `ord` too
                P.items[(size_t)ord * Q + q] = voxel;
                P.list[lbase + s_slot0[ord] + before] = ((unsigned)ord << kListRayBits) | (unsigned)q;
            } else {
                const int at = M->spb[rl] + ord - P.C1;
                P.sp_items[at] = voxel;
                P.sp_ray[at] = (int)q;
            }
        }
        sweep_wave_sync();
        r_lo = r_hi;
    }
#undef ASURF_COARSE_WORD
    tl.end();
}

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ASURF_SWEEP_EU, ASURF_SWEEP_EU)))
k_sweep_wave(KGrid g, KOpt o, const float* __restrict__ origins, const float* __restrict__ dirs, long long Q,
             PipeBuf P, int n_layers, int want_cut, int task_cap, int key_cap, int wave_bytes, int rpw) {
    sweep_wave_body<false>(g, o, origins, dirs, Q, P, n_layers, want_cut, task_cap, key_cap, wave_bytes, rpw);
}

// Small batches over links-only planes (asurf_masks2.h): the exact test also applies the value gates.  No occupancy
// attribute: the launch is a few hundred waves, and the 24 gathered values may have their registers.
__global__ void __launch_bounds__(256)
k_sweep_wave_lazy(KGrid g, KOpt o, const float* __restrict__ origins, const float* __restrict__ dirs, long long Q,
                  PipeBuf P, int n_layers, int want_cut, int task_cap, int key_cap, int wave_bytes, int rpw) {
    sweep_wave_body<true>(g, o, origins, dirs, Q, P, n_layers, want_cut, task_cap, key_cap, wave_bytes, rpw);
}

}  // namespace asurf
