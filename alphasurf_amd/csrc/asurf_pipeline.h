// asurf_pipeline.h -- the forward pass as three dense kernels (gfx950, wave64).
//
//   k_sweep (asurf_sweep.h): the work voxels of every ray, found slab-parallel in closed form, into the tables
//            described under k_walk; the default.  k_walk / k_walk_seg below find them by stepping: they serve
//            the calls that test the `live` plane, ASURF_WALK=step, and the tests' per-voxel reference.
//   k_walk : one ray per lane, nothing but the hierarchical DDA (walk_hier: empty 8^3 bricks are
//            left in one step) + occupancy-bit tests (few registers, many waves per SIMD).  The
//            first C1 work voxels of a ray go to a lane-coalesced table [C1][Q]; a ray with more
//            re-walks once and appends the rest to a contiguous slice of a global spill list, so
//            rays of any length stay on this path.  The occupied (slot, ray) cells of the table are
//            also appended to a dense list (slot-major within a wave).
//            Batches that do not fill the chip several times over use k_walk_seg: S lanes per ray, each
//            walking one t-segment (the walk is a serial chain; see the kernel).
//   k_geom : one WORK VOXEL per lane, a fixed grid of blocks striding over the dense
//            list, then over the spill list: every lane is busy no matter how
//            the work voxels are spread over rays, and no block is dispatched only to exit.
//            Gathers the 8 corners, forms and solves the cubic (f64), validates the roots and
//            leaves up to 3 sample candidates per voxel in a sharded pool.
//   k_comp : one ray per lane again, for what is inherently sequential along a ray: the
//            sigma gate, truncation window, compositing, the sample cache of the fused losses
//            and the records the replaying backward kernel consumes.  It visits only the slots
//            k_geom marked (cand_mask).  Shading stays wave-cooperative (pair_shade_fwd).
//
// A ray is handed back to the one-kernel path (k_forward with a compacted ray list) only
// when a voxel yields more than 3 candidates or a pool runs out; results are identical.
#pragma once
#include "asurf_kernels.h"

namespace asurf {

struct CandRec {  // 32 bytes
    float px, py, pz;  // position inside the voxel
    float ts;          // ray parameter of the sample (cache of the fused losses)
    float raw;         // trilinear raw opacity
    float ext;         // fake samples: signed distance to the nearest level set (std-normalised)
    int meta;          // bit0 fake, bits1-2 root index, bits3-7 level-set index
    int pad;
};

// value-select store: keeps the candidates in registers (selecting a destination ADDRESS
// would put them in scratch)
__device__ __forceinline__ void cand_put(CandRec& dst, const CandRec& src, bool sel) {
    dst.px = sel ? src.px : dst.px;
    dst.py = sel ? src.py : dst.py;
    dst.pz = sel ? src.pz : dst.pz;
    dst.ts = sel ? src.ts : dst.ts;
    dst.raw = sel ? src.raw : dst.raw;
    dst.ext = sel ? src.ext : dst.ext;
    dst.meta = sel ? src.meta : dst.meta;
}

// The candidate pool is cut into shards with their own cursors so that the allocating
// atomics of k_geom do not serialise on one address.
// work voxels per ray the table keeps (lane-coalesced planes); the rest of a longer ray goes to the spill list
constexpr int kPipeC1 = 32;
static_assert(kPipeC1 <= 32, "PipeBuf::cand_mask: one bit per table slot");
constexpr int kPoolShards = 256;
constexpr int kCtrList = 2 + kPoolShards;   // [kCtrList + 1]: entries in the backward's list
// The dense list of table cells is cut into shards too (producer block b appends to shard b % kListShards, one
// atomic per block): 4 * 10^4 blocks adding to ONE address took 0.3 ms of k_sweep's 0.5 (each add is a round
// trip to the memory side, ~8 ns apiece when they queue on one line).  Counters sit 64 bytes apart.
constexpr int kListShards = 16;
constexpr int kCtrShardStride = 16;
constexpr int kCtrListShard0 = kCtrList + 2;
// ... and into kListGroups groups of shards by WHERE the rays are: rays come in image order, a band of band_rays
// consecutive rays (8192: ten rows of an 800-pixel image) sees one slab of the scene, and band n belongs to group n % 8.
// k_geom's blocks on XCD x (block b runs on XCD b % 8 -- variants/timeline.py reads it back from XCC_ID) take group
// x: the links and values of a slab are then fetched into ONE of the eight L2s instead of all of them.  Interleaved
// bands keep the groups level (one contiguous eighth of the image per XCD leaves the XCDs of the background idle);
// a small batch is cut into shorter bands (PipeBuf::band_rays, at least 64 bands where the batch allows).
#ifndef ASURF_LIST_GROUPS
#define ASURF_LIST_GROUPS 8
#endif
constexpr int kListGroups = ASURF_LIST_GROUPS;
constexpr int kListSegs = kListGroups * kListShards;
constexpr int kBandRaysMax = 8192, kBandRaysMin = 256;  // powers of two; every producer block lies within a band
// the backward's list of visited sample records (asurf_bwd_pipeline.h) is cut likewise: k_bwd_seq block b appends to
// segment b % kRecShards (2500 blocks adding to one cursor were 12 of that kernel's 42 us)
#ifndef ASURF_REC_SHARDS
#define ASURF_REC_SHARDS 8
#endif
constexpr int kRecShards = ASURF_REC_SHARDS;
constexpr int kCtrRecShard0 = kCtrListShard0 + kListSegs * kCtrShardStride;
// [kCtrRecOver]: some ray's sample records did not fit (k_comp / the one-kernel forward): with [0] -- rays handed to the
// one-kernel path -- what tells the walking backward of the fused call at a glance that it has nothing to do
constexpr int kCtrRecOver = kCtrRecShard0 + kRecShards * kCtrShardStride;
constexpr int kCtrTotal = kCtrRecOver + kCtrShardStride;
constexpr int kListRayBits = 27;  // rays per call < 2^27 for the list-driven kernels
constexpr int kRecRayBits = 25;   // the backward's record list: record << 25 | ray (records a ray <= 128, rays < 2^25)
// The kernels that work through a dense list are launched with a fixed number of blocks and
// stride over it: the list length is only known on the device, and tens of thousands of
// blocks that exit at once still cost their dispatch.
constexpr int kListBlocks = 2048;

struct PipeBuf {
    int* items;      // [C1][Q] packed work voxels
    int* citem;      // [C1][Q] (pool index << 2) | candidate count
    int* n_items;    // [Q] work voxels of the ray; < 0: ray handed back to the one-kernel path
    int* cut;        // [Q] ordinal of the first work voxel the reference's BACKWARD would skip
    int* sp_base;    // [Q] first spill entry of the ray (valid when n_items > C1)
    int* sp_items;   // [sp_cap] spilled work voxels, contiguous per ray
    int* sp_citem;   // [sp_cap]
    int* sp_ray;     // [sp_cap] owning ray
    int* flagged;    // [Q] compacted ids of the rays handed back
    int* counters;   // [0] n_flagged, [1] spill cursor, [2 .. 2+kPoolShards) pool cursors,
                     // kListSegs cursors of `list`, kRecShards cursors of the backward's list (asurf_bwd_pipeline.h)
    unsigned* list;  // dense list of the occupied (slot, ray) cells of `items`: slot << 27 | ray; kListShards
                     // segments of list_cap cells
    int list_cap;
    int band_rays;   // rays per band (list_segment)
    CandRec* pool;
    // per ray: which of its work voxels yielded a candidate (bit s: the s-th one, table slots first, then the spill
    // slice; bit 63: one beyond the 63rd did) -- k_geom ORs the bits in, k_comp visits those only
    unsigned long long* cand_mask;
    int shard_cap;   // records per pool shard
    int sp_cap;
    int C1;
};

// Segment the producer block (rays_per_block consecutive rays, a power of two <= kBandRaysMin) appends to: group =
// its band's, shard = the block's place in the band + the band's place in the group, so that the blocks of a group
// go round its shards whatever the number of blocks per band.
__device__ __forceinline__ int list_segment(const PipeBuf& P, int rays_per_block) {
    const unsigned nbb = (unsigned)(P.band_rays / rays_per_block);
    const unsigned band = blockIdx.x / nbb, in_band = blockIdx.x - band * nbb;
    return (int)(band % kListGroups) * kListShards + (int)((in_band + band / kListGroups) % kListShards);
}

__device__ __forceinline__ int wave_excl_scan(int v, int* total) {
    const int lane = threadIdx.x & 63;
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    *total = __shfl(incl, 63);
    return incl - v;
}

// ------------------------------------------------------------------------------------
template <bool CLDS>
__global__ void __launch_bounds__(kBlock)
k_walk(KGrid g, KOpt o, const float* __restrict__ origins, const float* __restrict__ dirs,
       long long Q, PipeBuf P, int want_cut) {
    extern __shared__ unsigned int s_coarse[];
    const int tid = threadIdx.x;
    if constexpr (CLDS) {
        for (int i = tid; i < g.coarse_words; i += kBlock) s_coarse[i] = g.coarse[i];
        __syncthreads();
    }
    const long long q = (long long)blockIdx.x * kBlock + tid;
    const bool valid = q < Q;
    Ray r;
    r.tmin = 1.f;
    r.tmax = 0.f;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r.o[k] = origins[3 * q + k];
            r.d[k] = dirs[3 * q + k];
        }
        setup_ray(g, o, r);
    }
    const bool hits_grid = valid && !(r.tmin > r.tmax);
    int cnt = 0;
    int cutv = 0x7fffffff;
    if (hits_grid) {
        Walk w;
        walk_init(g, r, w);
        bool active = true;
        const bool jump = !(o.debug & 8);
        // every iteration either advances one axis or is a refused jump followed by such a step; the
        // guard only exists so that no defect in the walk can keep a wave alive
        int guard = 4 * (g.size[0] + g.size[1] + g.size[2]) + 64;
        while (active && --guard > 0) {
            const int kind = walk_hier<CLDS>(g, s_coarse, w, o.step_size, r.tmax, jump);
            active = (w.t <= r.tmax);
            if (kind == 2) {
                if (cnt < P.C1) P.items[(size_t)cnt * Q + q] = pack_voxel(w.vx, w.vy, w.vz);
                ++cnt;
            } else if (kind == 1 && want_cut && active && (w.t + o.step_size > r.tmax) && cutv == 0x7fffffff) {
                if (!links_present(g, w)) cutv = cnt;  // surf_trav.cu:1935-1936
            }
        }
    }
    // rays longer than the table: reserve a contiguous spill slice (one atomic per wave) and
    // walk again to fill it
    const int extra = (cnt > P.C1) ? cnt - P.C1 : 0;
    int total;
    const int pre = wave_excl_scan(extra, &total);
    if (total > 0) {
        int base0 = 0;
        if ((tid & 63) == 0) base0 = atomicAdd(&P.counters[1], total);
        base0 = __shfl(base0, 0);
        if (extra > 0) {
            const int base = base0 + pre;
            if (base + extra <= P.sp_cap) {
                P.sp_base[q] = base;
                Walk w;
                walk_init(g, r, w);
                bool active = true;
                int k = 0;
                const bool jump = !(o.debug & 8);
                int guard = 4 * (g.size[0] + g.size[1] + g.size[2]) + 64;
                while (active && --guard > 0) {
                    const bool hit = walk_hier<CLDS>(g, s_coarse, w, o.step_size, r.tmax, jump) == 2;
                    active = (w.t <= r.tmax);
                    if (hit) {
                        if (k >= P.C1) {
                            P.sp_items[base + k - P.C1] = pack_voxel(w.vx, w.vy, w.vz);
                            P.sp_ray[base + k - P.C1] = (int)q;
                        }
                        ++k;
                    }
                }
            } else {  // spill list exhausted: the one-kernel path renders this ray
                cnt = -1;
                P.flagged[atomicAdd(&P.counters[0], 1)] = (int)q;
                for (int e = base; e < min(base + extra, P.sp_cap); ++e) P.sp_ray[e] = -1;
            }
        }
    }
    if (valid) {
        P.n_items[q] = cnt;
        P.cut[q] = cutv;
        P.cand_mask[q] = 0ull;
    }
    // dense list of this wave's table cells, slot-major so that neighbouring list entries are
    // neighbouring rays at the same depth (k_geom works one entry per lane)
    {
        const int nt = (valid && cnt > 0) ? min(cnt, P.C1) : 0;
        int wave_total;
        wave_excl_scan(nt, &wave_total);
        if (wave_total > 0) {
            int base0 = 0;
            const int shard = list_segment(P, kBlock);
            if ((tid & 63) == 0)
                base0 = shard * P.list_cap + atomicAdd(&P.counters[kCtrListShard0 + shard * kCtrShardStride], wave_total);
            base0 = __shfl(base0, 0);
            const unsigned long long below = (1ull << (tid & 63)) - 1ull;
            for (int j = 0; __any(nt > j); ++j) {
                const unsigned long long m = __ballot(nt > j);
                if (nt > j) P.list[base0 + __popcll(m & below)] = ((unsigned)j << kListRayBits) | (unsigned)q;
                base0 += __popcll(m);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// k_walk with the ray cut into S segments walked by S neighbouring lanes.  A ray's walk is a serial
// chain of a few hundred dependent iterations and k_walk's duration is that of its longest chains, not
// of its instruction count; S lanes shorten them S-fold.  Exactness: walk_hier's decisions depend on
// (next voxel, ray) only, so a lane may start anywhere the sequential walk provably is: at a parameter A
// whose position o + A d lies inside the vertex lattice and at least 1/256 voxel away from every voxel
// plane (the margin the brick jumps use for the same question), the walk is in voxel floor(o + A d).
// Lane s starts there, does not report that voxel (its entry precedes A: the previous lane owns it),
// and stops before the first voxel entered at or after A(s+1).  Flagged voxels are kept in LDS and
// placed once the counts of the lanes in front are known; a lane with more than kSegCap of them walks
// its segment once more for the rest.  A ray that offers no landing point is walked by its lane 0.
// ------------------------------------------------------------------------------------
constexpr int kSegCap = 8;

template <bool CLDS, int S>
__global__ void __launch_bounds__(kBlock)
k_walk_seg(KGrid g, KOpt o, const float* __restrict__ origins, const float* __restrict__ dirs,
           long long Q, PipeBuf P, int want_cut) {
    static_assert(S == 2 || S == 4 || S == 8, "segments per ray");
    extern __shared__ unsigned int s_coarse[];
    __shared__ int s_tmp[kBlock * kSegCap];
    const int tid = threadIdx.x;
    if constexpr (CLDS) {
        for (int i = tid; i < g.coarse_words; i += kBlock) s_coarse[i] = g.coarse[i];
        __syncthreads();
    }
    constexpr int R = 64 / S;  // rays per wave
    const int lane = tid & 63;
    const int seg = lane & (S - 1), ri = lane / S;
    const long long q = ((long long)blockIdx.x * (kBlock / 64) + (tid >> 6)) * R + ri;
    const bool valid = q < Q;
    Ray r;
    r.tmin = 1.f;
    r.tmax = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.o[k] = 0.f;
        r.d[k] = 1.f;
    }
    if (valid) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r.o[k] = origins[3 * q + k];
            r.d[k] = dirs[3 * q + k];
        }
        setup_ray(g, o, r);
    }
    const bool hits_grid = valid && !(r.tmin > r.tmax);
    Walk w0;
    walk_init(g, r, w0);
    // ---- where my segment starts ----
    float a_start = r.tmin;
    int v0x = -1, v0y = -1, v0z = -1;
    bool fail = false;
    if (hits_grid && seg > 0) {
        const float len = r.tmax - r.tmin;
        const float kEps = 1.f / 256.f;
        float t = r.tmin + len * ((float)seg * (1.f / S));
        bool ok = false;
#pragma unroll
        for (int attempt = 0; attempt < 3; ++attempt) {
            if (!ok) {
                if (attempt > 0) t += 0.37f;
                const float px = fmaf(t, w0.dx, w0.ox), py = fmaf(t, w0.dy, w0.oy), pz = fmaf(t, w0.dz, w0.oz);
                const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
                const bool in_x = px > kEps && px < (float)w0.lim0 - kEps && (px - fx) > kEps && (px - fx) < 1.f - kEps;
                const bool in_y = py > kEps && py < (float)w0.lim1 - kEps && (py - fy) > kEps && (py - fy) < 1.f - kEps;
                const bool in_z = pz > kEps && pz < (float)w0.lim2 - kEps && (pz - fz) > kEps && (pz - fz) < 1.f - kEps;
                ok = in_x && in_y && in_z && (t + 1.f < r.tmax);
                v0x = (int)fx;
                v0y = (int)fy;
                v0z = (int)fz;
            }
        }
        fail = !ok || !(len > 16.f) || !w0.straight;
        a_start = t;
    }
    // a ray none of whose inner boundaries failed is split; otherwise its lane 0 walks all of it.
    // (boundaries must also be increasing: the retries move a start by at most 0.74 < len / S)
    const unsigned long long failed = __ballot(fail);
    const unsigned long long mine = ((1ull << S) - 1ull) << (ri * S);
    const bool unsplit = (failed & mine) != 0ull;
    float a_end = __shfl_down(a_start, 1);
    if (seg == S - 1 || unsplit) a_end = 3.0e38f;
    const bool walks = hits_grid && (seg == 0 || !unsplit);
    const bool skip_first = walks && seg > 0;
    const bool jump = !(o.debug & 8);

    int c = 0;                    // flagged voxels of my segment
    int cut_local = 0x7fffffff;   // my count at the backward-quirk event (surf_trav.cu:1935-1936)
    int offset = 0, total = 0, base = 0;
    bool spill_ok = true;
    // pass 0: walk, keep the first kSegCap flagged voxels in LDS; pass 1 (lanes with more): walk again and
    // write the rest to their final places
    auto place = [&](int slot, int voxel) {
        if (slot < P.C1) {
            P.items[(size_t)slot * Q + q] = voxel;
        } else if (spill_ok) {
            P.sp_items[base + slot - P.C1] = voxel;
            P.sp_ray[base + slot - P.C1] = (int)q;
        }
    };
    auto walk_segment = [&](int pass, bool run) {
        if (!run) return;
        Walk w = w0;
        if (skip_first) {
            w.nx = v0x;
            w.ny = v0y;
            w.nz = v0z;
            w.t = a_start;
        }
        bool active = true;
        int k = 0;
        int guard = 4 * (g.size[0] + g.size[1] + g.size[2]) + 64;
        while (active && --guard > 0) {
            const int kind = walk_hier<CLDS>(g, s_coarse, w, o.step_size, r.tmax, jump);
            const bool in_ray = (w.t <= r.tmax);
            active = in_ray && (w.t < a_end);
            const bool own = !(skip_first && w.vx == v0x && w.vy == v0y && w.vz == v0z);
            if (kind == 2 && own) {
                const int voxel = pack_voxel(w.vx, w.vy, w.vz);
                if (pass == 0) {
                    if (k < kSegCap) s_tmp[tid * kSegCap + k] = voxel;
                } else if (k >= kSegCap) {
                    place(offset + k, voxel);
                }
                ++k;
            } else if (pass == 0 && kind == 1 && own && want_cut && in_ray && (w.t + o.step_size > r.tmax) &&
                       cut_local == 0x7fffffff) {
                if (!links_present(g, w)) cut_local = k;
            }
        }
        if (pass == 0) c = k;
    };
    walk_segment(0, walks);
    // ---- counts of the lanes in front, spill slice of the ray ----
    {
        int incl = c;
#pragma unroll
        for (int d = 1; d < S; d <<= 1) {
            const int tv = __shfl_up(incl, d);
            if (seg >= d) incl += tv;
        }
        offset = incl - c;
        total = __shfl(incl, ri * S + S - 1);
    }
    int cutv = (cut_local == 0x7fffffff) ? 0x7fffffff : offset + cut_local;
#pragma unroll
    for (int d = 1; d < S; d <<= 1) cutv = min(cutv, __shfl_xor(cutv, d));
    int cnt = total;
    const int extra = (seg == 0 && cnt > P.C1) ? cnt - P.C1 : 0;
    int wave_extra;
    const int pre = wave_excl_scan(extra, &wave_extra);
    if (wave_extra > 0) {
        int base0 = 0;
        if (lane == 0) base0 = atomicAdd(&P.counters[1], wave_extra);
        base0 = __shfl(base0, 0);
        base = __shfl(base0 + pre, ri * S);
        const int ray_extra = __shfl(extra, ri * S);
        spill_ok = base + ray_extra <= P.sp_cap;
        if (seg == 0 && extra > 0) {
            if (spill_ok) {
                P.sp_base[q] = base;
            } else {  // spill list exhausted: the one-kernel path renders this ray
                P.flagged[atomicAdd(&P.counters[0], 1)] = (int)q;
                for (int e2 = base; e2 < min(base + extra, P.sp_cap); ++e2) P.sp_ray[e2] = -1;
            }
        }
        if (!spill_ok && cnt > P.C1) cnt = -1;
    }
    // ---- place what LDS holds, then the overflow ----
    {
        const int nk = min(c, kSegCap);
        for (int j = 0; j < nk; ++j) place(offset + j, s_tmp[tid * kSegCap + j]);
    }
    if (__any(c > kSegCap)) walk_segment(1, walks && c > kSegCap);
    if (valid && seg == 0) {
        P.n_items[q] = cnt;
        P.cut[q] = cutv;
        P.cand_mask[q] = 0ull;
    }
    // dense list of this wave's table cells (as k_walk)
    {
        const int nt = (valid && seg == 0 && cnt > 0) ? min(cnt, P.C1) : 0;
        int wave_total;
        wave_excl_scan(nt, &wave_total);
        if (wave_total > 0) {
            int base0 = 0;
            const int shard = list_segment(P, (kBlock / 64) * R);
            if (lane == 0)
                base0 = shard * P.list_cap + atomicAdd(&P.counters[kCtrListShard0 + shard * kCtrShardStride], wave_total);
            base0 = __shfl(base0, 0);
            const unsigned long long below = (1ull << lane) - 1ull;
            for (int j = 0; __any(nt > j); ++j) {
                const unsigned long long m = __ballot(nt > j);
                if (nt > j) P.list[base0 + __popcll(m & below)] = ((unsigned)j << kListRayBits) | (unsigned)q;
                base0 += __popcll(m);
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// One work voxel per lane, taken from a dense list: the table cells written by k_sweep / k_walk, then the
// spill list.  kListBlocks blocks stride over it.
#ifndef ASURF_GEOM_WAVES
#define ASURF_GEOM_WAVES 3
#endif
// RELAXED: no occupancy bound -- 204 registers, no scratch, two waves a SIMD.  A batch of a few thousand rays puts a wave
// or less on a SIMD: the waves' own chains are the kernel's time, and the 76 bytes of scratch of the three-wave form sit
// in them (5 184 rays: 0.0172 -> 0.0147 ms, 40 000: 0.0448 -> 0.0372; 640 000: 0.164 -> 0.171, so the full batch keeps
// the three-wave form).
template <bool RELAXED>
__global__ void __launch_bounds__(kBlock, RELAXED ? 1 : ASURF_GEOM_WAVES)
k_geom(KGrid g, KOpt o, const float* __restrict__ origins, const float* __restrict__ dirs,
       long long Q, PipeBuf P) {
    const TlScope tl(3);
    const int lane = threadIdx.x & 63;
    // The table cells come as kListGroups groups of kListShards segments.  This block takes group blockIdx % 8 (its
    // XCD's, see kListGroups) together with the other blocks of that XCD; the segments' running sums (wave-uniform)
    // turn a position in the group's concatenation into (segment, offset).  The spill list follows: its 64-entry
    // pieces go round the groups (a launch of its own cost 13 us, empty or not).
    const int grp = (int)(blockIdx.x % kListGroups);
    int seg_end[kListShards];
    int n_cells = 0;
#pragma unroll
    for (int sh = 0; sh < kListShards; ++sh) {
        n_cells += min(P.counters[kCtrListShard0 + (grp * kListShards + sh) * kCtrShardStride], P.list_cap);
        seg_end[sh] = n_cells;
    }
    const int n_spill = min(P.counters[1], P.sp_cap);
    const int sp_pieces = (n_spill + 63) >> 6;
    const int n_list = n_cells + ((sp_pieces - grp + kListGroups - 1) / kListGroups) * 64;  // (grp < kListGroups)
    const unsigned wave_id = (blockIdx.x / kListGroups) * (kBlock / 64) + (threadIdx.x >> 6);  // within the group
    const int stride = (int)(gridDim.x / kListGroups) * kBlock;
    // Three work voxels in four hold no crossing (their level set is within the corner values' range, but not along
    // this ray), and the Vieta solver (f64 acos, sincos, cbrt; both of its branches in every wave) is most of what a
    // voxel costs.  A test in front of it is worth nothing while lanes that pass and lanes that fail share a wave
    // (measured: 0.21 -> 0.31 ms), so the wave works in two stages: a LIGHT one over the list, 64 cells a turn --
    // links, surface values and a conservative "no root of the cubic where a sample could be" in float
    // (voxel_may_hold_root: Bernstein coefficients of one sign over the interval of ray parameters whose positions
    // root_position() accepts) -- which finishes the cells that hold nothing and queues the rest in LDS, and a HEAVY
    // one that runs the reference's arithmetic on full waves of queued cells.  (As two kernels with the queue in
    // memory, the light one at full occupancy: 0.091 + 0.086 ms against 0.178 in one, and 10 us more at the training
    // script's batch size.)  Fake samples need every voxel's geometry: with them enabled (or
    // ASURF_DEBUG & 32) every cell is queued unseen.
    __shared__ int s_queue[kBlock / 64][3][128];
    int (*queue)[128] = s_queue[threadIdx.x >> 6];
    const bool filter = !o.surf_fake_sample && !(o.debug & 32);
    int qn = 0;  // queued cells (wave-uniform)
    unsigned turn = 0;
    int base = (int)wave_id * 64;
    for (;;) {
      // ---- light stage: until a full wave is queued or the list ends ----
      while (qn < 64 && base < n_list) {
        const int i = base + lane;
        base += stride;
        long long q = 0;
        long long entry = 0;
        int packed_voxel = 0;
        bool mine = i < n_list;
        const bool SPILL = i >= n_cells;
        int slot = 0;  // ordinal of the voxel among the ray's work voxels
        if (SPILL) {
            const int t = i - n_cells;
            entry = (long long)(((t >> 6) * kListGroups + grp) << 6) + (t & 63);
            mine = mine && entry < n_spill;
            const int qq = mine ? P.sp_ray[entry] : -1;
            mine = (qq >= 0) && (P.n_items[qq] >= 0);
            q = (qq >= 0) ? qq : 0;
            if (mine) {
                packed_voxel = P.sp_items[entry];
                slot = P.C1 + (int)(entry - P.sp_base[q]);
            }
        } else {
            int sh = 0;
#pragma unroll
            for (int k = 0; k < kListShards - 1; ++k) sh += (i >= seg_end[k]) ? 1 : 0;
            int seg0 = 0;  // seg_end[sh - 1]
#pragma unroll
            for (int k = 0; k < kListShards - 1; ++k) seg0 = (sh == k + 1) ? seg_end[k] : seg0;
            const unsigned e = mine ? P.list[(size_t)(grp * kListShards + sh) * P.list_cap + (i - seg0)] : 0u;
            q = (long long)(e & ((1u << kListRayBits) - 1u));
            slot = (int)(e >> kListRayBits);
            entry = (long long)slot * Q + q;
            // (a ray handed back to the one-kernel path in the meantime is not looked up: its cells cost their work,
            // and what they leave behind is never read)
            if (mine) packed_voxel = P.items[entry];
        }
        bool maybe = mine;
        if (filter && mine) {
            Ray r;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                r.o[k] = origins[3 * q + k];
                r.d[k] = dirs[3 * q + k];
            }
            setup_ray(g, o, r);
            maybe = voxel_may_hold_root(g, r, packed_voxel);
            if (!maybe && slot >= 63) P.sp_citem[entry] = 0;  // nothing here (k_comp reads the cell if it scans)
        }
        const unsigned long long mq = __ballot(maybe);
        if (maybe) {
            const int at = qn + __popcll(mq & ((1ull << lane) - 1ull));
            queue[0][at] = (int)q;
            queue[1][at] = slot;
            queue[2][at] = packed_voxel;
        }
        qn += __popcll(mq);
      }
      if (qn == 0) break;  // (the list has ended)
      // ---- heavy stage: the last min(qn, 64) queued cells, one per lane ----
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const int take = min(qn, 64);
      qn -= take;
      const bool mine = lane < take;
      const long long q = mine ? queue[0][qn + lane] : 0;
      const int slot = mine ? queue[1][qn + lane] : 0;
      const int packed_voxel = mine ? queue[2][qn + lane] : 0;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (read before the next light stage writes)
      const bool SPILL = slot >= P.C1;
      const long long entry = SPILL ? (long long)P.sp_base[q] + (slot - P.C1) : (long long)slot * Q + q;
      {
        CandRec c0, c1, c2;
        c0.px = c0.py = c0.pz = c0.ts = c0.raw = c0.ext = 0.f;
        c0.meta = c0.pad = 0;
        c1 = c0;
        c2 = c0;
        int nc = 0;
        bool overflow = false;
        if (mine) {
            Ray r;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                r.o[k] = origins[3 * q + k];
                r.d[k] = dirs[3 * q + k];
            }
            setup_ray(g, o, r);
            Vox v;
            load_voxel(g, r, packed_voxel, v);
            bool vox_has_sample = false, vox_has_surf = false;
            for (int li = 0; li < g.level_set_num; ++li) {  // surf_trav.cu:273-379 up to the sigma gate
                const float lvf = g.level_set[li];
                if (lvf < v.smin || lvf > v.smax) continue;
                vox_has_surf = true;
                double st[3] = {-1, -1, -1};
                solve_cubic_vieta(v.fs[0] - (double)lvf, v.fs[1], v.fs[2], v.fs[3], st);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (!(st[j] > 0)) continue;
                    float pos[3];
                    if (!root_position(r, v, st[j], pos)) continue;
                    vox_has_sample = true;
                    if (o.only_outward_intersect) {
                        float sg[3];
                        field_grad(v.surf, pos, sg);
                        const float ndd = -(sg[0] * r.d[0] + sg[1] * r.d[1] + sg[2] * r.d[2]);
                        if (ndd >= 0.f) continue;
                    }
                    if (nc < 3) {
                        CandRec cr;
                        cr.px = pos[0];
                        cr.py = pos[1];
                        cr.pz = pos[2];
                        cr.ts = (float)((double)v.t_close + st[j]);
                        cr.raw = trilerp8(v.dens, pos);
                        cr.ext = 0.f;
                        cr.meta = (li << 3) | (j << 1);
                        cr.pad = 0;
                        cand_put(c0, cr, nc == 0);
                        cand_put(c1, cr, nc == 1);
                        cand_put(c2, cr, nc == 2);
                        ++nc;
                    } else {
                        overflow = true;
                    }
                }
            }
            // fake sample (surf_trav.cu:423-503): everything that does not depend on earlier samples
            if (o.surf_fake_sample && !vox_has_sample && (!o.limited_fake_sample || vox_has_surf) &&
                ((v.t_far - v.t_close) > o.surf_fake_sample_min_vox_len)) {
                float pos[3];
                fake_position(r, v, pos);
                const float raw = trilerp8(v.dens, pos);
                if (raw > o.sigma_thresh) {
                    double miu, sd;
                    c0.px = pos[0];
                    c0.py = pos[1];
                    c0.pz = pos[2];
                    c0.ts = (v.t_far + v.t_close) / 2.f;
                    c0.raw = raw;
                    c0.ext = fake_distance(g, o, v, pos, &miu, &sd);
                    c0.meta = 1;
                    c0.pad = 0;
                    nc = 1;
                }
            }
            if (overflow) nc = 0;
        }
        // one pool allocation per wave: exclusive prefix of the lanes' counts + a single atomic
        int total;
        const int pre = wave_excl_scan(nc, &total);
        const int shard = (int)((blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6) + turn * 61u) % kPoolShards);
        int base0 = 0;
        if (total > 0) {
            if (lane == 0) base0 = atomicAdd(&P.counters[2 + shard], total);
            base0 = __shfl(base0, 0);
        }
        if (mine) {
        int packed = 0;
        const int rel = base0 + pre;
        if (overflow || (nc > 0 && rel + nc > P.shard_cap)) {
            // hand the ray back once (other voxels of it may still run; k_comp ignores the ray)
            if (atomicExch(&P.n_items[q], -1) >= 0) P.flagged[atomicAdd(&P.counters[0], 1)] = (int)q;
        } else if (nc > 0) {
            const int at = shard * P.shard_cap + rel;
            P.pool[at] = c0;
            if (nc > 1) P.pool[at + 1] = c1;
            if (nc > 2) P.pool[at + 2] = c2;
            packed = (at << 2) | nc;
            // tell k_comp that this slot matters (one in nine of a ray's work voxels does)
            atomicOr(&P.cand_mask[q], 1ull << min(slot, 63));
        }
        if (SPILL) P.sp_citem[entry] = packed;
        else if (packed != 0) P.citem[entry] = packed;  // (table slots < 63: read through cand_mask only)
        }
      }
      ++turn;
    }
    tl.end();
}

// ------------------------------------------------------------------------------------
#ifndef ASURF_COMP_WAVES
#define ASURF_COMP_WAVES 5
#endif
template <int B, bool CACHE>
__global__ void __launch_bounds__(kBlock, (B <= 9 ? ASURF_COMP_WAVES : 4))
k_comp(KGrid g, KOpt o, const float* __restrict__ dirs, long long Q, PipeBuf P,
       float* __restrict__ rgb_out, int M, float* __restrict__ cache_a, float* __restrict__ cache_w,
       float* __restrict__ cache_t, int* __restrict__ cache_n, int* __restrict__ rec_v,
       int* __restrict__ rec_m, int* __restrict__ rec_n, int R, float* __restrict__ rec_f, int rpw) {
    __shared__ float s_sph[kBlock * kSphPad];
    __shared__ LaneRec s_rec[kBlock];
    const int tid = threadIdx.x;
    const float* sph_wave = s_sph + (tid & ~63) * kSphPad;
    const LaneRec* rec_wave = s_rec + (tid & ~63);
    // rpw rays a wave (lanes 0 .. rpw - 1 hold them, all 64 shade): a wave's time is the sum of its rays' shading
    // turns, and a batch that does not fill the chip is as slow as one wave -- fewer rays a wave then (rays_per_wave,
    // asurf_api.hip)
    const long long q = ((long long)blockIdx.x * (kBlock / 64) + (tid >> 6)) * rpw + (tid & 63);
    const bool in_range = (tid & 63) < rpw && q < Q;
    const int n = in_range ? P.n_items[q] : 0;
    const bool valid = in_range && n >= 0;  // handed-back rays belong to the one-kernel path
    if (valid) {
        float d[3] = {dirs[3 * q], dirs[3 * q + 1], dirs[3 * q + 2]};
        float sph[B];
        eval_sh<B>(d, sph);
#pragma unroll
        for (int k = 0; k < B; ++k) s_sph[tid * kSphPad + k] = sph[k];
    }
    const bool do_rec = CACHE && (rec_v != nullptr);
    // a ray handed to the one-kernel path leaves no records: the replaying backward must pass it by whether or not the
    // one-kernel forward has run yet (in the fused call it runs BESIDE the backward kernels, asurf_api.hip)
    if (in_range && n < 0 && do_rec) rec_n[q] = -1;
    const int cutv = (valid && do_rec) ? P.cut[q] : 0x7fffffff;
    const int spb = (valid && n > P.C1) ? P.sp_base[q] : 0;
    float outv[3] = {0.f, 0.f, 0.f};
    float log_transmit = 0.f;
    int sample_i = 0, intersect_i = -1, rec_cnt = 0;
    bool rec_over = false;
    // Where a ray's candidates are.  One work voxel in nine yields any, so a ray does not walk its slots: k_geom
    // left it a bit per slot that matters, and it visits those in order.  Only a ray with a candidate beyond its
    // 63rd work voxel scans its slots from there on (`scan`; k_geom leaves a word in every cell from slot 63 up, and
    // below it only where a candidate is).  Either way the words of the next four stops are fetched together
    // (independent loads): the loop is otherwise one dependent memory round trip per stop.
    const unsigned long long cmask = valid ? P.cand_mask[q] : 0ull;
    const bool scan = (cmask >> 63) != 0ull;
    unsigned long long todo = cmask & ~(1ull << 63);   // slots below 63 still to visit
    int s_scan = 63;                                   // next slot of the scan beyond them
    // stops this ray still makes; 0 when stop_thresh ends it
    int n_left = valid ? __popcll(todo) + (scan ? max(n - 63, 0) : 0) : 0;
    int ci4[4] = {0, 0, 0, 0}, pv4[4] = {0, 0, 0, 0}, s4[4] = {0, 0, 0, 0};
    for (int it = 0; __any(it < n_left); ++it) {
        const bool have = it < n_left;
        if ((it & 3) == 0) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                bool on = it + u < n_left;
                int su = 0;
                if (todo != 0ull) {
                    su = __ffsll((long long)todo) - 1;
                    todo &= todo - 1ull;
                } else if (on) {
                    su = s_scan++;
                }
                s4[u] = su;
                ci4[u] = pv4[u] = 0;
                if (on) {
                    ci4[u] = (su < P.C1) ? P.citem[(size_t)su * Q + q] : P.sp_citem[spb + su - P.C1];
                    pv4[u] = (su < P.C1) ? P.items[(size_t)su * Q + q] : P.sp_items[spb + su - P.C1];
                }
            }
        }
        const int sel = it & 3;
        const int s = sel == 0 ? s4[0] : sel == 1 ? s4[1] : sel == 2 ? s4[2] : s4[3];
        int ci = sel == 0 ? ci4[0] : sel == 1 ? ci4[1] : sel == 2 ? ci4[2] : ci4[3];
        const int pv = sel == 0 ? pv4[0] : sel == 1 ? pv4[1] : sel == 2 ? pv4[2] : pv4[3];
        if (!have) ci = 0;
        const int nc = ci & 3;
        if (!__any(nc > 0)) continue;
        const int base = ci >> 2;
        // the voxel's links are asked for together with its first candidate record (both addresses are known
        // here), not after the record has arrived and passed the gates: one memory round trip less per sample
        int lk8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (nc > 0) {
            const int vx = pv >> 20, vy = (pv >> 10) & 1023, vz = pv & 1023;
            const int32_t* p = g.links + ((size_t)vx * g.offx + (size_t)vy * g.offy + vz);
            lk8[0] = p[0];
            lk8[1] = p[1];
            lk8[2] = p[g.offy];
            lk8[3] = p[g.offy + 1];
            lk8[4] = p[g.offx];
            lk8[5] = p[g.offx + 1];
            lk8[6] = p[g.offx + g.offy];
            lk8[7] = p[g.offx + g.offy + 1];
        }
        bool links_loaded = false;
        bool composited = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const bool hc = c < nc;
            if (!__any(hc)) break;
            CandRec cr;
            cr.meta = 0;
            cr.px = cr.py = cr.pz = cr.ts = cr.raw = cr.ext = 0.f;
            if (hc) cr = P.pool[base + c];
            const bool fake = hc && (cr.meta & 1);
            bool shade = false;
            float a = 0.f;  // re-weighted alpha
            if (hc && !fake) {  // surf_trav.cu:369-385
                ++intersect_i;
                if (cr.raw > o.sigma_thresh) {
                    const float alpha = alpha_act(cr.raw, o.alpha_activation_type);
                    const float rw = o.truncated_vol_render
                                         ? trunc_rw((float)intersect_i, g.trunc_a, o.trunc_vol_weight_min)
                                         : 1.f;
                    a = alpha * rw;
                    shade = true;
                }
            } else if (fake) {  // :455-507 (the sigma gate was applied by k_geom)
                float alpha = alpha_act(cr.raw, o.alpha_activation_type);
                alpha = alpha * expf((float)(-.5 * (double)sqrf(cr.ext / g.fake_sample_std)));
                const float rw = o.truncated_vol_render
                                     ? trunc_rw((float)intersect_i, g.trunc_a, o.trunc_vol_weight_min)
                                     : 1.f;
                a = alpha * rw;
                shade = true;
            }
            if (shade && !links_loaded) {
#pragma unroll
                for (int k8 = 0; k8 < 8; ++k8) s_rec[tid].link[k8] = lk8[k8];
                links_loaded = true;
            }
            const unsigned long long need = __ballot(shade);
            if (!need) continue;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            float color[3] = {0.f, 0.f, 0.f};
            const float pos[3] = {cr.px, cr.py, cr.pz};
            pair_shade_fwd<B>(g, need, rec_wave, sph_wave, pos, color);
            if (!shade) continue;
            const float pcnt = -1 * logf(1 - a);  // :397-404
            const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
            log_transmit -= pcnt;
#pragma unroll
            for (int k = 0; k < 3; ++k) outv[k] += weight * fmaxf(color[k] + 0.5f, 0.f);
            composited = true;
            if (CACHE && sample_i < M && (!fake || o.fake_sample_l_dist)) {
                const size_t idx = (size_t)sample_i * Q + q;
                cache_a[idx] = a;
                cache_w[idx] = weight;
                cache_t[idx] = cr.ts;
                sample_i += 1;
            }
            if (do_rec && s < cutv) {
                if (rec_cnt < R) {
                    rec_v[(size_t)rec_cnt * Q + q] = pv;
                    rec_m[(size_t)rec_cnt * Q + q] = ((intersect_i + 1) << 8) | (cr.meta & 0xff);
                    write_rec_floats(rec_f, R, rec_cnt, Q, q, cr.raw, a, color, pos);
                    ++rec_cnt;
                } else {
                    rec_over = true;
                }
            }
        }
        // once per voxel (:544-547); log_transmit only changes where something was composited
        if (composited && expf(log_transmit) < o.stop_thresh) {
            log_transmit = -1e3f;
            n_left = 0;
        }
    }
    if (valid) {
        const float bg = expf(log_transmit) * o.background_brightness;
#pragma unroll
        for (int k = 0; k < 3; ++k) rgb_out[3 * q + k] = outv[k] + bg;
        if (CACHE) cache_n[q] = sample_i;
        if (do_rec) rec_n[q] = rec_over ? -1 : rec_cnt;
        if (do_rec && rec_over) P.counters[kCtrRecOver] = 1;
    }
}

}  // namespace asurf
