// asurf_masks2.h -- the occupancy pyramid from vertex bit planes (round 3).
//
// k_build_masks (asurf_kernels.h) decides every voxel from scratch: 8 links, then 8 densities, then 8 surface values,
// one x-slab after the other -- a wave that holds anything is a chain of 24 dependent round trips, and every vertex
// value is fetched by the eight voxels around it.  The pass took 0.067 ms at 256^3 in every variant that kept that
// structure, a third of a call at the training script's batch size.
// Here the decision is split where the data dependence is:
//   per VERTEX, what the voxel tests need of it, as bits over runs of 64 vertices along z (a __ballot each): E "the
//     vertex exists" (link >= 0), B "its density is below sigma_thresh", and per level set G "its surface value is
//     above the level" and H "... below the level".  A block takes one row of 8 bricks along z -- 8 x 8 vertex columns
//     of 64 -- plus the halo its voxels reach into: the ninth column along x and y and the next vertex along z.  Every
//     wave asks for the links of its eleven runs together, then for the values of the vertices that exist, together: two
//     round trips, whatever the wave holds;
//   per VOXEL, bit arithmetic.  With P' = P & (P moved by one along z) and all4(P) = the AND of P' over the four (x, y)
//     neighbours:
//         live = all4(E) & ~all4(B)                  8 links >= 0, not all 8 densities below (surf_trav.cu:230-239)
//         work = live & OR_i (~all4(G_i) & ~all4(H_i))     level i neither below the minimum nor above the maximum
//                                                    of the 8 surface values (:265-277)
//     lane = (x, y) column, word = the 64 voxels along z; then the three cuts of the brick words and the coarse bytes,
//     exactly the layout k_build_masks leaves.
// (A first form kept the vertex planes in memory between two kernels, 0.036 + 0.021 ms at 256^3; recomputing the halo
// -- 27 % more vertices -- in one kernel costs less than the second launch and the planes' round trip.)
// NaN: a NaN density is "not below" in both forms.  A NaN surface value is neither above nor below a level here, so a
// voxel that holds one stays a work voxel (k_geom then applies the reference's own arithmetic to it); the reference's
// fminf / fmaxf skip the NaN instead -- the two differ only in which voxels are LOOKED AT, not in what is rendered.
// Serves the surf_trav planes with at most kMaskFastSets level sets; the cuvol plane and longer level-set lists keep
// k_build_masks.
#pragma once
#include "asurf_kernels.h"

namespace asurf {

constexpr int kMaskFastSets = 3;
constexpr int kMaskPlanes = 2 + 2 * kMaskFastSets;

// rows of the pass: all of them (blockIdx), or the listed ones (stride over the list the header vouches for; a header
// that is not what the host table says -- asurf_api.hip, mask_cache_lookup -- makes the pass take every row)
struct MaskRows {
    MaskHdr* hdr;
    const int* live_rows;
    unsigned long long epoch;
    int n_rows;
    int use_list;      // 0: every row; 1: the list, if the header carries `epoch`; 2: the list k_rows_alive has just left;
                       // 3: the list, if the header carries the checksum of `links` THIS call has taken (hdr->acc[parity])
    int parity;        // (3) which set of slots this call's checksum was added up in
    int* h_invalid;    // (3) pinned host word, set when the checksum did not match: the host then drops the list
};

struct MaskTables2 {
    unsigned long long comb[kMaskPlanes][9][9];   // P' of the 9 x 9 vertex columns of the row and its halo
    unsigned long long w[64], l[64];              // work / live of the 8 x 8 voxel columns ([lx * 8 + ly])
    int alive;
};

// One row of 8 bricks along z (`row`), by all 512 threads of the block.  `listed`: the row is known to hold vertices (no
// emptiness test).  `more`: the block takes another row after this one -- the row then ends with a barrier, so that the
// next row's first stores to S.alive and the tables do not overtake a wave that still reads them.
__device__ __forceinline__ void masks2_row(const KGrid& g, float sigma_thresh, int need_surf, int n_sets,
                                           unsigned long long* fine_work, unsigned long long* fine_live,
                                           unsigned int* coarse_work, unsigned int* coarse_live, unsigned long long* fine_y,
                                           unsigned long long* fine_z, int orient_live, int nzg, int row, bool listed,
                                           MaskHdr* hdr_out, int* live_rows_out, MaskTables2& S, bool more) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const bool need_dens = sigma_thresh > -INFINITY;   // (no density is below -inf: the links-only planes read none)
    const int zg = row % nzg, bxy = row / nzg;
    const int by = bxy % g.nb[1], bx = bxy / g.nb[1];
    // The wave's runs, r = 0 .. 10, as vertex columns (xi, yi) of the 9 x 9 patch: its own x-slab with the 9 y's,
    // its share of the halo slab xi = 8 (column yi = w; wave 0 also takes the corner yi = 8).  Lane l holds vertex
    // z = 64 zg + l of every run; lane r < 11 also holds the vertex z = 64 (zg + 1) of run r (the halo along z).
    const int z = zg * 64 + l, zn = zg * 64 + 64;
    int32_t lk[11];
    if (!listed) {
        // The pass over the whole volume: most rows of a sparse grid are empty space, and a row none of whose own
        // vertices exists holds no voxel whatever its halo holds -- its words are zeros, nothing more is read.  (A
        // listed row is known to hold vertices: no such test, and one round trip less.)  (Reading the row's links as
        // 16-byte pieces for this test -- a lane 4 consecutive z, a wave 4 runs an instruction -- changes nothing:
        // 0.066 against 0.063 ms at 256^3, 0.345 against 0.329 at 512^3.)
        bool any = false;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int x = bx * 8 + w, y = by * 8 + r;
            lk[r] = (x < g.size[0] && y < g.size[1] && z < g.size[2]) ? g.links[(size_t)x * g.offx + (size_t)y * g.offy + z] : -1;
            any = any || (lk[r] >= 0);
        }
        if (threadIdx.x == 0) S.alive = 0;
        __syncthreads();
        if (__ballot(any) != 0ull && l == 0) S.alive = 1;
        __syncthreads();
        if (!S.alive) {   // (block-uniform)
            const size_t brick0 = ((size_t)bx * g.nb[1] + by) * g.nb[2] + (size_t)zg * 8;
            const int nbr = min(8, g.nb[2] - zg * 8);
            const int j = l >> 3, k = l & 7;
            if (j < nbr) {
                if (w == 0) {
                    fine_work[(brick0 + j) * 8 + k] = 0ull;
                    fine_live[(brick0 + j) * 8 + k] = 0ull;
                } else if (w == 1 && fine_y != nullptr) {
                    fine_y[(brick0 + j) * 8 + k] = 0ull;
                    fine_z[(brick0 + j) * 8 + k] = 0ull;
                }
            }
            if (threadIdx.x == 128) {
                reinterpret_cast<unsigned char*>(coarse_work)[(size_t)bxy * g.nwz * 4 + zg] = 0;
                reinterpret_cast<unsigned char*>(coarse_live)[(size_t)bxy * g.nwz * 4 + zg] = 0;
            }
            if (more) __syncthreads();
            return;
        }
    }
    if (listed) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {   // the row's own vertices
            const int x = bx * 8 + w, y = by * 8 + r;
            lk[r] = (x < g.size[0] && y < g.size[1] && z < g.size[2]) ? g.links[(size_t)x * g.offx + (size_t)y * g.offy + z] : -1;
        }
    }
#pragma unroll
    for (int r = 8; r < 11; ++r) {   // the halo: the ninth column along y, the wave's share of the ninth slab along x
        const int xi = r < 9 ? w : 8, yi = r < 9 ? r : (r == 9 ? w : 8);
        const int x = bx * 8 + xi, y = by * 8 + yi;
        const bool have = (r < 10 || w == 0) && x < g.size[0] && y < g.size[1] && z < g.size[2];
        lk[r] = have ? g.links[(size_t)x * g.offx + (size_t)y * g.offy + z] : -1;
    }
    int32_t lkn = -1;
    {
        const int xi = l < 9 ? w : 8, yi = l < 9 ? l : (l == 9 ? w : 8);
        const int x = bx * 8 + xi, y = by * 8 + yi;
        if ((l < 10 || (l == 10 && w == 0)) && x < g.size[0] && y < g.size[1] && zn < g.size[2])
            lkn = g.links[(size_t)x * g.offx + (size_t)y * g.offy + zn];
    }
    float dv[11], sv[11];
#pragma unroll
    for (int r = 0; r < 11; ++r) {
        dv[r] = (need_dens && lk[r] >= 0) ? g.density[lk[r]] : 0.f;
        sv[r] = (need_surf && lk[r] >= 0) ? g.surface[lk[r]] : 0.f;
    }
    const float dvn = (need_dens && lkn >= 0) ? g.density[lkn] : 0.f;
    const float svn = (need_surf && lkn >= 0) ? g.surface[lkn] : 0.f;
    float lv[kMaskFastSets];
#pragma unroll
    for (int s = 0; s < kMaskFastSets; ++s) lv[s] = (need_surf && s < n_sets) ? g.level_set[s] : 0.f;
    // bits of the halo vertices along z: bit r = run r's next vertex
    const bool exn = lkn >= 0;
    const unsigned long long nE = __ballot(exn), nB = __ballot(exn && dvn < sigma_thresh);
    unsigned long long nG[kMaskFastSets], nH[kMaskFastSets];
#pragma unroll
    for (int s = 0; s < kMaskFastSets; ++s) {
        nG[s] = __ballot(exn && svn > lv[s]);
        nH[s] = __ballot(exn && svn < lv[s]);
    }
    // P' = P & (P moved by one along z) of each run; lane r keeps run r's words
    unsigned long long cE = 0ull, cB = 0ull, cG[kMaskFastSets], cH[kMaskFastSets];
#pragma unroll
    for (int s = 0; s < kMaskFastSets; ++s) cG[s] = cH[s] = 0ull;
    bool own_alive = false;   // some vertex of the row itself exists (a function of `links` alone)
#pragma unroll
    for (int r = 0; r < 11; ++r) {
        const bool ex = lk[r] >= 0;
        const unsigned long long bE = __ballot(ex);
        if (r < 8 && w < 8) own_alive = own_alive || (bE != 0ull);
        if (bE == 0ull) continue;   // (wave-uniform) nothing exists in this run: its words stay 0
        const unsigned long long bB = __ballot(ex && dv[r] < sigma_thresh);
        const unsigned long long pE = bE & ((bE >> 1) | (((nE >> r) & 1ull) << 63));
        const unsigned long long pB = bB & ((bB >> 1) | (((nB >> r) & 1ull) << 63));
        cE = (l == r) ? pE : cE;
        cB = (l == r) ? pB : cB;
        if (need_surf) {
#pragma unroll
            for (int s = 0; s < kMaskFastSets; ++s) {
                if (s < n_sets) {
                    const unsigned long long bG = __ballot(ex && sv[r] > lv[s]), bH = __ballot(ex && sv[r] < lv[s]);
                    const unsigned long long pG = bG & ((bG >> 1) | (((nG[s] >> r) & 1ull) << 63));
                    const unsigned long long pH = bH & ((bH >> 1) | (((nH[s] >> r) & 1ull) << 63));
                    cG[s] = (l == r) ? pG : cG[s];
                    cH[s] = (l == r) ? pH : cH[s];
                }
            }
        }
    }
    if (threadIdx.x == 0) S.alive = 0;
    __syncthreads();   // (the tables of the row before are no longer read)
    if (l < 11 && (l < 10 || w == 0)) {
        const int xi = l < 9 ? w : 8, yi = l < 9 ? l : (l == 9 ? w : 8);
        S.comb[0][xi][yi] = cE;
        S.comb[1][xi][yi] = cB;
#pragma unroll
        for (int s = 0; s < kMaskFastSets; ++s) {
            if (s < n_sets) {
                S.comb[2 + 2 * s][xi][yi] = cG[s];
                S.comb[3 + 2 * s][xi][yi] = cH[s];
            }
        }
    }
    if (own_alive && l == 0) S.alive = 1;
    __syncthreads();
    if (w < 3) {
        // ---- voxel stage: lane = (lx, ly) column; computed by three waves alike, each leaves one kind of output ----
        const int lx = l >> 3, ly = l & 7;
#define ASURF_ALL4(q) (S.comb[q][lx][ly] & S.comb[q][lx][ly + 1] & S.comb[q][lx + 1][ly] & S.comb[q][lx + 1][ly + 1])
        const unsigned long long live = ASURF_ALL4(0) & ~ASURF_ALL4(1);
        unsigned long long work = live;
        if (need_surf) {
            unsigned long long any = 0ull;
#pragma unroll
            for (int s = 0; s < kMaskFastSets; ++s)
                if (s < n_sets) any |= ~ASURF_ALL4(2 + 2 * s) & ~ASURF_ALL4(3 + 2 * s);
            work = live & any;
        }
#undef ASURF_ALL4
        const size_t brick0 = ((size_t)bx * g.nb[1] + by) * g.nb[2] + (size_t)zg * 8;
        const int nbr = min(8, g.nb[2] - zg * 8);  // bricks of the row that exist
        const int j = l >> 3, k = l & 7;            // output lane: brick j of the row, word k of the brick
        const unsigned long long sel = orient_live ? live : work;
        if (w == 0) {
            // x cut (word = x-slab k, bit = 8 ly + lz) and y cut (word = y index k, bit = 8 lz + lx) through LDS
            S.w[l] = work;
            S.l[l] = live;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            unsigned long long xw = 0ull, xl = 0ull, yw = 0ull;
            const unsigned long long* s_sel = orient_live ? S.l : S.w;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                xw |= ((S.w[k * 8 + t] >> (8 * j)) & 0xffull) << (8 * t);
                xl |= ((S.l[k * 8 + t] >> (8 * j)) & 0xffull) << (8 * t);
                yw |= spread_bits_to_bytes((unsigned)((s_sel[t * 8 + k] >> (8 * j)) & 0xffull)) << t;
            }
            if (j < nbr) {
                fine_work[(brick0 + j) * 8 + k] = xw;
                fine_live[(brick0 + j) * 8 + k] = xl;
                if (fine_y != nullptr) fine_y[(brick0 + j) * 8 + k] = yw;
            }
        } else if (w == 1) {
            // z cut: word = z index lz, bit = 8 lx + ly = the lane number: a ballot per (brick, lz)
            if (fine_z != nullptr) {
                unsigned long long zw = 0ull;
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
#pragma unroll
                    for (int lz = 0; lz < 8; ++lz) {
                        const unsigned long long m = __ballot((sel >> (8 * jj + lz)) & 1ull);
                        zw = (l == jj * 8 + lz) ? m : zw;
                    }
                }
                if (j < nbr) fine_z[(brick0 + j) * 8 + k] = zw;
            }
        } else {
            // coarse bytes (bit j = brick j of the row holds a flagged voxel) and the list of rows that hold vertices
            unsigned cw = 0u, cl = 0u;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                cw |= (__ballot(((work >> (8 * jj)) & 0xffull) != 0ull) != 0ull ? 1u : 0u) << jj;
                cl |= (__ballot(((live >> (8 * jj)) & 0xffull) != 0ull) != 0ull ? 1u : 0u) << jj;
            }
            if (l == 0) {
                reinterpret_cast<unsigned char*>(coarse_work)[(size_t)bxy * g.nwz * 4 + zg] = (unsigned char)cw;
                reinterpret_cast<unsigned char*>(coarse_live)[(size_t)bxy * g.nwz * 4 + zg] = (unsigned char)cl;
                if (hdr_out != nullptr && S.alive) live_rows_out[atomicAdd(&hdr_out->n_live, 1)] = row;
            }
        }
    }
    if (more) __syncthreads();   // (block-uniform)
}

__global__ void __launch_bounds__(512)
k_build_masks2(KGrid g, float sigma_thresh, int need_surf, int n_sets, unsigned long long* fine_work,
               unsigned long long* fine_live, unsigned int* coarse_work, unsigned int* coarse_live,
               unsigned long long* fine_y, unsigned long long* fine_z, int orient_live, int nzg, MaskRows R,
               MaskHdr* hdr_out, int* live_rows_out, int* zero_ptr, int zero_n) {
    const TlScope tl(1);
    __shared__ MaskTables2 S;
    // (the pipeline's counters, zeroed here when no memset runs in front of the pass: asurf_api.hip, prepare())
    if (blockIdx.x == 0 && zero_ptr != nullptr)
        for (int i = threadIdx.x; i < zero_n; i += blockDim.x) zero_ptr[i] = 0;
    // use_list 2: the list was left by k_rows_alive in front of this very launch (the split full pass): no header to check
    bool listed = R.use_list == 2 || (R.use_list == 1 && R.hdr->magic == kMaskMagic && R.hdr->epoch == R.epoch);
    if (R.use_list == 3) {
        // The default call (no promise about `links`): k_links_checksum has just summed ALL of `links` into the slots of
        // this call's parity; the list stands if that is the checksum it was built under.  Every block reaches the same
        // verdict from the same words (the header's sum is written by the sealing kernel of an earlier call only).
        __shared__ int s_ok;
        if (threadIdx.x < 64) {
            unsigned long long a = 0ull, b = 0ull;
            if (threadIdx.x < kSumSlots) {
                a = __hip_atomic_load(&R.hdr->acc[R.parity][threadIdx.x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                b = __hip_atomic_load(&R.hdr->acc[R.parity][threadIdx.x][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                a += __shfl_xor(a, m);
                b += __shfl_xor(b, m);
            }
            if (threadIdx.x == 0)
                s_ok = (R.hdr->magic == kMaskMagic && R.hdr->sum_known != 0 && R.hdr->epoch == 0ull && R.hdr->sum[0] == a &&
                        R.hdr->sum[1] == b) ? 1 : 0;
        }
        __syncthreads();
        listed = s_ok != 0;
        if (blockIdx.x == 0) {
            // the other parity's slots belong to the next call: cleared here, where nobody adds to them
            if (threadIdx.x < kSumSlots) {
                R.hdr->acc[1 - R.parity][threadIdx.x][0] = 0ull;
                R.hdr->acc[1 - R.parity][threadIdx.x][1] = 0ull;
            }
            if (!listed && threadIdx.x == 0 && R.h_invalid != nullptr) *R.h_invalid = 1;   // (the host drops the list)
        }
    }
    const int n = listed ? min(R.hdr->n_live, R.n_rows) : R.n_rows;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int row = listed ? R.live_rows[i] : i;
        if (row < 0 || row >= R.n_rows) continue;   // (block-uniform)
        masks2_row(g, sigma_thresh, need_surf, n_sets, fine_work, fine_live, coarse_work, coarse_live, fine_y, fine_z,
                   orient_live, nzg, row, listed, hdr_out, live_rows_out, S, i + (int)gridDim.x < n);
    }
    tl.end();
}

// The split full pass (grids of 16 384 rows and more): every plane is cleared by one memset, this kernel streams `links`
// at 16 bytes a lane -- a wave a row: 16 lanes a vertex column of 64, four columns an instruction -- and lists the rows
// that hold a vertex, and k_build_masks2 then takes the listed rows only.  Inside k_build_masks2 the same question costs
// a 512-thread block, two barriers and the zeros of the row's words per EMPTY row (0.33 ms at 512^3, 0.95 at 768^3,
// against 0.02 + 0.11 + 0.11 and 0.05 + 0.37 + 0.38 this way); at 256^3 the two forms cost the same and the single kernel
// stays.  Needs size[2] % 4 == 0 (16-byte loads).
__global__ void __launch_bounds__(256)
k_rows_alive(KGrid g, int nzg, int n_rows, MaskHdr* hdr, int* live_rows) {
    const int lane = threadIdx.x & 63;
    const int row = (int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (row >= n_rows) return;
    const int zg = row % nzg, bxy = row / nzg;
    const int by = bxy % g.nb[1], bx = bxy / g.nb[1];
    const int z = zg * 64 + (lane & 15) * 4;
    int all_neg = -1;   // AND of the links seen: negative while none exists
#pragma unroll 4
    for (int c = lane >> 4; c < 64; c += 4) {
        const int x = bx * 8 + (c >> 3), y = by * 8 + (c & 7);
        if (x < g.size[0] && y < g.size[1] && z < g.size[2]) {
            // (z and size[2] are multiples of 4: the piece lies inside the column)
            const int4 v = *reinterpret_cast<const int4*>(g.links + ((size_t)x * g.offx + (size_t)y * g.offy + z));
            all_neg &= v.x & v.y & v.z & v.w;
        }
    }
    if (__ballot(all_neg >= 0) != 0ull && lane == 0) live_rows[atomicAdd(&hdr->n_live, 1)] = row;
}

// ---------------------------------------------------------------------------------------------------------------------
// Small batches (round 4): planes that depend on `links` alone.
//
// At the training script's batch size (5 000 rays, opt/util/config_util.py:340) the occupancy pass over the volume was
// 0.063 of the call's 0.158 ms -- for a batch that looks at a tenth of the voxels.  What makes the pass per-call are the
// VALUES (density gate, level set inside the voxel's range: surf_trav.cu:230-239, 265-277); `links` changes only when the
// grid is pruned or resampled.  So for small batches the planes hold "all 8 corner links exist" only (built by
// masks2_row with no level sets and a gate nothing is below), they stay in the workspace from call to call, and k_sweep
// applies the value gates to the few thousand voxels its rays actually meet (lazy_gates_pass, asurf_sweep.h) -- the
// tables it leaves are the ones the per-call planes give, entry for entry.
// What lets a call trust planes it did not build: the caller's promise (asurf_grid::links_epoch != 0), or, for the
// default call, a 128-bit checksum of `links` taken by THIS call (k_links_checksum: one pass at stream rate over the
// 64 MiB of a 256^3 grid) equal to the one the planes were built under.  k_links_planes decides on the device -- the
// header is the truth, the host's table (asurf_api.hip) only says whether the header was ever initialised -- and
// rebuilds every row when the header does not vouch for the planes; the block that finishes last seals the header.
constexpr unsigned long long kMaskMagicLinks = 0x61737572666c6e6bull;

// Σ over 16-byte pieces of two independently mixed 32-bit words of (piece index, 4 links): order-free, so the blocks add
// their shares to kSumSlots slots with plain atomics.  (Two 4-byte elements that trade places change both words unless
// their mixed sums collide, 2^-64.)
__device__ __forceinline__ unsigned sum_mix(unsigned h, unsigned m1, unsigned m2) {
    h ^= h >> 16;
    h *= m1;
    h ^= h >> 13;
    h *= m2;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ void sum_piece(unsigned p, int4 v, unsigned long long& s1, unsigned long long& s2) {
    const unsigned x = (unsigned)v.x, y = (unsigned)v.y, z = (unsigned)v.z, w = (unsigned)v.w;
    const unsigned a = x * 0x9E3779B1u + y * 0x85EBCA77u + z * 0xC2B2AE3Du + w * 0x27D4EB2Fu + p * 0x165667B1u;
    const unsigned b = x * 0xD6E8FEB9u + y * 0xA54FF53Bu + z * 0x510E527Fu + w * 0x9B05688Du + (p ^ 0x5bd1e995u) * 0xCC9E2D51u;
    s1 += sum_mix(a, 0x85EBCA6Bu, 0xC2B2AE35u);
    s2 += sum_mix(b, 0x7FEB352Du, 0x846CA68Bu);
}

__global__ void __launch_bounds__(256)
k_links_checksum(const int32_t* __restrict__ links, long long n, MaskHdr* hdr, int parity) {
    unsigned long long s1 = 0ull, s2 = 0ull;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, nth = (long long)gridDim.x * blockDim.x;
    if ((reinterpret_cast<uintptr_t>(links) & 15) == 0) {
        const long long np = n >> 2;
        const int4* p4 = reinterpret_cast<const int4*>(links);
        long long i = tid;
        for (; i + 3 * nth < np; i += 4 * nth) {   // four pieces in flight a lane
            const int4 v0 = p4[i], v1 = p4[i + nth], v2 = p4[i + 2 * nth], v3 = p4[i + 3 * nth];
            sum_piece((unsigned)i, v0, s1, s2);
            sum_piece((unsigned)(i + nth), v1, s1, s2);
            sum_piece((unsigned)(i + 2 * nth), v2, s1, s2);
            sum_piece((unsigned)(i + 3 * nth), v3, s1, s2);
        }
        for (; i < np; i += nth) sum_piece((unsigned)i, p4[i], s1, s2);
        for (long long e = (np << 2) + tid; e < n; e += nth) sum_piece((unsigned)e, make_int4(links[e], 1, 2, 3), s1, s2);
    } else {
        for (long long e = tid; e < n; e += nth) sum_piece((unsigned)e, make_int4(links[e], 1, 2, 3), s1, s2);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s1 += __shfl_xor(s1, m);
        s2 += __shfl_xor(s2, m);
    }
    __shared__ unsigned long long sh[2][4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sh[0][w] = s1;
        sh[1][w] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int slot = blockIdx.x % kSumSlots;
        atomicAdd(&hdr->acc[parity][slot][0], sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]);
        atomicAdd(&hdr->acc[parity][slot][1], sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3]);
    }
}

// use_sum: the header must carry the checksum this call has just taken (hdr->acc); else the caller's epoch.
__global__ void __launch_bounds__(512)
k_links_planes(KGrid g, unsigned long long* fine_work, unsigned long long* fine_live, unsigned int* coarse_work,
               unsigned int* coarse_live, unsigned long long* fine_y, unsigned long long* fine_z, int nzg, int n_rows,
               MaskHdr* hdr, unsigned long long epoch, int use_sum, int* zero_ptr, int zero_n) {
    const TlScope tl(1);
    __shared__ MaskTables2 S;
    __shared__ unsigned long long s_sum[2];
    __shared__ int s_valid;
    if (blockIdx.x == 0 && zero_ptr != nullptr)
        for (int i = threadIdx.x; i < zero_n; i += blockDim.x) zero_ptr[i] = 0;
    if (threadIdx.x < 64) {   // wave 0: this call's checksum = the sum of the slots; the verdict
        unsigned long long a = 0ull, b = 0ull;
        if (use_sum && threadIdx.x < kSumSlots) {
            a = __hip_atomic_load(&hdr->acc[0][threadIdx.x][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            b = __hip_atomic_load(&hdr->acc[0][threadIdx.x][1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            a += __shfl_xor(a, m);
            b += __shfl_xor(b, m);
        }
        if (threadIdx.x == 0) {
            s_sum[0] = a;
            s_sum[1] = b;
            const bool ok = hdr->magic == kMaskMagicLinks &&
                            (use_sum ? (hdr->sum_known != 0 && hdr->sum[0] == a && hdr->sum[1] == b)
                                     : (epoch != 0ull && hdr->epoch == epoch));
            s_valid = ok ? 1 : 0;
        }
    }
    __syncthreads();
    const bool valid = s_valid != 0;   // (the same in every block: nothing below the header is written before the last ticket)
    if (!valid) {
        for (int row = blockIdx.x; row < n_rows; row += gridDim.x)
            masks2_row(g, -INFINITY, 0, 0, fine_work, fine_live, coarse_work, coarse_live, fine_y, fine_z, 0, nzg, row, false,
                       nullptr, nullptr, S, true);
    }
    // the block whose ticket is the last seals the header and clears this call's slots for the next one
    __syncthreads();
    if (threadIdx.x == 0) {   // (what the blocks wrote is read by later kernels only: no fence)
        const unsigned t = atomicAdd(&hdr->ticket, 1u);
        s_valid = (t == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (s_valid) {
        if (threadIdx.x < kSumSlots) {
            hdr->acc[0][threadIdx.x][0] = 0ull;
            hdr->acc[0][threadIdx.x][1] = 0ull;
        }
        if (threadIdx.x == 0) {
            hdr->ticket = 0u;
            if (!valid) {
                hdr->sum[0] = use_sum ? s_sum[0] : 0ull;
                hdr->sum[1] = use_sum ? s_sum[1] : 0ull;
                hdr->sum_known = use_sum ? 1 : 0;
                hdr->epoch = epoch;
                hdr->n_live = 0;
                hdr->magic = kMaskMagicLinks;
            }
        }
    }
    tl.end();
}

}  // namespace asurf
