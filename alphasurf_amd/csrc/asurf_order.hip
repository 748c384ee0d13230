// Ray ordering for coherence (an extension of this build; the reference renders a batch in the order given).
// The walk and the gather kernels run one ray per lane: a wave is as slow as its slowest lane and its loads
// coalesce only when neighbouring lanes are neighbouring rays.  Training batches are shuffled.  One key kernel
// and one radix sort give a permutation that groups the rays of a camera (equal origins hash alike) and
// orders each group along a Morton curve of the unit direction; the caller gathers the rays with it, renders,
// and scatters the colours back.
#include "asurf_order.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <rocprim/device/device_radix_sort.hpp>

#include "asurf.h"

namespace {

__device__ __forceinline__ unsigned spread3(unsigned x) {  // two zero bits between the low 10 bits
    x &= 0x3FFu;
    x = (x | (x << 16)) & 0x30000FFu;
    x = (x | (x << 8)) & 0x300F00Fu;
    x = (x | (x << 4)) & 0x30C30C3u;
    x = (x | (x << 2)) & 0x9249249u;
    return x;
}

__global__ void __launch_bounds__(256)
k_order_keys(const float* __restrict__ origins, const float* __restrict__ dirs, long long n,
             unsigned long long* __restrict__ keys, unsigned* __restrict__ ids) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
    const float rn = rsqrtf(fmaxf(dx * dx + dy * dy + dz * dz, 1e-30f));
    const unsigned qx = (unsigned)fminf(fmaxf((dx * rn + 1.f) * 511.5f, 0.f), 1023.f);
    const unsigned qy = (unsigned)fminf(fmaxf((dy * rn + 1.f) * 511.5f, 0.f), 1023.f);
    const unsigned qz = (unsigned)fminf(fmaxf((dz * rn + 1.f) * 511.5f, 0.f), 1023.f);
    const unsigned morton = spread3(qx) | (spread3(qy) << 1) | (spread3(qz) << 2);
    // equal origins -> equal hash; the order of the groups does not matter
    unsigned h = __float_as_uint(origins[3 * i]) * 0x9E3779B1u;
    h = (h ^ __float_as_uint(origins[3 * i + 1])) * 0x85EBCA77u;
    h = (h ^ __float_as_uint(origins[3 * i + 2])) * 0xC2B2AE3Du;
    h ^= h >> 15;
    keys[i] = ((unsigned long long)(h & 0xFFFFFu) << 30) | morton;
    ids[i] = (unsigned)i;
}

// ---- the same ordering inside the fused call (asurf_api.hip, asurf_fused_params::order_rays) ----
// 30-bit keys: 6 bits of the origin's hash, then 24 bits of the direction in the octahedral map of the sphere -- three
// or four radix passes instead of seven and half the key bytes of the 50-bit form.  (With 9 bits per direction component the
// cells were four pixels wide and the rays inside a cell stayed shuffled: k_bwd_sh 0.29 ms against 0.17 in image order.)
__device__ __forceinline__ unsigned spread2(unsigned x) {  // one zero bit between the low 13 bits
    x &= 0x1FFFu;
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

__global__ void __launch_bounds__(256)
k_order_keys32(const float* __restrict__ origins, const float* __restrict__ dirs, long long n,
               unsigned* __restrict__ keys, unsigned* __restrict__ ids) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
    const float rn = 1.f / fmaxf(fabsf(dx) + fabsf(dy) + fabsf(dz), 1e-30f);
    float u = dx * rn, v = dy * rn;
    if (dz < 0.f) {   // the lower hemisphere folds over the corners of the square
        const float fu = (1.f - fabsf(v)) * (u >= 0.f ? 1.f : -1.f), fv = (1.f - fabsf(u)) * (v >= 0.f ? 1.f : -1.f);
        u = fu;
        v = fv;
    }
    // strips: v in 12 bits (a strip about a pixel wide), u in 12 bits along the strip -- scan-line order; the origin's
    // hash above them, so that a batch of one camera sorts on 24 bits (three radix passes: order32's key_bits).  The kernels
    // that merge the samples of consecutive rays (k_bwd_sh, k_bwd_geo) want neighbours ALONG a line: a Z curve over
    // the pixels (13 + 13 bits interleaved) cost them 0.243 / 0.144 ms against 0.169 / 0.110 in image order.
    const unsigned qu = (unsigned)fminf(fmaxf((u + 1.f) * 2047.75f, 0.f), 4095.f);
    const unsigned qv = (unsigned)fminf(fmaxf((v + 1.f) * 2047.75f, 0.f), 4095.f);
    const unsigned morton = (qv << 12) | qu;
    unsigned h = __float_as_uint(origins[3 * i]) * 0x9E3779B1u;
    h = (h ^ __float_as_uint(origins[3 * i + 1])) * 0x85EBCA77u;
    h = (h ^ __float_as_uint(origins[3 * i + 2])) * 0xC2B2AE3Du;
    h ^= h >> 15;
    keys[i] = ((h & 63u) << 24) | morton;
    ids[i] = (unsigned)i;
}

// rays and targets in the new order: a lane per ray, 12-byte pieces (the reads are the scattered side: three a ray)
struct F3 {
    float x, y, z;
};
__global__ void __launch_bounds__(256)
k_order_gather(const float* __restrict__ origins, const float* __restrict__ dirs, const float* __restrict__ gt,
               const unsigned* __restrict__ perm, long long n, float* __restrict__ o_s, float* __restrict__ d_s,
               float* __restrict__ gt_s) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t src = perm[i];
    reinterpret_cast<F3*>(o_s)[i] = reinterpret_cast<const F3*>(origins)[src];
    reinterpret_cast<F3*>(d_s)[i] = reinterpret_cast<const F3*>(dirs)[src];
    if (gt != nullptr) reinterpret_cast<F3*>(gt_s)[i] = reinterpret_cast<const F3*>(gt)[src];
}

__global__ void __launch_bounds__(256)
k_order_scatter(const float* __restrict__ rgb_s, const unsigned* __restrict__ perm, long long n, float* __restrict__ rgb) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    reinterpret_cast<F3*>(rgb)[perm[i]] = reinterpret_cast<const F3*>(rgb_s)[i];
}

// How coherent is the batch as it stands?  Of a sample of consecutive pairs (q, q + 1), those that share an origin and
// point within a few pixels of each other: out[0] = such pairs, out[1] = pairs looked at, out[2] = pairs whose origins
// differ.  One block, no atomics, no
// buffer to clear.
// kProbeBlocks blocks of 256 threads, one pair a thread; every block leaves its three counts (the host adds them up).
// (Round 3 ran the 8 192 pairs through ONE block of 1 024 threads, eight pairs a thread: 32 k scattered line requests
// from a single CU -- 78 us a launch in profiles/r3_kernel_stats.csv, where the profiler's synchronisation made every
// call launch one; a loop that runs ahead of the device skips most of them, which is why the timed steps never showed it.)
__global__ void __launch_bounds__(256)
k_order_probe(const float* __restrict__ origins, const float* __restrict__ dirs, long long n, long long stride,
              int* __restrict__ out) {
    __shared__ int s_ok[4], s_seen[4], s_diff[4];
    int ok = 0, seen = 0, diff = 0;
    const long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * stride;
    if (i + 1 < n) {
        seen = 1;
        bool same = true;
        float dot = 0.f, na = 0.f, nb = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            same = same && (origins[3 * i + k] == origins[3 * (i + 1) + k]);
            const float a = dirs[3 * i + k], b = dirs[3 * (i + 1) + k];
            dot += a * b;
            na += a * a;
            nb += b * b;
        }
        // within 0.6 degrees (8 pixels of an 800-pixel, 40-degree view): cos^2 > 0.99989
        ok = (same && dot > 0.f && dot * dot > 0.99989f * na * nb) ? 1 : 0;
        diff = same ? 0 : 1;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        ok += __shfl_xor(ok, d);
        seen += __shfl_xor(seen, d);
        diff += __shfl_xor(diff, d);
    }
    if ((threadIdx.x & 63) == 0) {
        s_ok[threadIdx.x >> 6] = ok;
        s_seen[threadIdx.x >> 6] = seen;
        s_diff[threadIdx.x >> 6] = diff;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        out[3 * blockIdx.x + 0] = s_ok[0] + s_ok[1] + s_ok[2] + s_ok[3];
        out[3 * blockIdx.x + 1] = s_seen[0] + s_seen[1] + s_seen[2] + s_seen[3];
        out[3 * blockIdx.x + 2] = s_diff[0] + s_diff[1] + s_diff[2] + s_diff[3];
    }
}

struct Plan {
    size_t off_keys_in, off_keys_out, off_ids_in, off_sort, total;
    size_t sort_bytes;
};

Plan plan_for(long long n) {
    Plan p;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t nn = (size_t)(n > 0 ? n : 0);
    p.sort_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, p.sort_bytes, (const unsigned long long*)nullptr,
                                    (unsigned long long*)nullptr, (const unsigned*)nullptr, (unsigned*)nullptr,
                                    (size_t)(nn > 0 ? nn : 1), 0u, 50u);
    size_t off = 0;
    p.off_keys_in = off;
    off = up(off + nn * 8);
    p.off_keys_out = off;
    off = up(off + nn * 8);
    p.off_ids_in = off;
    off = up(off + nn * 4);
    p.off_sort = off;
    off = up(off + p.sort_bytes);
    p.total = off;
    return p;
}

}  // namespace

// ---- the ordering as a two-level sort of its own (order_strips) ----
// rocPRIM's radix sort is 21 launches for 640 k keys (143 us of the 197 us the ordering took).  The keys have structure:
// the high 18 bits name a strip (origin hash, v), a batch lights up a few hundred to a few thousand of the 262 144
// strips, and a strip holds a few hundred rays.  So: count the rays per strip (one atomic a ray; its return value is
// the ray's place in the strip), scan the counts (one block; it also lists the strips that hold anything), scatter
// (u, ray) pairs to their strips, and sort every strip by u in LDS (a counting sort over its 4096 u cells) -- that kernel
// writes the permutation AND the gathered rays / targets.  A memset and five
// launches.  (First form: one counter a strip and one block scanning 2^18 counters -- 0.11 and 0.34 ms: the counters
// were hit 700 times each by atomics that return a value, and the scan's loads did not coalesce.)
constexpr int kStripBits = 16;    // strips: 4 bits of the origin's hash, 12 bits of v
constexpr int kSubBits = 4;       // counters per strip: a strip's rays are counted in 16 cells (by the ray's number), because
                                  // atomics on ONE address that return a value retire at about 150 ns apiece -- 700 rays on one
                                  // counter were 0.11 ms
constexpr int kCells = 1 << (kStripBits + kSubBits);
constexpr int kScanBlocks = kCells / 1024;
constexpr int kStripMax = 2048;   // rays of one strip sorted together (a longer strip is sorted in pieces of this size)

// The key of the strip sort.  Where the ray's direction is given in the frame of a camera at the ray's origin that looks at
// the scene's centre with the world's z as its up vector -- n = -o / |o|, e1 = n x z, e2 = n x e1 --, as the gnomonic
// coordinates (d.e1 / d.n, d.e2 / d.n): image rows of such a camera (the orbit cameras of NeRF-synthetic, of bench.py) are
// lines of constant second coordinate, so a strip IS a pixel row (or two), and the ordered batch is the image-order
// batch; rows of any other camera are still straight lines of the chart, cut by the strips at an angle.  (The octahedral
// map of the first form bends the rows: the kernels that merge neighbouring rays' samples ran 0.207 / 0.128 ms on its
// strips against 0.166 / 0.110 on pixel rows.)  Rays that point away from the centre, or start at it, keep the
// octahedral map.
__device__ __forceinline__ unsigned strip_key(const float* __restrict__ origins, const float* __restrict__ dirs, long long i) {
    const float ox = origins[3 * i], oy = origins[3 * i + 1], oz = origins[3 * i + 2];
    const float dx = dirs[3 * i], dy = dirs[3 * i + 1], dz = dirs[3 * i + 2];
    unsigned h = __float_as_uint(ox) * 0x9E3779B1u;
    h = (h ^ __float_as_uint(oy)) * 0x85EBCA77u;
    h = (h ^ __float_as_uint(oz)) * 0xC2B2AE3Du;
    h ^= h >> 15;
    float u, v;   // both mapped to [0, 4096)
    const float on = sqrtf(ox * ox + oy * oy + oz * oz);
    const float nx = -ox / fmaxf(on, 1e-30f), ny = -oy / fmaxf(on, 1e-30f), nz = -oz / fmaxf(on, 1e-30f);
    // e1 = n x up, up = z (x where n is nearly vertical)
    const bool vertical = fabsf(nz) > 0.999f;
    float e1x = vertical ? 0.f : ny, e1y = vertical ? nz : -nx, e1z = vertical ? -ny : 0.f;
    const float e1n = rsqrtf(fmaxf(e1x * e1x + e1y * e1y + e1z * e1z, 1e-30f));
    e1x *= e1n;
    e1y *= e1n;
    e1z *= e1n;
    const float e2x = ny * e1z - nz * e1y, e2y = nz * e1x - nx * e1z, e2z = nx * e1y - ny * e1x;
    const float dn = dx * nx + dy * ny + dz * nz;
    const float dl = sqrtf(dx * dx + dy * dy + dz * dz);
    if (on > 1e-6f && dn > 0.2f * dl) {
        const float inv = 1.f / dn;
        u = ((dx * e1x + dy * e1y + dz * e1z) * inv + 2.f) * 1023.9f;
        v = ((dx * e2x + dy * e2y + dz * e2z) * inv + 2.f) * 1023.9f;
    } else {
        const float rn = 1.f / fmaxf(fabsf(dx) + fabsf(dy) + fabsf(dz), 1e-30f);
        float a = dx * rn, b = dy * rn;
        if (dz < 0.f) {
            const float fa = (1.f - fabsf(b)) * (a >= 0.f ? 1.f : -1.f), fb = (1.f - fabsf(a)) * (b >= 0.f ? 1.f : -1.f);
            a = fa;
            b = fb;
        }
        u = (a + 1.f) * 2047.75f;
        v = (b + 1.f) * 2047.75f;
    }
    const unsigned qu = (unsigned)fminf(fmaxf(u, 0.f), 4095.f);
    const unsigned qv = (unsigned)fminf(fmaxf(v, 0.f), 4095.f);
    return ((h & 15u) << 24) | (qv << 12) | qu;   // strip = key >> 12, u = key & 4095
}

__global__ void __launch_bounds__(256)
k_strip_count(const float* __restrict__ origins, const float* __restrict__ dirs, long long n, unsigned* __restrict__ keys,
              unsigned* __restrict__ rank, unsigned* __restrict__ counts) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned key = strip_key(origins, dirs, i);
    keys[i] = key;
    rank[i] = atomicAdd(&counts[((key >> 12) << kSubBits) | ((unsigned)i & ((1u << kSubBits) - 1u))], 1u);
}

// 1024 cells a block: the cells' exclusive prefix within the block (in place), the block's total, and the strips of the
// block (64 of them, 16 cells each) that hold rays, in order, with their number
__global__ void __launch_bounds__(1024)
k_strip_scan_local(unsigned* __restrict__ counts, unsigned* __restrict__ block_tot, unsigned* __restrict__ strips,
                   unsigned* __restrict__ block_nz) {
    __shared__ unsigned s_w[16], s_z[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const unsigned g = blockIdx.x * 1024u + t;
    const unsigned c = counts[g];
    unsigned incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    // the strip's total in its first lane
    unsigned st = c;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) st += __shfl_xor(st, d);
    const bool head = (lane & 15) == 0 && st > 0;
    const unsigned long long mh = __ballot(head);
    if (lane == 63) s_w[w] = incl;
    if (lane == 0) s_z[w] = (unsigned)__popcll(mh);
    __syncthreads();
    unsigned base = 0, zbase = 0;
    for (int k = 0; k < w; ++k) {
        base += s_w[k];
        zbase += s_z[k];
    }
    counts[g] = base + incl - c;
    if (head) strips[blockIdx.x * 64u + zbase + (unsigned)__popcll(mh & ((1ull << lane) - 1ull))] = g >> kSubBits;
    if (t == 1023) {
        block_tot[blockIdx.x] = base + incl;
        block_nz[blockIdx.x] = zbase + (unsigned)__popcll(mh);
    }
}

// exclusive scans of the blocks' totals and strip counts in place (kScanBlocks = 1024 values each); [kScanBlocks] = totals
__global__ void __launch_bounds__(1024)
k_strip_scan_base(unsigned* __restrict__ block_tot, unsigned* __restrict__ block_nz) {
    __shared__ unsigned s_a[1024], s_b[1024];
    const int t = threadIdx.x;
    const unsigned a = block_tot[t], b = block_nz[t];
    s_a[t] = a;
    s_b[t] = b;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned x = t >= d ? s_a[t - d] : 0u, y = t >= d ? s_b[t - d] : 0u;
        __syncthreads();
        s_a[t] += x;
        s_b[t] += y;
        __syncthreads();
    }
    block_tot[t] = s_a[t] - a;
    block_nz[t] = s_b[t] - b;
    if (t == 1023) {
        block_tot[1024] = s_a[t];
        block_nz[1024] = s_b[t];
    }
}

__device__ __forceinline__ unsigned cell_offset(const unsigned* __restrict__ counts, const unsigned* __restrict__ block_tot,
                                                unsigned cell) {
    return cell < (unsigned)kCells ? counts[cell] + block_tot[cell >> 10] : block_tot[1024];
}

__global__ void __launch_bounds__(256)
k_strip_scatter(const unsigned* __restrict__ keys, const unsigned* __restrict__ rank, const unsigned* __restrict__ counts,
                const unsigned* __restrict__ block_tot, long long n, unsigned long long* __restrict__ pairs) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned key = keys[i];
    const unsigned cell = ((key >> 12) << kSubBits) | ((unsigned)i & ((1u << kSubBits) - 1u));
    pairs[cell_offset(counts, block_tot, cell) + rank[i]] = ((unsigned long long)(key & 4095u) << 32) | (unsigned)i;
}

// one strip a block at a time: its (u, ray) pairs sorted by u in LDS -- a counting sort over the 4096 u cells (count, scan,
// place: five barriers; the bitonic network of the first form took 55 for a strip of 800 rays and 49 us of the ordering's
// 110) --, then the permutation and the rays in the new order.  Rays of one u cell (a pixel, or the same pixel of two rows)
// come out in the order of the LDS atomics.
__global__ void __launch_bounds__(256)
k_strip_sort_gather(const unsigned* __restrict__ counts, const unsigned* __restrict__ block_tot,
                    const unsigned* __restrict__ strips, const unsigned* __restrict__ block_nz,
                    const unsigned long long* __restrict__ pairs, const float* __restrict__ origins,
                    const float* __restrict__ dirs, const float* __restrict__ gt, unsigned* __restrict__ perm,
                    float* __restrict__ o_s, float* __restrict__ d_s, float* __restrict__ gt_s) {
    __shared__ unsigned s_hist[4096];            // rays per u cell, then the cell's first place
    __shared__ unsigned s_out[kStripMax];        // ray numbers in the new order
    __shared__ unsigned s_part[256];
    __shared__ unsigned s_nz[kScanBlocks + 1];
    const int t = threadIdx.x;
    for (int i = t; i <= kScanBlocks; i += 256) s_nz[i] = block_nz[i];
    __syncthreads();
    const unsigned ns = s_nz[kScanBlocks];
    for (unsigned si = blockIdx.x; si < ns; si += gridDim.x) {
        // the si-th strip that holds rays: which scan block listed it (s_nz: exclusive prefix), and where
        int b = 0;
        for (int step = kScanBlocks / 2; step >= 1; step >>= 1)
            if (s_nz[b + step] <= si) b += step;
        const unsigned strip = strips[(unsigned)b * 64u + (si - s_nz[b])];
        const unsigned begin = cell_offset(counts, block_tot, strip << kSubBits);
        const unsigned end = cell_offset(counts, block_tot, (strip + 1u) << kSubBits);
        for (unsigned p0 = begin; p0 < end; p0 += kStripMax) {
            const int m = (int)min((unsigned)kStripMax, end - p0);
            for (int i = t; i < 4096; i += 256) s_hist[i] = 0u;
            __syncthreads();
            unsigned long long mine[kStripMax / 256];
            unsigned rank[kStripMax / 256];
#pragma unroll
            for (int r = 0; r < kStripMax / 256; ++r) {
                const int i = t + 256 * r;
                mine[r] = i < m ? pairs[p0 + i] : 0ull;
                rank[r] = i < m ? atomicAdd(&s_hist[(unsigned)(mine[r] >> 32)], 1u) : 0u;
            }
            __syncthreads();
            // exclusive scan of the 4096 counts: 16 cells a thread, the threads' sums scanned across the block
            unsigned loc[16], sum = 0u;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                loc[c] = s_hist[t * 16 + c];
                sum += loc[c];
            }
            s_part[t] = sum;
            __syncthreads();
            for (int d = 1; d < 256; d <<= 1) {
                const unsigned v = t >= d ? s_part[t - d] : 0u;
                __syncthreads();
                s_part[t] += v;
                __syncthreads();
            }
            unsigned run = s_part[t] - sum;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                s_hist[t * 16 + c] = run;
                run += loc[c];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < kStripMax / 256; ++r) {
                const int i = t + 256 * r;
                if (i < m) s_out[s_hist[(unsigned)(mine[r] >> 32)] + rank[r]] = (unsigned)(mine[r] & 0xffffffffull);
            }
            __syncthreads();
            for (int i = t; i < m; i += 256) {
                const unsigned src = s_out[i];
                const size_t dst = (size_t)p0 + i;
                perm[dst] = src;
                reinterpret_cast<F3*>(o_s)[dst] = reinterpret_cast<const F3*>(origins)[src];
                reinterpret_cast<F3*>(d_s)[dst] = reinterpret_cast<const F3*>(dirs)[src];
                if (gt != nullptr) reinterpret_cast<F3*>(gt_s)[dst] = reinterpret_cast<const F3*>(gt)[src];
            }
            __syncthreads();
        }
    }
}

// ---- internal interface for asurf_api.hip (asurf_order.h) ----
namespace asurf_order {

size_t tmp_bytes32(long long n) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    size_t sb = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sb, (const unsigned*)nullptr, (unsigned*)nullptr, (const unsigned*)nullptr,
                                    (unsigned*)nullptr, nn, 0u, 32u);
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    return up(nn * 4) * 3 + up(sb);   // keys in, keys out, ids in, sort scratch
}

int order32(const float* origins, const float* dirs, long long n, unsigned* perm, void* tmp, size_t tmp_bytes,
            hipStream_t s, int key_bits) {
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    if (n <= 0) return 0;
    if (tmp_bytes < tmp_bytes32(n)) return -1;
    char* base = static_cast<char*>(tmp);
    const size_t plane = up((size_t)n * 4);
    auto* keys_in = reinterpret_cast<unsigned*>(base);
    auto* keys_out = reinterpret_cast<unsigned*>(base + plane);
    auto* ids_in = reinterpret_cast<unsigned*>(base + 2 * plane);
    hipLaunchKernelGGL(k_order_keys32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, origins, dirs, n, keys_in, ids_in);
    size_t sb = tmp_bytes - 3 * plane;
    const hipError_t e = rocprim::radix_sort_pairs(base + 3 * plane, sb, keys_in, keys_out, ids_in, perm, (size_t)n, 0u,
                                                   (unsigned)key_bits, s);
    return (e == hipSuccess && hipGetLastError() == hipSuccess) ? 0 : -3;
}

size_t tmp_bytes_strips(long long n) {
    const size_t nn = (size_t)(n > 0 ? n : 1);
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    // keys, ranks | cell counts / offsets | the blocks' totals and strip counts (+1) | the strips listed per block | pairs
    return up(nn * 4) * 2 + up((size_t)kCells * 4) + 2 * up((kScanBlocks + 1) * 4) + up((size_t)kScanBlocks * 64 * 4) + up(nn * 8);
}

int order_strips(const float* origins, const float* dirs, const float* gt, long long n, unsigned* perm, float* o_s,
                 float* d_s, float* gt_s, void* tmp, size_t tmp_bytes, hipStream_t s) {
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    if (n <= 0) return 0;
    if (tmp_bytes < tmp_bytes_strips(n)) return -1;
    char* base = static_cast<char*>(tmp);
    const size_t plane = up((size_t)n * 4), cells = up((size_t)kCells * 4), small = up((kScanBlocks + 1) * 4);
    auto* keys = reinterpret_cast<unsigned*>(base);
    auto* rank = reinterpret_cast<unsigned*>(base + plane);
    auto* counts = reinterpret_cast<unsigned*>(base + 2 * plane);
    auto* block_tot = reinterpret_cast<unsigned*>(base + 2 * plane + cells);
    auto* block_nz = reinterpret_cast<unsigned*>(base + 2 * plane + cells + small);
    auto* strips = reinterpret_cast<unsigned*>(base + 2 * plane + cells + 2 * small);
    auto* pairs = reinterpret_cast<unsigned long long*>(base + 2 * plane + cells + 2 * small + up((size_t)kScanBlocks * 64 * 4));
    if (hipMemsetAsync(counts, 0, (size_t)kCells * 4, s) != hipSuccess) return -3;
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_strip_count, dim3(nb), dim3(256), 0, s, origins, dirs, n, keys, rank, counts);
    hipLaunchKernelGGL(k_strip_scan_local, dim3(kScanBlocks), dim3(1024), 0, s, counts, block_tot, strips, block_nz);
    hipLaunchKernelGGL(k_strip_scan_base, dim3(1), dim3(1024), 0, s, block_tot, block_nz);
    hipLaunchKernelGGL(k_strip_scatter, dim3(nb), dim3(256), 0, s, keys, rank, counts, block_tot, n, pairs);
    hipLaunchKernelGGL(k_strip_sort_gather, dim3(2048), dim3(256), 0, s, counts, block_tot, strips, block_nz, pairs, origins,
                       dirs, gt, perm, o_s, d_s, gt_s);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

void gather(const float* origins, const float* dirs, const float* gt, const unsigned* perm, long long n, float* o_s,
            float* d_s, float* gt_s, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_order_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, origins, dirs, gt, perm, n,
                       o_s, d_s, gt_s);
}

void scatter_rgb(const float* rgb_s, const unsigned* perm, long long n, float* rgb, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_order_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, rgb_s, perm, n, rgb);
}

// Tile order of an image batch (asurf_fused_params::image_width): the n = H x W rays of a row-major image go to where
// their 8 x 8 pixel tile comes, tiles row by row, pixels row by row inside a tile (edge tiles are narrower / lower).  The
// 64 samples a wave of the backward kernels takes then fall into a PATCH of the surface instead of a line across it --
// 1.07 distinct gradient rows a sample instead of 2.13 (bench scene) -- and the 16 rays of a wave of k_sweep are 2 x 8
// pixels.  One thread a source pixel (coalesced reads; the writes are runs of 8 pixels).
__global__ void __launch_bounds__(256)
k_tile_scatter(const float* __restrict__ origins, const float* __restrict__ dirs, const float* __restrict__ gt, long long n,
               int W, unsigned* __restrict__ perm, float* __restrict__ o_s, float* __restrict__ d_s, float* __restrict__ gt_s) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int H = (int)(n / W);
    const int y = (int)(i / W), x = (int)(i - (long long)y * W);
    const int ty = y >> 3, tx = x >> 3;
    const int th = min(8, H - 8 * ty), tw = min(8, W - 8 * tx);
    const long long dst = (long long)8 * ty * W + (long long)8 * tx * th + (y & 7) * tw + (x & 7);
    perm[dst] = (unsigned)i;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o_s[3 * dst + k] = origins[3 * i + k];
        d_s[3 * dst + k] = dirs[3 * i + k];
        gt_s[3 * dst + k] = gt[3 * i + k];
    }
}

void tile_order(const float* origins, const float* dirs, const float* gt, long long n, int width, unsigned* perm, float* o_s,
                float* d_s, float* gt_s, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_tile_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, origins, dirs, gt, n, width, perm,
                       o_s, d_s, gt_s);
}

void probe(const float* origins, const float* dirs, long long n, int* out, hipStream_t s) {
    if (n < 2) return;
    const long long pairs = (long long)kProbeBlocks * 256, stride = (n / pairs) > 1 ? (n / pairs) : 1;
    hipLaunchKernelGGL(k_order_probe, dim3(kProbeBlocks), dim3(256), 0, s, origins, dirs, n, stride, out);
}

}  // namespace asurf_order

extern "C" {

size_t asurf_ray_order_tmp_bytes(int64_t n_rays) { return plan_for(n_rays).total; }

int asurf_ray_order(const float* origins, const float* dirs, int64_t n_rays, int32_t* perm_out, void* tmp,
                    size_t tmp_bytes, void* stream) {
    if (n_rays < 0 || n_rays >= (1ll << 31)) return -1;
    if (n_rays == 0) return 0;
    if (!origins || !dirs || !perm_out || !tmp) return -1;
    const Plan p = plan_for(n_rays);
    if (tmp_bytes < p.total) return -1;
    char* base = static_cast<char*>(tmp);
    auto* keys_in = reinterpret_cast<unsigned long long*>(base + p.off_keys_in);
    auto* keys_out = reinterpret_cast<unsigned long long*>(base + p.off_keys_out);
    auto* ids_in = reinterpret_cast<unsigned*>(base + p.off_ids_in);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(k_order_keys, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, s, origins, dirs,
                       (long long)n_rays, keys_in, ids_in);
    size_t sb = p.sort_bytes;
    const hipError_t e = rocprim::radix_sort_pairs(base + p.off_sort, sb, keys_in, keys_out, ids_in,
                                                   reinterpret_cast<unsigned*>(perm_out), (size_t)n_rays, 0u, 50u, s);
    return (e == hipSuccess && hipGetLastError() == hipSuccess) ? 0 : -3;
}

}  // extern "C"
