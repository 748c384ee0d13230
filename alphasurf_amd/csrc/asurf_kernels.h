// asurf_kernels.h -- HIP kernels of the surf_trav hot path for gfx950 (wave64).
//
// Design (DESIGN.md has the long form):
//   * k_build_masks  -- per call, streams the dense `links` volume once and emits an
//     occupancy pyramid: one bit per voxel ("this voxel can produce a sample") packed
//     as 8 x uint64 per 8^3 brick (64 B = one cache line per brick) + one bit per brick.
//     This replaces the reference's 8 dependent link loads per traversed voxel
//     (surf_trav.cu:212-239) by register-cached bit tests; the DDA itself is unchanged.
//   * k_forward / k_backward -- ONE RAY PER LANE.  Each wave runs a while-while loop:
//     (A) every lane advances its own DDA until it reaches a work voxel, (B) lanes that
//     found one form the cubic and solve it (f64, thread-per-ray, dense), (C) lanes with
//     a valid sample are served one after another by the whole wave: the 8 x D spherical
//     harmonic coefficients are read (and, backward, atomically accumulated) with the
//     lanes spread over (corner, coefficient), i.e. contiguous 4*D-byte row segments.
//     The reference instead runs the whole traversal redundantly on 3*basis_dim lanes of
//     a 32-wide warp (surf_trav.cu:3139-3157).
#pragma once
#include "asurf_device.h"

// minimum waves per SIMD the ray kernels are compiled for (__launch_bounds__ 2nd argument):
// caps the VGPR allocation at 512 / ASURF_MIN_WAVES
#ifndef ASURF_MIN_WAVES
#define ASURF_MIN_WAVES 1
#endif

namespace asurf {

// ------------------------------------------------------------------------------------
// Occupancy pyramid
// ------------------------------------------------------------------------------------
// grid: nb[0] * nb[1] * ceil(nb[2] / 8) blocks of 512 threads; block = 8 bricks in a row along z.  Wave w is the
// row y = 8 by + w of the bricks, its 64 lanes are 64 consecutive z (one 256-byte request per wave and link
// plane), and the block walks the 8 x-slabs: 8 iterations, no barrier between them.  A ballot is then a run of
// 64 z for one (x, y); byte j of it belongs to brick j of the row.  The brick words are put together from
// those bytes: word x (bit 8 y + z) across the waves in LDS, word y (bit 8 z + x) inside the wave over the
// iterations, word z (bit 8 x + y) per lane over the iterations and across the waves in LDS at the end.
// (The first version gave a thread one voxel of ONE brick and walked 32 bricks with two barriers each: a chain of
// 32 dependent round trips of 32-byte pieces, 0.07 ms at 256^3.)  The coarse plane is written as bytes: a
// block's 8 bricks are one byte of its little-endian 32-bit words.
__device__ __forceinline__ unsigned long long spread_bits_to_bytes(unsigned b) {  // bit i of a byte -> bit 8 i
    const unsigned long long x = ((unsigned long long)b * 0x0101010101010101ull) & 0x8040201008040201ull;
    const unsigned long long lo7 = 0x7f7f7f7f7f7f7f7full;
    return ((((x & lo7) + lo7) | x) >> 7) & 0x0101010101010101ull;
}

struct MaskTables {  // per block
    unsigned long long xw[64], xl[64], z[64];  // [x-slab][brick], [x-slab][brick], [brick][lz]
    int alive;
};

// one row of 8 bricks along z (row = (bx * nb1 + by) * nzg + zg); returns whether this wave saw a vertex in it
// the end of a row: the block's tables go out as the bricks' words (x, y and z cuts) and the coarse bytes
__device__ __forceinline__ void build_masks_row_finish(const KGrid& g, unsigned long long* fine_work,
                                                       unsigned long long* fine_live, unsigned int* coarse_work,
                                                       unsigned int* coarse_live, unsigned long long* fine_y,
                                                       unsigned long long* fine_z, int bx, int by, int bxy, int zg,
                                                       unsigned long long acc_y, unsigned long long acc_z,
                                                       MaskTables& S) {
    unsigned long long* s_xw = S.xw;
    unsigned long long* s_xl = S.xl;
    unsigned long long* s_z = S.z;
    const int t = threadIdx.x;
    const int ly = t >> 6, l = t & 63;
    if (fine_y && acc_z) atomicOr(&s_z[l], acc_z);
    __syncthreads();
    const size_t brick0 = ((size_t)bx * g.nb[1] + by) * g.nb[2] + (size_t)zg * 8;
    const int nbr = min(8, g.nb[2] - zg * 8);  // bricks of the row that exist
    if (fine_y && l < nbr) fine_y[(brick0 + l) * 8 + ly] = acc_y;
    if (t < 64) {
        const int lx = t >> 3, j = t & 7;
        if (j < nbr) {
            fine_work[(brick0 + j) * 8 + lx] = s_xw[t];
            fine_live[(brick0 + j) * 8 + lx] = s_xl[t];
        }
        const int j2 = t >> 3, lz = t & 7;
        if (fine_y && j2 < nbr) fine_z[(brick0 + j2) * 8 + lz] = s_z[t];
    }
    if (t == 0) {
        unsigned cw = 0, cl = 0;
        for (int j = 0; j < 8; ++j) {
            unsigned long long aw = 0ull, al = 0ull;
            for (int lx = 0; lx < 8; ++lx) {
                aw |= s_xw[lx * 8 + j];
                al |= s_xl[lx * 8 + j];
            }
            cw |= (aw != 0ull ? 1u : 0u) << j;
            cl |= (al != 0ull ? 1u : 0u) << j;
        }
        reinterpret_cast<unsigned char*>(coarse_work)[(size_t)bxy * g.nwz * 4 + zg] = (unsigned char)cw;
        reinterpret_cast<unsigned char*>(coarse_live)[(size_t)bxy * g.nwz * 4 + zg] = (unsigned char)cl;
    }
}

__device__ __forceinline__ bool build_masks_row(const KGrid& g, float sigma_thresh, int need_surf,
                                                unsigned long long* fine_work, unsigned long long* fine_live,
                                                unsigned int* coarse_work, unsigned int* coarse_live,
                                                unsigned long long* fine_y, unsigned long long* fine_z, int orient_live,
                                                int nzg, unsigned row, MaskTables& S) {
    unsigned long long* s_xw = S.xw;
    unsigned long long* s_xl = S.xl;
    unsigned long long* s_z = S.z;
    const int t = threadIdx.x;
    const int ly = t >> 6, l = t & 63;
    const int zg = (int)(row % (unsigned)nzg);
    const int bxy = (int)(row / (unsigned)nzg);
    const int by = bxy % g.nb[1];
    const int bx = bxy / g.nb[1];
    if (t < 64) s_xw[t] = s_xl[t] = s_z[t] = 0ull;
    __syncthreads();
    const int y = by * 8 + ly, z = zg * 64 + l;
    const bool inside_yz = y + 1 < g.size[1] && z + 1 < g.size[2];
    unsigned long long acc_y = 0ull, acc_z = 0ull;
    bool alive = false;  // some run of this wave holds a vertex (a function of `links` alone)
    // The kernel streams `links` and is as fast as the bytes it keeps in flight (one 256-byte request per wave
    // would be 2 MB on the whole chip: 1 TB/s at the 2 us such a request takes): all eight slabs' first links
    // are asked for before the first one is looked at.
    int32_t l0s[8];
#pragma unroll
    for (int lx = 0; lx < 8; ++lx) {
        const int x = bx * 8 + lx;
        l0s[lx] = (inside_yz && x + 1 < g.size[0])
                      ? g.links[(size_t)x * g.offx + (size_t)y * g.offy + z] : -1;
    }
#pragma unroll
    for (int lx = 0; lx < 8; ++lx) {
        const int x = bx * 8 + lx;
        bool live = false, work = false;
        const bool inside = inside_yz && x + 1 < g.size[0];
        const int32_t* p = g.links + ((size_t)x * g.offx + (size_t)y * g.offy + z);
        const int32_t l0 = l0s[lx];
        // most of a sparse grid is empty space: when no cell of the run has its own vertex (and the surf_trav
        // planes need all eight), the other seven loads are skipped for the whole wave
        const bool run_dead = need_surf != 2 && __ballot(l0 >= 0) == 0ull;
        alive = alive || !run_dead;
        if (inside && !run_dead) {
            const int32_t l1 = p[1], l2 = p[g.offy], l3 = p[g.offy + 1];
            const int32_t l4 = p[g.offx], l5 = p[g.offx + 1], l6 = p[g.offx + g.offy],
                          l7 = p[g.offx + g.offy + 1];
            if (need_surf == 2) {
                // cuvol cells (render_lerp_kernel_cuvol.cu:66-77): a missing corner reads as 0 and
                // the sample counts iff the trilinear density exceeds sigma_thresh -- impossible
                // unless some corner value does (a convex combination)
                const int32_t lk[8] = {l0, l1, l2, l3, l4, l5, l6, l7};
                float mx = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) mx = fmaxf(mx, lk[i] >= 0 ? g.density[lk[i]] : 0.f);
                live = work = mx > sigma_thresh;
            } else if ((l0 | l1 | l2 | l3 | l4 | l5 | l6 | l7) >= 0) {
                const int32_t lk[8] = {l0, l1, l2, l3, l4, l5, l6, l7};
                bool all_below = true;  // surf_trav.cu:230-239
#pragma unroll
                for (int i = 0; i < 8; ++i) all_below = all_below && (g.density[lk[i]] < sigma_thresh);
                if (!all_below) {
                    live = true;
                    work = true;
                    if (need_surf) {  // surf_trav.cu:265-277: some level set within [min,max]
                        float smin = g.surface[lk[0]], smax = smin;
#pragma unroll
                        for (int i = 1; i < 8; ++i) {
                            const float sv = g.surface[lk[i]];
                            smin = fminf(smin, sv);
                            smax = fmaxf(smax, sv);
                        }
                        work = false;
                        for (int li = 0; li < g.level_set_num; ++li) {
                            const float lv = g.level_set[li];
                            if (!(lv < smin || lv > smax)) work = true;
                        }
                    }
                }
            }
        }
        const unsigned long long bw = __ballot(work), bl = __ballot(live);
        if ((bw | bl) == 0ull) continue;  // (wave-uniform)
        if (l < 8) {  // lane j: brick j of the row
            const unsigned byte_w = (unsigned)(bw >> (8 * l)) & 0xffu, byte_l = (unsigned)(bl >> (8 * l)) & 0xffu;
            if (byte_w) atomicOr(&s_xw[lx * 8 + l], (unsigned long long)byte_w << (8 * ly));
            if (byte_l) atomicOr(&s_xl[lx * 8 + l], (unsigned long long)byte_l << (8 * ly));
            acc_y |= spread_bits_to_bytes(orient_live ? byte_l : byte_w) << lx;
        }
        if (orient_live ? live : work) acc_z |= 1ull << (lx * 8 + ly);
    }
    build_masks_row_finish(g, fine_work, fine_live, coarse_work, coarse_live, fine_y, fine_z, bx, by, bxy, zg, acc_y, acc_z,
                           S);
    return alive;
}

// MaskHdr: what a workspace remembers about `links` between calls (asurf_grid::links_epoch, include/asurf.h): the
// rows of bricks in which some 64-run of vertices along z holds a link >= 0.  Every other row is empty space whatever
// the densities and surfaces: its words in all planes are zero and stay zero.
constexpr int kSumSlots = 64;
struct MaskHdr {
    unsigned long long magic, epoch;
    int n_live, pad;
    // links-only planes (asurf_masks2.h, k_links_planes): the checksum of `links` they were built under, the blocks'
    // ticket, and the slots this call's checksum is added up in (zero between calls)
    unsigned long long sum[2];
    unsigned int ticket;
    int sum_known;
    unsigned long long acc[2][kSumSlots][2];   // [parity]: a call adds into one set while it clears the other for the next call
};
constexpr unsigned long long kMaskMagic = 0x61737572666d736bull;

__global__ void __launch_bounds__(512)
k_build_masks(KGrid g, float sigma_thresh, int need_surf, unsigned long long* fine_work,
              unsigned long long* fine_live, unsigned int* coarse_work,
              unsigned int* coarse_live, unsigned long long* fine_y, unsigned long long* fine_z,
              int orient_live, int nzg, MaskHdr* hdr, int* live_rows) {
    const TlScope tl(1);
    __shared__ MaskTables S;
    if (threadIdx.x == 0) S.alive = 0;
    const bool alive = build_masks_row(g, sigma_thresh, need_surf, fine_work, fine_live, coarse_work, coarse_live, fine_y,
                                       fine_z, orient_live, nzg, blockIdx.x, S);
    if (hdr != nullptr) {  // (wave-uniform) leave the list of rows that are not empty space
        if (alive && (threadIdx.x & 63) == 0) S.alive = 1;
        __syncthreads();
        if (threadIdx.x == 0 && S.alive) live_rows[atomicAdd(&hdr->n_live, 1)] = (int)blockIdx.x;
    }
    tl.end();
}

__global__ void k_mask_seal(MaskHdr* hdr, unsigned long long epoch) {
    hdr->magic = kMaskMagic;
    hdr->epoch = epoch;
}

// the list was built under the checksum of `links` this call has added up in acc[parity] (one wave)
__global__ void __launch_bounds__(64)
k_mask_seal_sum(MaskHdr* hdr, int parity) {
    unsigned long long a = threadIdx.x < kSumSlots ? hdr->acc[parity][threadIdx.x][0] : 0ull;
    unsigned long long b = threadIdx.x < kSumSlots ? hdr->acc[parity][threadIdx.x][1] : 0ull;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        a += __shfl_xor(a, m);
        b += __shfl_xor(b, m);
    }
    if (threadIdx.x == 0) {
        hdr->sum[0] = a;
        hdr->sum[1] = b;
        hdr->sum_known = 1;
        hdr->epoch = 0ull;
        hdr->magic = kMaskMagic;
    }
}

// The same pass over the rows of the list only.  (Its waves are all resident from the start and each is a chain of
// 24 dependent round trips; with the loads of four x-slabs in flight together the chain is 6 long and the registers
// halve the resident waves: 0.055 ms either way at 256^3, a third of the rows live.)
__global__ void __launch_bounds__(512)
k_build_masks_live(KGrid g, float sigma_thresh, int need_surf, unsigned long long* fine_work,
                   unsigned long long* fine_live, unsigned int* coarse_work, unsigned int* coarse_live,
                   unsigned long long* fine_y, unsigned long long* fine_z, int orient_live, int nzg,
                   const MaskHdr* hdr, const int* live_rows, unsigned long long epoch, int n_rows, int* zero_ptr,
                   int zero_n) {
    const TlScope tl(1);
    __shared__ MaskTables S;
    // (the pipeline's counters, zeroed here for the kernels that follow: a memset node of their own is 4 us.  Not in
    // k_build_masks: the same five stores in its block 0 cost that kernel 9 us -- 0.067 -> 0.076 ms, measured both ways)
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < zero_n; i += blockDim.x) zero_ptr[i] = 0;
    // Not the header the host table promised (a full pass captured for another grid has been replayed over this
    // workspace: it cleared the header and rewrote every plane): the list means nothing, and the rows that were empty
    // space may hold the other grid's bits -- every row is rebuilt.
    const bool listed = hdr->magic == kMaskMagic && hdr->epoch == epoch;
    const int n = listed ? min(hdr->n_live, n_rows) : n_rows;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int row = listed ? live_rows[i] : i;
        if (row >= 0 && row < n_rows)
            build_masks_row(g, sigma_thresh, need_surf, fine_work, fine_live, coarse_work, coarse_live, fine_y, fine_z,
                            orient_live, nzg, (unsigned)row, S);
        __syncthreads();  // (the next row zeroes the tables)
    }
    tl.end();
}

// ------------------------------------------------------------------------------------
// Per-lane voxel walk (the DDA of surf_trav.cu:67-79, 86-197)
//
// The reference recomputes all three far-plane crossing times of the current voxel every
// step.  The two on the axes that were not stepped are the same expression on the same
// operands as in the previous step, so only the stepped axis is re-divided here; every
// value, comparison and tie-break is bit-identical to the reference's.
// ------------------------------------------------------------------------------------
struct Walk {
    int nx, ny, nz;      // next voxel
    int vx, vy, vz;      // current voxel
    float tf0, tf1, tf2; // far-plane crossing times of the NEXT voxel
    int sx, sy, sz;      // step direction per axis (+1 / -1)
    float t, t_far;
    // per-ray constants as scalars: selecting between array elements of Ray / KGrid makes the
    // compiler index them in scratch / kernarg memory inside the DDA loop
    float ox, oy, oz, dx, dy, dz, t_leave;
    int lim0, lim1, lim2;
    int noskip_key;      // brick whose jump attempt was refused (walk_hier)
    float rdx, rdy, rdz; // correctly rounded reciprocals of the direction (walk_hier's quotients)
    bool fast_div;       // ... usable: no component is zero, subnormal or so small that x * (1/d) overflows
    bool straight;       // no direction component is ~0 (brick jumps allowed)
    // register cache of the occupancy pyramid
    int cur_key;         // packed brick coordinates
    int cur_brick;       // linear brick index
    unsigned cur_flag;
    int cur_lx;
    unsigned long long cur_word;
};

__device__ __forceinline__ float plane_time(int n, int s, float o, float d) {
    // ((float)far_plane - origin) / dir with far_plane = n + (dir > 0)
    return ((float)(n + ((s + 1) >> 1)) - o) / d;
}

// x / d from the correctly rounded reciprocal rd = RN(1 / d): q0 = RN(x rd) is within an ulp of x / d, the
// residual x - q0 d comes out of one FMA exactly, and RN(q0 + residual * rd) is the correctly rounded quotient
// (Markstein's correction step; the IEEE sequence the compiler emits for `/` is ten instructions).  Checked
// against IEEE division on 2e8 operand pairs of the walk's ranges and on all-ones / power-of-two mantissas.
// Not for d = 0, subnormal d or quotients near overflow: Walk::fast_div.
__device__ __forceinline__ float quotient(float x, float d, float rd) {
    const float q0 = x * rd;
    const float e = fmaf(-q0, d, x);
    return fmaf(e, rd, q0);
}

__device__ __forceinline__ void walk_init(const KGrid& g, const Ray& r, Walk& w) {
    w.t = r.tmin;
    w.ox = r.o[0];
    w.oy = r.o[1];
    w.oz = r.o[2];
    w.dx = r.d[0];
    w.dy = r.d[1];
    w.dz = r.d[2];
    w.t_leave = r.tmax + 1.f;
    w.lim0 = g.size[0] - 1;
    w.lim1 = g.size[1] - 1;
    w.lim2 = g.size[2] - 1;
    w.nx = min(max((int)fmaf(w.t, w.dx, w.ox), 0), w.lim0 - 1);
    w.ny = min(max((int)fmaf(w.t, w.dy, w.oy), 0), w.lim1 - 1);
    w.nz = min(max((int)fmaf(w.t, w.dz, w.oz), 0), w.lim2 - 1);
    w.sx = (w.dx > 0.f) ? 1 : -1;
    w.sy = (w.dy > 0.f) ? 1 : -1;
    w.sz = (w.dz > 0.f) ? 1 : -1;
    w.tf0 = plane_time(w.nx, w.sx, w.ox, w.dx);
    w.tf1 = plane_time(w.ny, w.sy, w.oy, w.dy);
    w.tf2 = plane_time(w.nz, w.sz, w.oz, w.dz);
    w.vx = w.vy = w.vz = 0;
    w.t_far = 0.f;
    w.cur_key = -1;
    w.noskip_key = -1;
    w.straight = fminf(fminf(fabsf(w.dx), fabsf(w.dy)), fabsf(w.dz)) > 1e-5f;
    w.rdx = 1.f / w.dx;
    w.rdy = 1.f / w.dy;
    w.rdz = 1.f / w.dz;
    w.fast_div = fmaxf(fmaxf(fabsf(w.rdx), fabsf(w.rdy)), fabsf(w.rdz)) < 1e30f;  // false for inf and NaN too
    w.cur_brick = 0;
    w.cur_flag = 0;
    w.cur_lx = -1;
    w.cur_word = 0;
}

// one DDA step: enters (nx,ny,nz), takes its far crossing, picks the next voxel
__device__ __forceinline__ void walk_step(const KGrid& g, const Ray& r, Walk& w) {
    w.vx = w.nx;
    w.vy = w.ny;
    w.vz = w.nz;
    const float t_far = fminf(fminf(w.tf0, w.tf1), w.tf2);
    w.t_far = t_far;
    const bool a0 = (t_far == w.tf0);
    const bool a1 = !a0 && (t_far == w.tf1);
    // read every member unconditionally first: a load inside a ?: arm is merged by the
    // optimiser into one load from a selected ADDRESS, which pins the struct in scratch
    const int nx = w.nx, ny = w.ny, nz = w.nz, sx = w.sx, sy = w.sy, sz = w.sz;
    const int l0 = w.lim0, l1 = w.lim1, l2 = w.lim2;
    const float ox = w.ox, oy = w.oy, oz = w.oz, dx = w.dx, dy = w.dy, dz = w.dz;
    const int n_new = a0 ? nx + sx : (a1 ? ny + sy : nz + sz);
    const int lim = a0 ? l0 : (a1 ? l1 : l2);
    const int s = a0 ? sx : (a1 ? sy : sz);
    const float o = a0 ? ox : (a1 ? oy : oz);
    const float d = a0 ? dx : (a1 ? dy : dz);
    const bool leaves = (n_new < 0) || (n_new >= lim);
    w.t = leaves ? w.t_leave : t_far;
    const float q = plane_time(n_new, s, o, d);
    if (a0) {
        w.nx = n_new;
        w.tf0 = q;
    } else if (a1) {
        w.ny = n_new;
        w.tf1 = q;
    } else {
        w.nz = n_new;
        w.tf2 = q;
    }
}

// work bit of the current voxel; brick flag and the 64-voxel slab word are kept in registers.
// LDS: the per-brick flags were staged in `lds_coarse` (address space resolved at compile time).
template <bool LDS>
__device__ __forceinline__ bool walk_test(const KGrid& g, const unsigned int* lds_coarse, Walk& w) {
    const int key = ((w.vx >> 3) << 20) | ((w.vy >> 3) << 10) | (w.vz >> 3);
    if (key != w.cur_key) {
        w.cur_key = key;
        const int bz = w.vz >> 3;
        const int bxy = (w.vx >> 3) * g.nb[1] + (w.vy >> 3);
        w.cur_brick = bxy * g.nb[2] + bz;
        const int wi = bxy * g.nwz + (bz >> 5);
        unsigned word;
        if constexpr (LDS) word = lds_coarse[wi];
        else word = g.coarse[wi];
        w.cur_flag = (word >> (bz & 31)) & 1u;
        w.cur_lx = -1;
    }
    if (!w.cur_flag) return false;
    const int lx = w.vx & 7;
    if (lx != w.cur_lx) {
        w.cur_lx = lx;
        w.cur_word = g.fine[(size_t)w.cur_brick * 8 + lx];
    }
    return (w.cur_word >> (((w.vy & 7) << 3) | (w.vz & 7))) & 1ull;
}

// Hierarchical walk step, exact.  One body serves two cell sizes so that the lanes of a wave do
// not diverge: the cell is the voxel about to be entered, or -- when its 8^3 brick holds no
// flagged voxel -- the whole brick.  The walk processes plane crossings in (time, axis) order and
// every crossing time is the same float expression plane_time(index) however it is reached, so:
//   * the cell is left through the axis with the smallest exit time T (ties: lowest axis, as in
//     walk_step); for a voxel cell these are exactly walk_step's far-plane times;
//   * for a brick cell the voxel index on the other two axes at T is floor(o + T d), PROVIDED
//     that position is not within 1/256 of a voxel plane: the float error of o + T d and of the
//     plane times is < 1e-3 voxel at grid sizes <= 1024, so which crossings precede T is decided
//     without ambiguity.  Otherwise, and near the end of the ray (where the backward's step_size
//     quirk looks at individual voxels), the brick is walked voxel by voxel.
// Returns 0: no voxel visited (brick jump, or a jump attempt that was refused -- state unchanged),
//         1: voxel (vx,vy,vz) visited, not flagged, 2: visited and flagged.
template <bool LDS>
__device__ __forceinline__ int walk_hier(const KGrid& g, const unsigned int* lds_coarse, Walk& w,
                                         float step_size, float tmax, bool allow_jump) {
    const int nx = w.nx, ny = w.ny, nz = w.nz;
    const int key = ((nx >> 3) << 20) | ((ny >> 3) << 10) | (nz >> 3);
    if (key != w.cur_key) {
        w.cur_key = key;
        const int bz = nz >> 3;
        const int bxy = (nx >> 3) * g.nb[1] + (ny >> 3);
        w.cur_brick = bxy * g.nb[2] + bz;
        const int wi = bxy * g.nwz + (bz >> 5);
        unsigned word;
        if constexpr (LDS) word = lds_coarse[wi];
        else word = g.coarse[wi];
        w.cur_flag = (word >> (bz & 31)) & 1u;
        w.cur_lx = -1;
    }
    // The cell: the whole brick if it holds no flagged voxel, else the voxel.  (Finer empty cells
    // read off the slab word -- 1x8x8 slabs, 1x4x4 quadrants -- were measured: the word load moves
    // onto the critical path of every step and the walk gets slower.)
    int cell_x = 0, cell_yz = 0;  // half-extent masks of the cell along x and along y/z
    if (allow_jump && w.straight && !w.cur_flag && key != w.noskip_key) cell_x = cell_yz = 7;
    const bool jump = (cell_x | cell_yz) != 0;
    const int sx = w.sx, sy = w.sy, sz = w.sz;
    const int l0 = w.lim0, l1 = w.lim1, l2 = w.lim2;
    const float ox = w.ox, oy = w.oy, oz = w.oz, dx = w.dx, dy = w.dy, dz = w.dz;
    // last voxel of the cell in the direction of travel, per axis, and the time it is left
    const int ex = (sx > 0) ? min(nx | cell_x, l0 - 1) : (nx & ~cell_x);
    const int ey = (sy > 0) ? min(ny | cell_yz, l1 - 1) : (ny & ~cell_yz);
    const int ez = (sz > 0) ? min(nz | cell_yz, l2 - 1) : (nz & ~cell_yz);
    float Tx, Ty, Tz;  // == plane_time(e, s, o, d) per axis
    if (w.fast_div) {
        Tx = quotient((float)(ex + ((sx + 1) >> 1)) - ox, dx, w.rdx);
        Ty = quotient((float)(ey + ((sy + 1) >> 1)) - oy, dy, w.rdy);
        Tz = quotient((float)(ez + ((sz + 1) >> 1)) - oz, dz, w.rdz);
    } else {
        Tx = plane_time(ex, sx, ox, dx);
        Ty = plane_time(ey, sy, oy, dy);
        Tz = plane_time(ez, sz, oz, dz);
    }
    const float T = fminf(fminf(Tx, Ty), Tz);
    const bool a0 = (T == Tx);
    const bool a1 = !a0 && (T == Ty);
    const bool a2 = !a0 && !a1;
    int cx = nx, cy = ny, cz = nz;
    if (jump) {
        const float px = fmaf(T, dx, ox), py = fmaf(T, dy, oy), pz = fmaf(T, dz, oz);
        const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
        // t <= ~2000 and |d| <= 1 at grid sizes <= 1024: T, o + T d and the plane times each carry
        // < 2.5e-4 voxel of float error, 6e-4 together; 1/256 leaves a 6x margin
        const float kEps = 1.f / 256.f;
        // kEps < frac < 1 - kEps as one comparison (|frac - 1/2| below 1/2 - kEps: 0.5 and kEps are exact binary
        // fractions, and a frac that close to a plane fails either form)
        const bool far_x = a0 || fabsf((px - fx) - 0.5f) < 0.5f - kEps;
        const bool far_y = a1 || fabsf((py - fy) - 0.5f) < 0.5f - kEps;
        const bool far_z = a2 || fabsf((pz - fz) - 0.5f) < 0.5f - kEps;
        cx = (int)fx;
        cy = (int)fy;
        cz = (int)fz;
        // the other axes must land inside the cell and not behind the current voxel: while the
        // ray is still approaching the grid (indices clamped by walk_init) its position is not
        // inside the voxel the state names
        // ((c - n) ^ m) >= m with m = s >> 31 is (c - n) * s >= 0 without the quarter-rate multiply
        const int mx_ = sx >> 31, my_ = sy >> 31, mz_ = sz >> 31;
        const bool rx = a0 || (((cx ^ nx) & ~cell_x) == 0 && ((cx - nx) ^ mx_) >= mx_);
        const bool ry = a1 || (((cy ^ ny) & ~cell_yz) == 0 && ((cy - ny) ^ my_) >= my_);
        const bool rz = a2 || (((cz ^ nz) & ~cell_yz) == 0 && ((cz - nz) ^ mz_) >= mz_);
        const bool ok = far_x && far_y && far_z && rx && ry && rz && (T <= tmax) && !(T + step_size > tmax);
        if (!ok) {  // walk this brick voxel by voxel
            w.noskip_key = key;
            return 0;
        }
    }
    w.vx = nx;
    w.vy = ny;
    w.vz = nz;
    const int n_exit = a0 ? ex + sx : (a1 ? ey + sy : ez + sz);
    const int lim = a0 ? l0 : (a1 ? l1 : l2);
    const bool leaves = (n_exit < 0) || (n_exit >= lim);
    w.t_far = T;
    w.t = leaves ? w.t_leave : T;
    w.nx = a0 ? n_exit : cx;
    w.ny = a1 ? n_exit : cy;
    w.nz = a2 ? n_exit : cz;
    if (jump) return 0;
    if (!w.cur_flag) return 1;
    const int lx = nx & 7;
    if (lx != w.cur_lx) {
        w.cur_lx = lx;
        w.cur_word = g.fine[(size_t)w.cur_brick * 8 + lx];
    }
    return 1 + (int)((w.cur_word >> (((ny & 7) << 3) | (nz & 7))) & 1ull);
}

__device__ __forceinline__ bool live_test(const KGrid& g, const Walk& w) {
    const int brick = ((w.vx >> 3) * g.nb[1] + (w.vy >> 3)) * g.nb[2] + (w.vz >> 3);
    const unsigned long long word = g.fine_live[(size_t)brick * 8 + (w.vx & 7)];
    return (word >> (((w.vy & 7) << 3) | (w.vz & 7))) & 1ull;
}

// all 8 corner links present?  (only needed for the backward loop quirk, surf_trav.cu:1925-1937)
__device__ __forceinline__ bool links_present_at(const KGrid& g, int vx, int vy, int vz) {
    if (vx + 1 >= g.size[0] || vy + 1 >= g.size[1] || vz + 1 >= g.size[2]) return false;
    const int32_t* p = g.links + ((size_t)vx * g.offx + (size_t)vy * g.offy + vz);
    return (p[0] | p[1] | p[g.offy] | p[g.offy + 1] | p[g.offx] | p[g.offx + 1] |
            p[g.offx + g.offy] | p[g.offx + g.offy + 1]) >= 0;
}
__device__ __forceinline__ bool links_present(const KGrid& g, const Walk& w) {
    return links_present_at(g, w.vx, w.vy, w.vz);
}

// ------------------------------------------------------------------------------------
// Voxel geometry for a lane that reached a work voxel (surf_trav.cu:212-260)
// ------------------------------------------------------------------------------------
struct Vox {
    int vx, vy, vz;
    int lk[8];
    float dens[8];
    float surf[8];
    double fs[4];
    float new_origin[3];  // o + t_close d  (float, as computed by the reference's FFMA)
    float smin, smax;
    float t_close, t_far;
};

// all 8 corner densities below the gate: the reference skips the voxel (surf_trav.cu:230-239).  The per-call planes hold
// the test; under planes of links only (KOpt::lazy_gates) the one-kernel paths apply it to the voxels they load.
__device__ __forceinline__ bool voxel_all_below(const Vox& v, float sigma_thresh) {
    bool all_below = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) all_below = all_below && (v.dens[i] < sigma_thresh);
    return all_below;
}

// Work voxels found by the walk are parked in a small per-lane FIFO (LDS) so that the lanes
// of a wave do not have to rendezvous at every one of them: x,y,z packed 10 bits each.
constexpr int kQueueDepth = 8;
__device__ __forceinline__ int pack_voxel(int x, int y, int z) { return (x << 20) | (y << 10) | z; }

// Entry/exit times are recomputed from the voxel coordinates with the DDA's own expressions
// (surf_trav.cu:162-184), so they are bit-identical to what the walk saw.
__device__ __forceinline__ void voxel_times(const Ray& r, int vx, int vy, int vz, float* tc, float* tf) {
    tc[0] = ((float)(r.d[0] > 0.f ? vx : vx + 1) - r.o[0]) / r.d[0];
    tc[1] = ((float)(r.d[1] > 0.f ? vy : vy + 1) - r.o[1]) / r.d[1];
    tc[2] = ((float)(r.d[2] > 0.f ? vz : vz + 1) - r.o[2]) / r.d[2];
    tf[0] = ((float)(r.d[0] > 0.f ? vx + 1 : vx) - r.o[0]) / r.d[0];
    tf[1] = ((float)(r.d[1] > 0.f ? vy + 1 : vy) - r.o[1]) / r.d[1];
    tf[2] = ((float)(r.d[2] > 0.f ? vz + 1 : vz) - r.o[2]) / r.d[2];
}

// Rays whose walk the closed-form tests below describe: finite origin, direction and bounds.  A ZERO direction
// component is fine: its plane times are +-inf (or NaN, which the reference's fminf / fmaxf / == ignore, i.e.
// "never crossed"; walk_visits reads a NaN the same way).  With -0 the axis is simply never stepped; with +0
// the reference steps down that axis at t = -inf until it leaves the grid and every voxel it sees has
// t_close = +inf (no sample): the closed form keeps the start voxel only, which yields no sample either.
__device__ __forceinline__ bool ray_is_regular(const Ray& r) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) ok = ok && (fabsf(r.d[k]) < 3.0e38f) && (fabsf(r.o[k]) < 3.0e38f);
    return ok && (fabsf(r.tmin) < 3.0e38f) && (fabsf(r.tmax) < 3.0e38f);
}

// first voxel of the walk (surf_trav.cu:75-79)
__device__ __forceinline__ void walk_start(const KGrid& g, const Ray& r, int* n0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) n0[k] = min(max((int)fmaf(r.tmin, r.d[k], r.o[k]), 0), g.size[k] - 2);
}

// Does the reference's walk (surf_trav.cu:86-197) visit voxel v?  The walk is a three-way merge: per axis
// the far-plane crossing times of the successive indices form an increasing sequence (the same float
// expression of the plane index however it is reached), each step takes the smallest pending one (equal
// times: the lowest axis), and it stops at the first step whose time exceeds tmax or that leaves the
// lattice.  A merge passes through the state v exactly when every crossing that leads to v -- on an axis
// where v differs from the start voxel n0, the near plane of v -- comes before every crossing that leads
// past it -- the far planes of v on the other axes -- in (time, axis) order; v is then entered at the
// latest of those near-plane times, which must not exceed tmax.  tc / tf: voxel_times(v).
// *t_entry = that time (tmin for the start voxel).
__device__ __forceinline__ bool walk_visits(const Ray& r, const int* n0, const int* v, const float* tc_in,
                                            const float* tf_in, float* t_entry) {
    bool ok = true;
    bool moved[3];
    float tc[3], tf[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {  // 0 / 0 on an axis the ray does not move along: a crossing that never comes
        tc[a] = (tc_in[a] != tc_in[a]) ? INFINITY : tc_in[a];  // (only read for an axis on which v is off the start)
        tf[a] = (tf_in[a] != tf_in[a]) ? INFINITY : tf_in[a];
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int dlt = (r.d[a] > 0.f) ? v[a] - n0[a] : n0[a] - v[a];
        ok = ok && (dlt >= 0);
        moved[a] = dlt > 0;
    }
    float te = r.tmin;
    bool any = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (a == b) continue;
            const bool before = (tc[a] < tf[b]) || (tc[a] == tf[b] && a < b);
            ok = ok && (!moved[a] || before);
        }
        te = (moved[a] && (!any || tc[a] > te)) ? tc[a] : te;
        any = any || moved[a];
    }
    *t_entry = te;
    return ok && (!any || te <= r.tmax);
}

// The reference's BACKWARD loop adds step_size to t after a voxel with a missing corner link before it tests
// t <= tmax (surf_trav.cu:1925-1937), so it may stop up to step_size before the forward loop does.  Is the
// visited voxel v cut off that way, i.e. does the walk pass, before v, through a voxel with a missing link
// whose exit time T satisfies T + step_size > tmax?  Exit times grow along the walk, so only the voxels
// right before v matter: the walk is retraced from v -- the predecessor lies across the near plane crossed
// last, in (time, axis) order -- while T + step_size > tmax.
__device__ __forceinline__ bool walk_cut_before(const KGrid& g, const Ray& r, float step_size, const int* n0,
                                                const int* v, const float* tc_v) {
    int c[3] = {v[0], v[1], v[2]};
    float tc[3] = {tc_v[0], tc_v[1], tc_v[2]};
    // The ray's components as plain values.  Selected among by `best` straight out of the struct, the compiler turns
    // the selects of loads into ONE load at a selected address -- an indexed access that pins the caller's whole Ray
    // in scratch memory (36 bytes a lane, written for every entry of k_sweep's phase 3 although this function runs
    // for a handful).  The empty asm makes them values it cannot trace back to memory.
    float ox = r.o[0], oy = r.o[1], oz = r.o[2], dx = r.d[0], dy = r.d[1], dz = r.d[2];
    asm volatile("" : "+v"(ox), "+v"(oy), "+v"(oz), "+v"(dx), "+v"(dy), "+v"(dz));
    for (int guard = 0; guard < 4096; ++guard) {
        int best = -1;
        float T = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (c[a] != n0[a] && (best < 0 || tc[a] >= T)) {  // >=: equal times, the higher axis came later
                best = a;
                T = tc[a];
            }
        }
        if (best < 0) return false;               // reached the start voxel
        if (!(T + step_size > r.tmax)) return false;
        // (indexing r.d / c by `best` would put them in scratch: select instead)
        const float o_b = best == 0 ? ox : (best == 1 ? oy : oz);
        const float d_b = best == 0 ? dx : (best == 1 ? dy : dz);
        const int s = (d_b > 0.f) ? 1 : -1;
        const int p0 = c[0] - (best == 0 ? s : 0), p1 = c[1] - (best == 1 ? s : 0), p2 = c[2] - (best == 2 ? s : 0);
        if (!links_present_at(g, p0, p1, p2)) return true;
        c[0] = p0;
        c[1] = p1;
        c[2] = p2;
        const int c_b = best == 0 ? p0 : (best == 1 ? p1 : p2);
        const float tcn = ((float)(d_b > 0.f ? c_b : c_b + 1) - o_b) / d_b;
        tc[0] = best == 0 ? tcn : tc[0];
        tc[1] = best == 1 ? tcn : tc[1];
        tc[2] = best == 2 ? tcn : tc[2];
    }
    return false;
}

__device__ __forceinline__ void load_voxel_links(const KGrid& g, Vox& v) {
    const int32_t* p = g.links + ((size_t)v.vx * g.offx + (size_t)v.vy * g.offy + v.vz);
    v.lk[0] = p[0];
    v.lk[1] = p[1];
    v.lk[2] = p[g.offy];
    v.lk[3] = p[g.offy + 1];
    v.lk[4] = p[g.offx];
    v.lk[5] = p[g.offx + 1];
    v.lk[6] = p[g.offx + g.offy];
    v.lk[7] = p[g.offx + g.offy + 1];
}

// the corner values and the cubic, once the links are there
__device__ __forceinline__ void load_voxel_values(const KGrid& g, const Ray& r, const float* tc, const float* tf,
                                                  Vox& v) {
    v.t_close = fmaxf(fmaxf(fmaxf(tc[0], tc[1]), tc[2]), 0.f);
    v.t_far = fminf(fminf(tf[0], tf[1]), tf[2]);
    struct { int vx, vy, vz; } w = {v.vx, v.vy, v.vz};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v.dens[i] = g.density[v.lk[i]];
        v.surf[i] = g.surface[v.lk[i]];
    }
    double no[3];
    const int vv[3] = {w.vx, w.vy, w.vz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        v.new_origin[k] = fmaf(v.t_close, r.d[k], r.o[k]);
        no[k] = (double)v.new_origin[k] - (double)vv[k];
    }
    surface_cubic(v.surf, no, r.d, v.fs);
    v.smin = v.surf[0];
    v.smax = v.surf[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        v.smin = fminf(v.smin, v.surf[i]);
        v.smax = fmaxf(v.smax, v.surf[i]);
    }
}

__device__ __forceinline__ void load_voxel(const KGrid& g, const Ray& r, int packed, Vox& v) {
    v.vx = packed >> 20;
    v.vy = (packed >> 10) & 1023;
    v.vz = packed & 1023;
    float tc[3], tf[3];
    voxel_times(r, v.vx, v.vy, v.vz, tc, tf);
    load_voxel_links(g, v);
    load_voxel_values(g, r, tc, tf, v);
}

// sample position of root st inside the voxel; false when outside [0,1]^3 (surf_trav.cu:322-340)
__device__ __forceinline__ bool root_position(const Ray& r, const Vox& v, double st, float* pos) {
    const int vv[3] = {v.vx, v.vy, v.vz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pos[k] = fmaf((float)st, r.d[k], v.new_origin[k]);
        pos[k] -= (float)vv[k];  // l = min(voxel, size-2) == voxel: the walk never leaves [0,size-2]
    }
    return !((pos[0] < 0) | (pos[0] > 1) | (pos[1] < 0) | (pos[1] > 1) | (pos[2] < 0) | (pos[2] > 1));
}

// Conservative test in front of the cubic solver (k_geom's light stage): false only when NO root of any in-range
// level set's cubic can pass `st > 0 && root_position()`.  Float arithmetic on the voxel's links and surface values
// alone (no densities, no f64 coefficients: those are the heavy stage's), with margins that cover it:
//   * where a sample can be: root_position() wants pos_k = fmaf(st, d_k, new_origin_k) - voxel_k in [0, 1] on every
//     axis; in st that is an interval per axis, here widened by 5e-4 voxel for the float roundings of pos (a few ulp
//     of a coordinate <= 1024: 1.2e-4), cut at st = 0.  Its length h satisfies h |d_k| <= 1.001.
//   * the cubic over that interval [a, a + h] in Bernstein form: its values lie between the smallest and the largest
//     coefficient, so coefficients of one sign mean no root.  In units of the interval every term of the
//     coefficients is a few times max |s| at most, ~40 float operations deep: the margin, 1e-4 (max |s| + |level|),
//     is ten times the bound of their rounding (and 3e-4 voxel of an SDF in voxel units: it hardly costs a voxel).
//   * the solver's roots are those of the polynomial up to ITS rounding, far below these margins, except where the
//     cubic term is so small against the quadratic one that the Vieta form cancels (root = m cos(..) - f2 / 3 f3):
//     f3 -- the third difference of the corners times d_x d_y d_z, formed in f64 -- at or above the solver's own
//     1e-10 switch to the quadratic formula but below 1e-9 of the quadratic term (in units of the interval).  Such
//     voxels are passed on unseen.
__device__ __forceinline__ bool voxel_may_hold_root(const KGrid& g, const Ray& r, int packed) {
    const int vv[3] = {packed >> 20, (packed >> 10) & 1023, packed & 1023};
    float t_close = 0.f;  // as load_voxel_values
#pragma unroll
    for (int k = 0; k < 3; ++k) t_close = fmaxf(t_close, ((float)(r.d[k] > 0.f ? vv[k] : vv[k] + 1) - r.o[k]) / r.d[k]);
    const int32_t* p = g.links + ((size_t)vv[0] * g.offx + (size_t)vv[1] * g.offy + vv[2]);
    const int32_t lk[8] = {p[0], p[1], p[g.offy], p[g.offy + 1], p[g.offx], p[g.offx + 1], p[g.offx + g.offy],
                           p[g.offx + g.offy + 1]};
    float sf[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sf[i] = g.surface[lk[i]];
    float smin = sf[0], smax = sf[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) {
        smin = fminf(smin, sf[i]);
        smax = fmaxf(smax, sf[i]);
    }
    float no[3];  // the entry point inside the voxel
    float a = 0.f, b = INFINITY;
    const float tol = 5e-4f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        no[k] = fmaf(t_close, r.d[k], r.o[k]) - (float)vv[k];
        const float lo = -no[k] - tol, hi = -no[k] + (1.f + tol);
        if (fabsf(r.d[k]) > 1e-20f) {
            const float inv = 1.f / r.d[k];
            const float t0 = lo * inv, t1 = hi * inv;
            a = fmaxf(a, fminf(t0, t1));
            b = fminf(b, fmaxf(t0, t1));
        } else if (lo > 0.f || hi < 0.f) {
            return false;  // the position never moves along k and is outside
        }
    }
    if (!(a <= b)) return false;
    if (!(b < 1e30f) || !(t_close < 1e30f)) return true;
    const float h = (b - a) * 1.0002f + 1e-30f;  // (both ends out by 1e-4 of the length: the divisions above)
    a -= 1e-4f * (b - a);
    // the trilinear field along the ray, multiplied out z, y, x (surface_cubic, in float)
    float c[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a0 = fmaf(sf[4 * i + 1] - sf[4 * i], no[2], sf[4 * i]), b0 = r.d[2] * (sf[4 * i + 1] - sf[4 * i]);
        const float a1 = fmaf(sf[4 * i + 3] - sf[4 * i + 2], no[2], sf[4 * i + 2]),
                    b1 = r.d[2] * (sf[4 * i + 3] - sf[4 * i + 2]);
        c[i][0] = fmaf(a1 - a0, no[1], a0);
        c[i][1] = fmaf(b1 - b0, no[1], b0) + r.d[1] * (a1 - a0);
        c[i][2] = r.d[1] * (b1 - b0);
    }
    const float f0 = fmaf(c[1][0] - c[0][0], no[0], c[0][0]);
    const float f1 = fmaf(c[1][1] - c[0][1], no[0], c[0][1]) + r.d[0] * (c[1][0] - c[0][0]);
    const float f2 = fmaf(c[1][2] - c[0][2], no[0], c[0][2]) + r.d[0] * (c[1][1] - c[0][1]);
    const float f3 = r.d[0] * (c[1][2] - c[0][2]);
    // in units of the interval: p(a + u h) = c0 + c1 u + c2 u^2 + c3 u^3
    const float c1 = h * fmaf(fmaf(3.f * f3, a, 2.f * f2), a, f1);
    const float c2 = h * h * fmaf(3.f * f3, a, f2);
    const float c3 = h * h * h * f3;
    const float p_a = fmaf(fmaf(fmaf(f3, a, f2), a, f1), a, f0);
    // Vieta's cancellation zone
    const double f3d = (double)r.d[0] * (double)r.d[1] * (double)r.d[2] *
                       ((((double)sf[7] - (double)sf[6]) - ((double)sf[5] - (double)sf[4])) -
                        (((double)sf[3] - (double)sf[2]) - ((double)sf[1] - (double)sf[0])));
    const float smag = fmaxf(fabsf(smin), fabsf(smax));
    if (fabs(f3d) >= 0.9e-10 && fabs(f3d) * (double)h * (double)h * (double)h < 1e-9 * (double)(fabsf(c2) + 1e-3f * smag))
        return true;
    bool maybe = false;
    for (int li = 0; li < g.level_set_num; ++li) {
        const float lvf = g.level_set[li];
        if (lvf < smin || lvf > smax) continue;
        const float c0 = p_a - lvf;
        const float b1 = fmaf(c1, 1.f / 3.f, c0), b2 = fmaf(2.f * c1 + c2, 1.f / 3.f, c0), b3 = c0 + c1 + c2 + c3;
        // relative to the field's magnitude (the float evaluation above), with an absolute floor: where the solver takes
        // its quadratic / linear form (|f3| or |f2| below 1e-10) it drops a term worth up to 1e-9 over the interval,
        // whatever the field's scale -- a field of tiny values must not be filtered on a margin smaller than that
        const float margin = 1e-4f * (smag + fabsf(lvf)) + 1e-8f;
        const float mn = fminf(fminf(c0, b1), fminf(b2, b3)), mx = fmaxf(fmaxf(c0, b1), fmaxf(b2, b3));
        if (!(mn > margin || mx < -margin)) maybe = true;
    }
    return maybe;
}

// fake-sample geometry (surf_trav.cu:437-490)
__device__ __forceinline__ void fake_position(const Ray& r, const Vox& v, float* pos) {
    const float tm = (v.t_far + v.t_close) / 2.f;
    const int vv[3] = {v.vx, v.vy, v.vz};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        pos[k] = fmaf(tm, r.d[k], r.o[k]);
        pos[k] -= (float)vv[k];
    }
}
__device__ __forceinline__ float fake_distance(const KGrid& g, const KOpt& o, const Vox& v,
                                               const float* pos, double* miu_out, double* std_out) {
    double miu = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) miu += (double)v.surf[i];
    miu = miu / 8;
    double var = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) var += sqrd((double)v.surf[i] - miu);
    double sd = (double)sqrtf((float)fmax((double)1e-9f, var / 8));
    if (!o.fake_sample_normalize_surf) sd = 1.;
    float ns[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ns[i] = (float)((double)v.surf[i] / sd);
    const float fs_s = trilerp8(ns, pos);
    float dist = INFINITY;
    for (int i = 0; i < g.level_set_num; ++i) {
        const float dd = fs_s - g.level_set[i];
        dist = (fabsf(dd) < fabsf(dist)) ? dd : dist;
    }
    *miu_out = miu;
    *std_out = sd;
    return dist;
}

// ------------------------------------------------------------------------------------
// Wave-cooperative shading.  Lane layout: lane = cs * D + e with cs the corner slot of
// this round and e = ch * B + k the SH coefficient; CPR corners are served per round.
// ------------------------------------------------------------------------------------
struct LaneRec {
    int link[8];
};

template <int B>
struct Coop {
    static constexpr int D = 3 * B;
    static constexpr int CPR = (64 / D) < 8 ? (64 / D) : 8;
    static constexpr int ROUNDS = (8 + CPR - 1) / CPR;
};

// trilinear weight of corner c (bit 2 = x, bit 1 = y, bit 0 = z)
__device__ __forceinline__ float corner_w(int c, float px, float py, float pz) {
    const float wx = (c & 4) ? px : 1.f - px;
    const float wy = (c & 2) ? py : 1.f - py;
    const float wz = (c & 1) ? pz : 1.f - pz;
    return wx * wy * wz;
}

// Forward: colour[ch] = sum_k sph[k] * trilerp(sh[.., ch*B+k]) for every lane in `need`.
template <int B>
__device__ __forceinline__ void coop_shade_fwd(const KGrid& g, unsigned long long need,
                                               const LaneRec* rec_wave, const float* sph_wave,
                                               const float* pos, float* color) {
    using C = Coop<B>;
    const int lane = threadIdx.x & 63;
    const int cs = lane / C::D;
    const int e = lane - cs * C::D;
    const int ch = e / B;
    const int k = e - ch * B;
    const bool on = cs < C::CPR;
    // neighbouring rays mostly sit in the same voxel: its SH rows stay in registers until the
    // voxel (identified by the row id of its first corner) changes -- a wave-uniform test
    const int my_first = rec_wave[lane].link[0];
    int held = -1;
    float val[C::ROUNDS];
#pragma unroll
    for (int rd = 0; rd < C::ROUNDS; ++rd) val[rd] = 0.f;
    while (need) {
        const int r = __ffsll((long long)need) - 1;
        need &= need - 1;
        const float px = bcast_f(pos[0], r), py = bcast_f(pos[1], r), pz = bcast_f(pos[2], r);
        const int first = bcast_i(my_first, r);
        if (first != held || first < 0) {  // a missing first corner (cuvol cells) identifies nothing
            held = first;
#pragma unroll
            for (int rd = 0; rd < C::ROUNDS; ++rd) {
                const int c = rd * C::CPR + cs;
                if (on && c < 8) {
                    const int link = rec_wave[r].link[c];  // < 0 in cuvol cells: the corner reads as 0
                    val[rd] = link >= 0 ? g.sh[(size_t)link * C::D + e] : 0.f;
                }
            }
        }
        float acc = 0.f;
#pragma unroll
        for (int rd = 0; rd < C::ROUNDS; ++rd) {
            const int c = rd * C::CPR + cs;
            if (on && c < 8) acc = fmaf(corner_w(c, px, py, pz), val[rd], acc);
        }
        acc = on ? acc * sph_wave[r * kSphPad + k] : 0.f;
        const float s0 = wave_sum(ch == 0 ? acc : 0.f);
        const float s1 = wave_sum(ch == 1 ? acc : 0.f);
        const float s2 = wave_sum(ch == 2 ? acc : 0.f);
        if (lane == r) {
            color[0] = s0;
            color[1] = s1;
            color[2] = s2;
        }
    }
}

// Sum over each aligned group of 8 lanes, left in the group's last lane (3 DPP row shifts).
__device__ __forceinline__ float dpp_sum8(float v) {
#define ASURF_DPP_ADD8(ctrl)                                                                           \
    v += __builtin_bit_cast(float,                                                                     \
                            __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, false))
    ASURF_DPP_ADD8(0x111);
    ASURF_DPP_ADD8(0x112);
    ASURF_DPP_ADD8(0x114);
#undef ASURF_DPP_ADD8
    return v;
}

// Sum over each half wave, left in lanes 31 and 63.
__device__ __forceinline__ float dpp_sum32(float v) {
#define ASURF_DPP_ADD32(ctrl, row_mask)                                                                \
    v += __builtin_bit_cast(                                                                           \
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, row_mask, 0xf, false))
    ASURF_DPP_ADD32(0x111, 0xf);
    ASURF_DPP_ADD32(0x112, 0xf);
    ASURF_DPP_ADD32(0x114, 0xf);
    ASURF_DPP_ADD32(0x118, 0xf);
    ASURF_DPP_ADD32(0x142, 0xa);  // row_bcast:15 into rows 1 and 3
#undef ASURF_DPP_ADD32
    return v;
}

// The next two set lanes of `need`: half 0 of the wave serves r0, half 1 serves r1 (if `two`).
__device__ __forceinline__ void next_pair(unsigned long long& need, int* r0, int* r1, bool* two) {
    *r0 = __ffsll((long long)need) - 1;
    need &= need - 1;
    *two = need != 0;
    *r1 = *two ? __ffsll((long long)need) - 1 : *r0;
    if (*two) need &= need - 1;
}

// coop_shade_fwd in the (channel, corner) layout: two samples are served at a time, one per half
// wave.  In a half, lane (ch, c) = (lane >> 3 & 3, lane & 7) -- the fourth octet idles -- loads the B
// coefficients of channel ch in corner c's SH row (contiguous), dots them with the ray's SH basis,
// scales by the corner's trilinear weight, and a 3-step DPP sum over the octet is the channel's
// colour: ~35 wave instructions per sample instead of ~150 (three 64-lane reductions per sample).
template <int B>
__device__ __forceinline__ void pair_shade_fwd(const KGrid& g, unsigned long long need, const LaneRec* rec_wave,
                                               const float* sph_wave, const float* pos, float* color) {
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, ch = (lane >> 3) & 3, c = lane & 7;
    const int my_first = rec_wave[lane].link[0];
    int held = -1;
    float val[B];
#pragma unroll
    for (int k = 0; k < B; ++k) val[k] = 0.f;
    while (need) {
        int r0, r1;
        bool two;
        next_pair(need, &r0, &r1, &two);
        const int rr = half ? r1 : r0;
        const bool mine = ch < 3 && (half == 0 || two);
        const float px = half ? bcast_f(pos[0], r1) : bcast_f(pos[0], r0);
        const float py = half ? bcast_f(pos[1], r1) : bcast_f(pos[1], r0);
        const float pz = half ? bcast_f(pos[2], r1) : bcast_f(pos[2], r0);
        const int first = half ? bcast_i(my_first, r1) : bcast_i(my_first, r0);
        if (mine && first != held) {
            held = first;
            const float* row = g.sh + ((size_t)rec_wave[rr].link[c] * (3 * B) + ch * B);
#pragma unroll
            for (int k = 0; k < B; ++k) val[k] = row[k];
        }
        float d = 0.f;
#pragma unroll
        for (int k = 0; k < B; ++k) d = fmaf(sph_wave[rr * kSphPad + k], val[k], d);
        const float a = dpp_sum8(mine ? corner_w(c, px, py, pz) * d : 0.f);
        const float s00 = bcast_f(a, 7), s01 = bcast_f(a, 15), s02 = bcast_f(a, 23);
        const float s10 = bcast_f(a, 39), s11 = bcast_f(a, 47), s12 = bcast_f(a, 55);
        if (lane == r0) {
            color[0] = s00;
            color[1] = s01;
            color[2] = s02;
        }
        if (two && lane == r1) {
            color[0] = s10;
            color[1] = s11;
            color[2] = s12;
        }
    }
}

// Backward: recomputes the colour, scatters d/d(sh) with row-contiguous atomics and
// returns (to the owning lane) sum_c gout_c * max(colour_c + .5, 0) and d(loss)/d(pos)
// through the SH coefficients.  surf_trav.cu:2089-2139, 2241-2264.
template <int B>
__device__ __forceinline__ void coop_shade_bwd(const KGrid& g, const KGrads& G, bool want_dpos,
                                               unsigned long long need, const LaneRec* rec_wave,
                                               const float* sph_wave, const float* pos, float weight,
                                               const float* gout, float* total_color,
                                               float* grad_xyz) {
    using C = Coop<B>;
    const int lane = threadIdx.x & 63;
    const int cs = lane / C::D;
    const int e = lane - cs * C::D;
    const int ch = e / B;
    const int k = e - ch * B;
    const bool on = cs < C::CPR;
    while (need) {
        const int r = __ffsll((long long)need) - 1;
        need &= need - 1;
        const float px = bcast_f(pos[0], r), py = bcast_f(pos[1], r), pz = bcast_f(pos[2], r);
        const float wgt = bcast_f(weight, r);
        const float g0 = bcast_f(gout[0], r), g1 = bcast_f(gout[1], r), g2 = bcast_f(gout[2], r);
        float acc = 0.f, ax = 0.f, ay = 0.f, az = 0.f;
        int link_r[C::ROUNDS];
        float w_r[C::ROUNDS];
#pragma unroll
        for (int rd = 0; rd < C::ROUNDS; ++rd) {
            const int c = rd * C::CPR + cs;
            link_r[rd] = -1;
            w_r[rd] = 0.f;
            if (on && c < 8) {
                const int link = rec_wave[r].link[c];
                const float val = link >= 0 ? g.sh[(size_t)link * C::D + e] : 0.f;
                const float wx = (c & 4) ? px : 1.f - px;
                const float wy = (c & 2) ? py : 1.f - py;
                const float wz = (c & 1) ? pz : 1.f - pz;
                const float w = wx * wy * wz;
                acc = fmaf(w, val, acc);
                ax = fmaf(((c & 4) ? wy : -wy) * wz, val, ax);
                ay = fmaf(((c & 2) ? wx : -wx) * wz, val, ay);
                az = fmaf(((c & 1) ? wx : -wx) * wy, val, az);
                link_r[rd] = link;
                w_r[rd] = w;
            }
        }
        const float sphk = on ? sph_wave[r * kSphPad + k] : 0.f;
        acc *= sphk;
        const float s0 = wave_sum(ch == 0 ? acc : 0.f) + 0.5f;
        const float s1 = wave_sum(ch == 1 ? acc : 0.f) + 0.5f;
        const float s2 = wave_sum(ch == 2 ? acc : 0.f) + 0.5f;
        const float t0 = fmaxf(s0, 0.f), t1 = fmaxf(s1, 0.f), t2 = fmaxf(s2, 0.f);
        const float total = (t0 * g0 + t2 * g2) + t1 * g1;  // (c0 + c2) + c1, :2112-2115
        const float in01 = (ch == 0) ? (t0 == s0 ? 1.f : 0.f)
                                     : (ch == 1) ? (t1 == s1 ? 1.f : 0.f) : (t2 == s2 ? 1.f : 0.f);
        const float gch = (ch == 0) ? g0 : (ch == 1) ? g1 : g2;
        const float cg = sphk * (wgt * in01 * gch);
        if (G.grad_sh != nullptr) {
#pragma unroll
            for (int rd = 0; rd < C::ROUNDS; ++rd) {
                if (link_r[rd] >= 0)
                    atomicAdd(&G.grad_sh[(size_t)link_r[rd] * C::D + e], w_r[rd] * cg);
            }
        }
        float gx = 0.f, gy = 0.f, gz = 0.f;
        if (want_dpos) {
            gx = wave_sum(cg * ax);
            gy = wave_sum(cg * ay);
            gz = wave_sum(cg * az);
        }
        if (lane == r) {
            *total_color = total;
            grad_xyz[0] = gx;
            grad_xyz[1] = gy;
            grad_xyz[2] = gz;
        }
    }
}

// Float fields of a sample record, planes [8][R][Q]: raw opacity, re-weighted alpha, the three
// pre-clamp colours, position inside the voxel (read by the split backward, asurf_bwd_pipeline.h)
__device__ __forceinline__ void write_rec_floats(float* rec_f, int R, int slot, long long Q, long long q,
                                                 float raw, float a, const float* color, const float* pos) {
    if (rec_f == nullptr) return;
    const size_t plane = (size_t)R * Q;
    float* p = rec_f + (size_t)slot * Q + q;
    p[0] = raw;
    p[plane] = a;
    p[2 * plane] = color[0];
    p[3 * plane] = color[1];
    p[4 * plane] = color[2];
    p[5 * plane] = pos[0];
    p[6 * plane] = pos[1];
    p[7 * plane] = pos[2];
}

// per-lane scatter of 8 corner contributions into a [N,1] gradient plane + touched mask
__device__ __forceinline__ void scatter8(float* out, uint8_t* mask, uint8_t* mask2, const int* lk, const float* w) {
    if (out == nullptr) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        atomicAdd(&out[lk[i]], w[i]);
        if (mask != nullptr) mask[lk[i]] = 1;
        if (mask2 != nullptr) mask2[lk[i]] = 1;
    }
}

// ------------------------------------------------------------------------------------
// Forward kernel: trace_ray_surf_trav (surf_trav.cu:37-562) for one ray per lane.
//   CACHE: also record (rwalpha, weight, t) of the first M samples for the fused losses.
//   COUNT: accumulate the V_t / V_a / S counters of SURVEY.md section 8(d).
// Sample cache layout: entry i of ray q at [i * Q + q] (lane-coalesced); cache_n[q] entries
// are valid, the rest are implicit zeros.
// ------------------------------------------------------------------------------------
template <int B, bool CACHE, bool COUNT, bool CLDS>
__global__ void __launch_bounds__(kBlock, ASURF_MIN_WAVES)
k_forward(KGrid g, KOpt o, const float* __restrict__ origins, const float* __restrict__ dirs,
          long long Q, float* __restrict__ rgb_out, int M, float* __restrict__ cache_a,
          float* __restrict__ cache_w, float* __restrict__ cache_t, int* __restrict__ cache_n,
          int* __restrict__ rec_v, int* __restrict__ rec_m, int* __restrict__ rec_n, int R,
          float* __restrict__ rec_f, const int* __restrict__ ray_list, const int* __restrict__ n_list,
          unsigned long long* counters) {
    extern __shared__ unsigned int s_coarse[];
    __shared__ float s_sph[kBlock * kSphPad];
    __shared__ LaneRec s_rec[kBlock];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // Sample records for the replaying backward kernel (k_backward<REC>): one entry per
    // composited sample, in order.  rec_cut = ordinal of the first work voxel the
    // reference's BACKWARD walk would not reach (its `t += step_size` quirk, :1935-1936).
    const bool do_rec = CACHE && (rec_v != nullptr);
    int rec_cnt = 0;
    bool rec_over = false;
    int n_pushed = 0, n_drained = 0, rec_cut = 0x7fffffff;
    // (as the pipeline's fallback this kernel usually finds an empty list: leave before anything is staged)
    if (ray_list != nullptr && (long long)blockIdx.x * kBlock >= (long long)*n_list) return;
    if constexpr (CLDS) {
        for (int i = tid; i < g.coarse_words; i += kBlock) s_coarse[i] = g.coarse[i];
        __syncthreads();
    }
    const float* sph_wave = s_sph + (tid & ~63) * kSphPad;
    const LaneRec* rec_wave = s_rec + (tid & ~63);

    // ray_list: render only the rays the three-kernel pipeline handed back (compacted ids; a
    // ray may be listed twice, which rewrites the same outputs)
    long long q = (long long)blockIdx.x * kBlock + tid;
    bool valid = q < Q;
    if (ray_list != nullptr) {
        valid = q < (long long)min(*n_list, (int)min(Q, (long long)0x7fffffff));
        q = valid ? ray_list[q] : 0;
    }
    if (!__any(valid)) return;
    Ray r;
    r.tmin = 1.f;
    r.tmax = 0.f;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r.o[k] = origins[3 * q + k];
            r.d[k] = dirs[3 * q + k];
        }
        float sph[B];
        eval_sh<B>(r.d, sph);
#pragma unroll
        for (int k = 0; k < B; ++k) s_sph[tid * kSphPad + k] = sph[k];
        setup_ray(g, o, r);
    }
    float outv[3] = {0.f, 0.f, 0.f};
    float log_transmit = 0.f;
    int sample_i = 0;
    int intersect_i = -1;
    int n_trav = 0, n_act = 0, n_samp = 0;
    bool active = valid && !(r.tmin > r.tmax);
    Walk w;
    if (active) walk_init(g, r, w);

    // Counting mode keeps a queue of one so that a ray stopped by stop_thresh is not walked
    // (and counted) past the voxel the reference would stop at.
    constexpr int QD = COUNT ? 1 : kQueueDepth;
    __shared__ int s_queue[QD * kBlock];
    int qn = 0;
    while (true) {
        // (A) walk; park work voxels in this lane's FIFO until it is full or the ray ends
        while (active && qn < QD) {
            walk_step(g, r, w);
            const bool hit = walk_test<CLDS>(g, s_coarse, w);
            if (COUNT) {
                ++n_trav;
                n_act += live_test(g, w) ? 1 : 0;
            }
            active = (w.t <= r.tmax);
            if (hit) {
                s_queue[qn * kBlock + tid] = pack_voxel(w.vx, w.vy, w.vz);
                ++qn;
                ++n_pushed;
            } else if (do_rec && active && (w.t + o.step_size > r.tmax) && rec_cut == 0x7fffffff) {
                if (!links_present(g, w)) rec_cut = n_pushed;
            }
        }
        if (ASURF_AID(o, 1)) qn = 0;  // measurement aid: traversal only
        if (!__any(qn > 0)) break;
      // (B,C) drain the FIFOs in order, slot by slot
      for (int qi = 0; __any(qi < qn); ++qi) {
        bool found = qi < qn;
        const bool rec_ok = do_rec && found && (n_drained < rec_cut);
        if (found) ++n_drained;
        // (B) geometry
        Vox v;
        if (found) load_voxel(g, r, s_queue[qi * kBlock + tid], v);
        if (o.lazy_gates && found && voxel_all_below(v, o.sigma_thresh)) found = false;   // (planes of links only: :230-239 here)
        bool vox_has_sample = false, vox_has_surf = false;
        // (C) real intersections, one level set at a time
        for (int li = 0; li < g.level_set_num; ++li) {
            const float lvf = g.level_set[li];
            const bool lv_ok = found && !(lvf < v.smin || lvf > v.smax) && !ASURF_AID(o, 2);
            if (!__any(lv_ok)) continue;
            double st[3] = {-1, -1, -1};
            if (lv_ok) {
                vox_has_surf = true;
                solve_cubic_vieta(v.fs[0] - (double)lvf, v.fs[1], v.fs[2], v.fs[3], st);
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                bool cand = lv_ok && (st[j] > 0) && !ASURF_AID(o, 4);
                float pos[3] = {0.f, 0.f, 0.f};
                if (cand) cand = root_position(r, v, st[j], pos);
                if (cand) vox_has_sample = true;
                if (cand && o.only_outward_intersect) {  // :350-367
                    float sg[3];
                    field_grad(v.surf, pos, sg);
                    const float ndd = -(sg[0] * r.d[0] + sg[1] * r.d[1] + sg[2] * r.d[2]);
                    if (ndd >= 0.f) cand = false;
                }
                bool shade = false;
                float rwalpha = 0.f, raw_alpha = 0.f;
                if (cand) {
                    ++intersect_i;
                    raw_alpha = trilerp8(v.dens, pos);
                    if (raw_alpha > o.sigma_thresh) {
                        const float alpha = alpha_act(raw_alpha, o.alpha_activation_type);
                        const float rw = o.truncated_vol_render
                                             ? trunc_rw((float)intersect_i, g.trunc_a, o.trunc_vol_weight_min)
                                             : 1.f;
                        rwalpha = alpha * rw;
                        shade = true;
#pragma unroll
                        for (int i = 0; i < 8; ++i) s_rec[tid].link[i] = v.lk[i];
                    }
                }
                const unsigned long long need = __ballot(shade);
                if (need) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    float color[3] = {0.f, 0.f, 0.f};
                    coop_shade_fwd<B>(g, need, rec_wave, sph_wave, pos, color);
                    if (shade) {
                        const float pcnt = -1 * logf(1 - rwalpha);
                        const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
                        log_transmit -= pcnt;
#pragma unroll
                        for (int c = 0; c < 3; ++c) outv[c] += weight * fmaxf(color[c] + 0.5f, 0.f);
                        if (COUNT) ++n_samp;
                        if (CACHE && sample_i < M) {
                            const size_t ci = (size_t)sample_i * Q + q;
                            cache_a[ci] = rwalpha;
                            cache_w[ci] = weight;
                            cache_t[ci] = (float)((double)v.t_close + st[j]);
                            sample_i += 1;
                        }
                        if (rec_ok) {
                            if (rec_cnt < R) {
                                rec_v[(size_t)rec_cnt * Q + q] = pack_voxel(v.vx, v.vy, v.vz);
                                rec_m[(size_t)rec_cnt * Q + q] = ((intersect_i + 1) << 8) | (li << 3) | (j << 1);
                                write_rec_floats(rec_f, R, rec_cnt, Q, q, raw_alpha, rwalpha, color, pos);
                                ++rec_cnt;
                            } else {
                                rec_over = true;
                            }
                        }
                    }
                }
            }
        }
        // fake sample (surf_trav.cu:423-541)
        {
            bool fk = found && o.surf_fake_sample && !vox_has_sample &&
                      (!o.limited_fake_sample || vox_has_surf) &&
                      ((v.t_far - v.t_close) > o.surf_fake_sample_min_vox_len);
            float pos[3] = {0.f, 0.f, 0.f};
            float alpha = 0.f, raw_f = 0.f;
            if (fk) {
                fake_position(r, v, pos);
                alpha = trilerp8(v.dens, pos);
                raw_f = alpha;
                fk = alpha > o.sigma_thresh;
            }
            if (fk) {
                alpha = alpha_act(alpha, o.alpha_activation_type);
                double miu, sd;
                const float dist = fake_distance(g, o, v, pos, &miu, &sd);
                alpha = alpha * expf((float)(-.5 * (double)sqrf(dist / g.fake_sample_std)));
                const float rw = o.truncated_vol_render
                                     ? trunc_rw((float)intersect_i, g.trunc_a, o.trunc_vol_weight_min)
                                     : 1.f;
                alpha = alpha * rw;
#pragma unroll
                for (int i = 0; i < 8; ++i) s_rec[tid].link[i] = v.lk[i];
            }
            const unsigned long long need = __ballot(fk);
            if (need) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                float color[3] = {0.f, 0.f, 0.f};
                coop_shade_fwd<B>(g, need, rec_wave, sph_wave, pos, color);
                if (fk) {
                    const float pcnt = -1 * logf(1 - alpha);
                    const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
                    log_transmit -= pcnt;
#pragma unroll
                    for (int c = 0; c < 3; ++c) outv[c] += weight * fmaxf(color[c] + 0.5f, 0.f);
                    if (COUNT) ++n_samp;
                    if (CACHE && sample_i < M && o.fake_sample_l_dist) {
                        const size_t ci = (size_t)sample_i * Q + q;
                        cache_a[ci] = alpha;
                        cache_w[ci] = weight;
                        cache_t[ci] = (v.t_far + v.t_close) / 2.f;
                        sample_i += 1;
                    }
                    if (rec_ok) {
                        if (rec_cnt < R) {
                            rec_v[(size_t)rec_cnt * Q + q] = pack_voxel(v.vx, v.vy, v.vz);
                            rec_m[(size_t)rec_cnt * Q + q] = ((intersect_i + 1) << 8) | 1;
                            write_rec_floats(rec_f, R, rec_cnt, Q, q, raw_f, alpha, color, pos);
                            ++rec_cnt;
                        } else {
                            rec_over = true;
                        }
                    }
                }
            }
        }
        if (found && expf(log_transmit) < o.stop_thresh) {  // :544-547
            log_transmit = -1e3f;
            active = false;
            qn = 0;  // voxels parked beyond the stopping point are dropped
        }
      }
        qn = 0;
    }
    if (valid) {
        const float bg = expf(log_transmit) * o.background_brightness;  // :553-555 (== bg for a miss)
#pragma unroll
        for (int c = 0; c < 3; ++c) rgb_out[3 * q + c] = outv[c] + bg;
        if (CACHE) cache_n[q] = sample_i;
        if (do_rec) rec_n[q] = rec_over ? -1 : rec_cnt;
    }
    if (COUNT && counters != nullptr) {
        const float ft = wave_sum((float)n_trav), fa = wave_sum((float)n_act),
                    fs = wave_sum((float)n_samp), fr = wave_sum(valid ? 1.f : 0.f);
        if (lane == 0) {
            atomicAdd(&counters[0], (unsigned long long)ft);
            atomicAdd(&counters[1], (unsigned long long)fa);
            atomicAdd(&counters[2], (unsigned long long)fs);
            atomicAdd(&counters[3], (unsigned long long)fr);
        }
    }
}

// ------------------------------------------------------------------------------------
// Backward kernel: render_ray_backward_kernel prologue + trace_ray_surf_trav_backward
// (surf_trav.cu:3241-3368, 1710-2911), one ray per lane.
//   FUSED: grad_in is rgb_gt and the image-loss gradient is formed here (:3306-3316);
//          otherwise grad_in is dL/drgb.
// ------------------------------------------------------------------------------------
//   REC:   (fused only) no walk at all -- replay the sample records the forward kernel left
//          behind (voxel, level set, root index, intersection ordinal per composited sample);
//          rays whose record overflowed (rec_n < 0) are left to a second, walking launch that
//          is restricted to them through `ray_filter`.
template <int B, bool FUSED, bool CLDS, bool REC>
__global__ void __launch_bounds__(kBlock, ASURF_MIN_WAVES)
k_backward(KGrid g, KOpt o, KLoss L, KGrads G, const float* __restrict__ origins,
           const float* __restrict__ dirs, long long Q, const float* __restrict__ grad_in,
           const float* __restrict__ color_cache, int M, const float* __restrict__ cache_a,
           const float* __restrict__ cache_w, const float* __restrict__ cache_t,
           const int* __restrict__ cache_n, const int* __restrict__ rec_v,
           const int* __restrict__ rec_m, const int* __restrict__ rec_n,
           const int* __restrict__ ray_filter, const int* __restrict__ any_a, const int* __restrict__ any_b) {
    extern __shared__ unsigned int s_coarse[];
    __shared__ float s_sph[kBlock * kSphPad];
    __shared__ LaneRec s_rec[kBlock];
    const int tid = threadIdx.x;
    // (the fused call's walking backward: two words say whether ANY ray was left to it -- rays handed to the one-kernel
    // path, rays whose sample records did not fit: on nearly every call neither, and the launch is two scalar loads)
    if (any_a != nullptr && *any_a == 0 && *any_b == 0) return;
    const long long q = (long long)blockIdx.x * kBlock + tid;
    bool valid = q < Q;
    if (valid) {
        if constexpr (REC) valid = rec_n[q] >= 0;
        else if (ray_filter != nullptr) valid = ray_filter[q] < 0;
    }
    // (as the replay's fallback this kernel usually finds no ray to walk: leave before anything is staged)
    if (ray_filter != nullptr && !__syncthreads_or(valid ? 1 : 0)) return;
    if constexpr (CLDS && !REC) {
        for (int i = tid; i < g.coarse_words; i += kBlock) s_coarse[i] = g.coarse[i];
        __syncthreads();
    }
    const float* sph_wave = s_sph + (tid & ~63) * kSphPad;
    const LaneRec* rec_wave = s_rec + (tid & ~63);

    Ray r;
    r.tmin = 1.f;
    r.tmax = 0.f;
    float gout[3] = {0.f, 0.f, 0.f};
    float accum = 0.f;
    if (valid) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            r.o[k] = origins[3 * q + k];
            r.d[k] = dirs[3 * q + k];
        }
        float sph[B];
        eval_sh<B>(r.d, sph);
#pragma unroll
        for (int k = 0; k < B; ++k) s_sph[tid * kSphPad + k] = sph[k];
        setup_ray(g, o, r);
        float cc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            cc[i] = color_cache[3 * q + i];
            if (FUSED) {
                const float resid = cc[i] - grad_in[3 * q + i];
                gout[i] = resid * L.norm_l2 * L.lambda_l2;
                gout[i] += (resid > 0.f) ? (L.norm_l1 * L.lambda_l1) : (-L.norm_l1 * L.lambda_l1);
            } else {
                gout[i] = grad_in[3 * q + i];
            }
        }
        accum = fmaf(cc[0], gout[0], fmaf(cc[1], gout[1], cc[2] * gout[2]));  // :1800-1802
    }

    // prologue over this ray's sample cache (:1756-1793); entries >= n are zeros
    const int n = (FUSED && valid && M > 0) ? min(cache_n[q], M) : 0;
    auto AL = [&](int i) -> float { return (i < n) ? cache_a[(size_t)i * Q + q] : 0.f; };
    auto WE = [&](int i) -> float { return (i < n) ? cache_w[(size_t)i * Q + q] : 0.f; };
    auto TS = [&](int i) -> float { return (i < n) ? cache_t[(size_t)i * Q + q] : 0.f; };
    float sample_alpha_sum = 0.f, sample_weight_sum = 0.f;
    for (int i = 0; i < n; ++i) {
        sample_alpha_sum += AL(i);
        sample_weight_sum += WE(i);
    }
    sample_alpha_sum = clamp_lo(sample_alpha_sum);
    sample_weight_sum = clamp_lo(sample_weight_sum);
    float Den_Dasum = 0.f, Den_Dwsum = 0.f, sample_t_mean = 0.f;
    int valid_sample_n = 0;
    float shared_Dmeant_sign = 0.f;
    int max_sample_id = 0;
    {
        float max_weight = 0.f;
        for (int i = 0; i < n; ++i) {
            const float a = AL(i), ww = WE(i);
            Den_Dasum += a * (log_ratio(a, sample_alpha_sum) + 1.f) / sqrf(sample_alpha_sum);
            Den_Dwsum += ww * (log_ratio(ww, sample_weight_sum) + 1.f) / sqrf(sample_weight_sum);
            sample_t_mean += ww / sample_weight_sum * TS(i);
            if (ww > max_weight) {
                max_weight = ww;
                max_sample_id = i;
            }
        }
        for (int i = 0; i < n; ++i) {
            const float ts = TS(i);
            if (ts > 0.f) {
                valid_sample_n++;
                shared_Dmeant_sign += (sample_t_mean > ts) ? 1.f : ((sample_t_mean < ts) ? -1.f : 0.f);
            }
        }
    }

    float log_transmit = 0.f;
    int sample_i = 0;
    int intersect_i = -1;
    bool active = valid && !(r.tmin > r.tmax);
    Walk w;
    if (active) walk_init(g, r, w);

    // l_dist / l_entropy terms of d/d(rwalpha) (:2141-2210, 2602-2666)
    auto aux_rwalpha = [&](float rwalpha, float& curr) {
        const float denom = fminf(rwalpha - 1.f, -1e-8f);
        if (L.lambda_l_dist > 0.f) {
            float Dl = 0.f;
            float log_Tk = log_transmit;
            for (int k = sample_i; k < valid_sample_n; ++k) {
                float Dwk = 0.f;
                const float tk = TS(k);
                for (int j = 0; j < n; ++j) Dwk += WE(j) * fabsf(tk - TS(j));
                if (k == sample_i) {
                    Dl += Dwk * expf(log_transmit);
                } else {
                    log_Tk += logf(clamp_lo(1.f - AL(k - 1)));
                    Dl += Dwk * expf(log_Tk) * AL(k) / denom;
                }
            }
            curr += L.lambda_l_dist * Dl;
        }
        if (L.lambda_l_entropy > 0.f) {
            auto Den_Dw = [&](int k) -> float {
                if (L.no_norm_weight_l_entropy) return -(logf(clamp_lo(WE(k))) + 1.f);
                return -(log_ratio(WE(k), sample_weight_sum) + 1.f) / sample_weight_sum + Den_Dwsum;
            };
            float Den_Dai = Den_Dw(sample_i) * expf(log_transmit);
            float log_Tk = log_transmit;
            for (int k = sample_i + 1; k < valid_sample_n; ++k) {
                const float Dwk = Den_Dw(k);
                log_Tk += logf(clamp_lo(1.f - AL(k - 1)));
                Den_Dai += Dwk * expf(log_Tk) * AL(k) / denom;
            }
            curr += L.lambda_l_entropy * Den_Dai;
        }
    };
    auto sparsity_term = [&](float curr, float a, float weight) -> float {  // :2286-2300
        const float _1_a = clamp_lo(1.f - a);
        const double add = -(double)L.sparsity_loss * (1.0 / fmin((double)(_1_a * logf(_1_a)), -1e-8)) *
                           (double)(1.f - weight / sample_weight_sum);
        return (float)((double)curr + add);
    };

    constexpr int QD = REC ? 1 : kQueueDepth;
    __shared__ int s_queue[QD * kBlock];
    int qn = 0;
    // record replay state (REC)
    int rec_i = 0;
    int n_rec = (REC && valid) ? rec_n[q] : 0;
    int rec_meta = 0;
    bool rec_last_of_voxel = true;
    while (true) {
        if constexpr (REC) {
            qn = 0;
            if (rec_i < n_rec) {
                const int pv = rec_v[(size_t)rec_i * Q + q];
                s_queue[tid] = pv;
                rec_meta = rec_m[(size_t)rec_i * Q + q];
                ++rec_i;
                rec_last_of_voxel = (rec_i >= n_rec) || (rec_v[(size_t)rec_i * Q + q] != pv);
                qn = 1;
            }
        } else {
            while (active && qn < QD) {
                walk_step(g, r, w);
                const bool hit = walk_test<CLDS>(g, s_coarse, w);
                active = (w.t <= r.tmax);
                // backward-only loop quirk: `t += step_size` on voxels with a missing corner
                // (:1935-1936) can end the loop one voxel early near the exit face.
                if (active && !hit && (w.t + o.step_size > r.tmax)) {
                    if (!links_present(g, w)) active = false;
                }
                if (hit) {
                    s_queue[qn * kBlock + tid] = pack_voxel(w.vx, w.vy, w.vz);
                    ++qn;
                }
            }
            if (ASURF_AID(o, 1)) qn = 0;  // measurement aid: traversal only
        }
        if (!__any(qn > 0)) break;
      for (int qi = 0; __any(qi < qn); ++qi) {
        bool found = qi < qn;
        Vox v;
        if (found) load_voxel(g, r, s_queue[qi * kBlock + tid], v);
        if (!REC && o.lazy_gates && found && voxel_all_below(v, o.sigma_thresh)) found = false;   // (as k_forward)
        bool vox_has_sample = false, vox_has_surf = false;
        // record fields: bit0 fake, bits1-2 root index, bits3-7 level-set index,
        // bits8.. intersection ordinal + 1 at the sample
        const bool rec_fake = REC && (rec_meta & 1);
        const int rec_st = (rec_meta >> 1) & 3;
        const int rec_li = (rec_meta >> 3) & 31;
        if (REC && found) intersect_i = (rec_meta >> 8) - 1 - (rec_fake ? 0 : 1);

        for (int li = 0; li < g.level_set_num; ++li) {
            const float lvf = g.level_set[li];
            bool lv_ok = found && !(lvf < v.smin || lvf > v.smax);
            if (REC) lv_ok = lv_ok && !rec_fake && (li == rec_li);
            if (!__any(lv_ok)) continue;
            double st[3] = {-1, -1, -1};
            double fs[4] = {0, 0, 0, 0};
            int root_type = CUBIC_NO_ROOT;
            if (lv_ok) {
                vox_has_surf = true;
                fs[0] = v.fs[0] - (double)lvf;
                fs[1] = v.fs[1];
                fs[2] = v.fs[2];
                fs[3] = v.fs[3];
                root_type = solve_cubic_vieta(fs[0], fs[1], fs[2], fs[3], st);
            }
#pragma unroll
            for (int st_id = 0; st_id < 3; ++st_id) {
                bool cand = lv_ok && (st[st_id] > 0);
                if (REC) cand = cand && (st_id == rec_st);
                float pos[3] = {0.f, 0.f, 0.f};
                if (cand) cand = root_position(r, v, st[st_id], pos);
                if (cand) vox_has_sample = true;
                if (cand && o.only_outward_intersect) {
                    float sg[3];
                    field_grad(v.surf, pos, sg);
                    const float ndd = -(sg[0] * r.d[0] + sg[1] * r.d[1] + sg[2] * r.d[2]);
                    if (ndd >= 0.f) cand = false;
                }
                bool shade = false;
                float raw_alpha = 0.f, alpha = 0.f, trunc_reweight = 1.f, rwalpha = 0.f, pcnt = 0.f,
                      weight = 0.f;
                if (cand) {
                    ++intersect_i;
                    raw_alpha = trilerp8(v.dens, pos);
                    if (raw_alpha > o.sigma_thresh) {
                        alpha = alpha_act(raw_alpha, o.alpha_activation_type);
                        trunc_reweight = o.truncated_vol_render
                                             ? trunc_rw((float)intersect_i, g.trunc_a, o.trunc_vol_weight_min)
                                             : 1.f;
                        rwalpha = alpha * trunc_reweight;
                        pcnt = -logf(clamp_lo(1.f - rwalpha));
                        weight = expf(log_transmit) * (1.f - expf(-pcnt));
                        shade = true;
#pragma unroll
                        for (int i = 0; i < 8; ++i) s_rec[tid].link[i] = v.lk[i];
                    }
                }
                const unsigned long long need = __ballot(shade);
                if (!need) continue;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                float total_color = 0.f;
                float grad_xyz[3] = {0.f, 0.f, 0.f};
                coop_shade_bwd<B>(g, G, !o.no_surf_grad_from_sh, need, rec_wave, sph_wave, pos, weight,
                                  gout, &total_color, grad_xyz);
                if (!shade) continue;
                accum -= weight * total_color;
                float curr_grad_rwalpha =
                    accum / fminf(rwalpha - 1.f, -1e-8f) + total_color * expf(log_transmit);
                aux_rwalpha(rwalpha, curr_grad_rwalpha);
                log_transmit -= pcnt;

                // the reference's lane-0 section (:2266-2448), here per ray
                const float ts_i = TS(sample_i);
                if (L.lambda_l_dist_a > 0.f) {
                    float ga = 0.f;
                    for (int j = 0; j < n; ++j) ga += AL(j) * fabsf(ts_i - TS(j));
                    curr_grad_rwalpha += L.lambda_l_dist_a * ga;
                }
                if (L.lambda_l_entropy_a > 0.f) {
                    const float Den_Dai = -(log_ratio(AL(sample_i), sample_alpha_sum) + 1.f) / sample_alpha_sum;
                    curr_grad_rwalpha += L.lambda_l_entropy_a * (Den_Dai + Den_Dasum);
                }
                if (L.sparsity_loss > 0.f && raw_alpha > 0.f)
                    curr_grad_rwalpha = sparsity_term(curr_grad_rwalpha, alpha, weight);
                if (L.lambda_inwards_norm_loss > 0.f) {  // :2302-2323
                    float sg[3];
                    field_grad(v.surf, pos, sg);
                    const float nn = clamp_lo(sqrtf(sqrf(sg[0]) + sqrf(sg[1]) + sqrf(sg[2])));
                    const float nd = (-sg[0] / nn) * r.d[0] + (-sg[1] / nn) * r.d[1] + (-sg[2] / nn) * r.d[2];
                    if (nd > 0.f) curr_grad_rwalpha += L.lambda_inwards_norm_loss * sqrf(nd);
                }
                const float curr_grad_alpha = curr_grad_rwalpha * trunc_reweight;
                float curr_grad_raw_alpha = curr_grad_alpha * alpha_act_grad(alpha, o.alpha_activation_type);
                {
                    float cw[8];
                    corner_weights(pos, curr_grad_raw_alpha, cw);
                    scatter8(G.grad_density, G.mask, G.mask2, v.lk, cw);
                }
                if (L.lambda_l_di > 0.f && alpha < L.l_di_alpha_thresh)
                    curr_grad_raw_alpha += L.lambda_l_di * (-1.f) * alpha_act_grad(alpha, o.alpha_activation_type);
                trilerp_dpos(v.dens, pos, curr_grad_raw_alpha, grad_xyz);
                float grad_st = grad_xyz[0] * r.d[0] + grad_xyz[1] * r.d[1] + grad_xyz[2] * r.d[2];
                if (L.lambda_l_dist > 0.f || L.lambda_l_dist_a > 0.f) {
                    float gw = 0.f, ga = 0.f;
                    const float wi = WE(sample_i), ai = AL(sample_i);
                    for (int j = 0; j < n; ++j) {
                        const float tj = TS(j);
                        const float sgn = (ts_i > tj) ? 1.f : (ts_i < tj) ? -1.f : 0.f;
                        gw += sgn * wi * WE(j);
                        ga += sgn * ai * AL(j);
                    }
                    grad_st += L.lambda_l_dist * gw + L.lambda_l_dist_a * ga;
                }
                if (L.lambda_l_samp_dist > 0.f) {
                    const float sgn = (sample_t_mean > ts_i) ? 1.f : (sample_t_mean < ts_i) ? -1.f : 0.f;
                    grad_st += L.lambda_l_samp_dist *
                               (shared_Dmeant_sign * AL(sample_i) / sample_weight_sum + sgn * (-1.f));
                }
                if (L.lambda_conv_mode_samp > 0.f && trunc_reweight > o.trunc_vol_weight_min) {
                    const float tm = TS(max_sample_id);
                    const float gcm = (ts_i > tm) ? 1.f : (ts_i < tm) ? -1.f : 0.f;
                    grad_st += L.lambda_conv_mode_samp * gcm;
                }
                float grad_fs[4] = {grad_st, grad_st, grad_st, grad_st};
                cubic_root_grad_vieta(root_type, st_id, fs, grad_fs);
                float gsurf[8];
                const float nof[3] = {(float)((double)v.new_origin[0] - (double)v.vx),
                                      (float)((double)v.new_origin[1] - (double)v.vy),
                                      (float)((double)v.new_origin[2] - (double)v.vz)};
                surface_cubic_grad(nof, r.d, grad_fs, gsurf);
                scatter8(G.grad_surface, G.mask, G.mask2, v.lk, gsurf);
                if (alpha < L.surf_sparse_alpha_thresh) {  // :2436-2445
                    float cw[8];
                    corner_weights(pos, L.lambda_inplace_surf_sparse, cw);
                    scatter8(G.grad_surface, G.mask, G.mask2, v.lk, cw);
                }
                if (sample_i < M - 1) sample_i += 1;
            }
        }

        // fake-sample gradient (:2460-2866)
        {
            bool fk = found && o.surf_fake_sample && !vox_has_sample &&
                      (!o.limited_fake_sample || vox_has_surf) &&
                      ((v.t_far - v.t_close) > o.surf_fake_sample_min_vox_len);
            if (REC) fk = found && rec_fake;  // the forward already decided
            float pos[3] = {0.f, 0.f, 0.f};
            float raw_alpha = 0.f;
            if (fk) {
                fake_position(r, v, pos);
                raw_alpha = trilerp8(v.dens, pos);
                fk = raw_alpha > o.sigma_thresh;
            }
            float alpha = 0.f, reweight = 0.f, trunc_reweight = 1.f, rw_alpha = 0.f, pcnt = 0.f,
                  weight = 0.f, dist = 0.f;
            double miu = 0., sd = 1.;
            if (fk) {
                alpha = alpha_act(raw_alpha, o.alpha_activation_type);
                dist = fake_distance(g, o, v, pos, &miu, &sd);
                reweight = expf((float)(-.5 * (double)sqrf(dist / g.fake_sample_std)));
                trunc_reweight = o.truncated_vol_render
                                     ? trunc_rw((float)intersect_i, g.trunc_a, o.trunc_vol_weight_min)
                                     : 1.f;
                rw_alpha = alpha * reweight * trunc_reweight;
                pcnt = -1 * logf(clamp_lo(1.f - rw_alpha));
                weight = expf(log_transmit) * (1.f - expf(-pcnt));
#pragma unroll
                for (int i = 0; i < 8; ++i) s_rec[tid].link[i] = v.lk[i];
            }
            const unsigned long long need = __ballot(fk);
            if (need) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                float total_color = 0.f;
                float gx_unused[3] = {0.f, 0.f, 0.f};
                // fake samples carry no d/dpos (:2722)
                coop_shade_bwd<B>(g, G, false, need, rec_wave, sph_wave, pos, weight, gout,
                                  &total_color, gx_unused);
                if (fk) {
                    accum -= weight * total_color;
                    float curr_grad_rwalpha =
                        accum / fminf(rw_alpha - 1.f, -1e-8f) + total_color * expf(log_transmit);
                    if (o.fake_sample_l_dist) aux_rwalpha(rw_alpha, curr_grad_rwalpha);
                    log_transmit -= pcnt;
                    if (o.fake_sample_l_dist) {
                        if (L.lambda_l_dist_a > 0.f) {
                            float ga = 0.f;
                            const float ts_i = TS(sample_i);
                            for (int j = 0; j < n; ++j) ga += AL(j) * fabsf(ts_i - TS(j));
                            curr_grad_rwalpha += L.lambda_l_dist_a * ga;
                        }
                        if (L.lambda_l_entropy_a > 0.f) {
                            const float Den_Dai =
                                -(log_ratio(AL(sample_i), sample_alpha_sum) + 1.f) / sample_alpha_sum;
                            curr_grad_rwalpha += L.lambda_l_entropy_a * (Den_Dai + Den_Dasum);
                        }
                        if (sample_i < M - 1) sample_i += 1;
                    }
                    if (L.sparsity_loss > 0.f && raw_alpha > 0.f)
                        curr_grad_rwalpha = sparsity_term(curr_grad_rwalpha, rw_alpha, weight);
                    if (L.lambda_inwards_norm_loss > 0.f) {
                        float sg[3];
                        field_grad(v.surf, pos, sg);
                        const float nn = clamp_lo(sqrtf(sqrf(sg[0]) + sqrf(sg[1]) + sqrf(sg[2])));
                        const float nd = (-sg[0] / nn) * r.d[0] + (-sg[1] / nn) * r.d[1] + (-sg[2] / nn) * r.d[2];
                        if (nd > 0.f) curr_grad_rwalpha += L.lambda_inwards_norm_loss * sqrf(nd);
                    }
                    const float curr_grad_alpha = curr_grad_rwalpha * reweight * trunc_reweight;
                    const float curr_grad_raw_alpha =
                        curr_grad_alpha * alpha_act_grad(alpha, o.alpha_activation_type);
                    {
                        float cw[8];
                        corner_weights(pos, curr_grad_raw_alpha, cw);
                        scatter8(G.grad_density, G.mask, G.mask2, v.lk, cw);
                    }
                    // surface gradient through the gaussian re-weighting (:2792-2851)
                    const float grad_fake_dist =
                        curr_grad_rwalpha * (-alpha * trunc_reweight * dist * reweight / sqrf(g.fake_sample_std));
                    float grad_ns[8];
                    corner_weights(pos, grad_fake_dist, grad_ns);
                    float gsurf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                    if (!o.fake_sample_normalize_surf) {
#pragma unroll
                        for (int ks = 0; ks < 8; ++ks) gsurf[ks] = grad_ns[ks];
                    } else {
                        const double std3 = cubd(sd);
#pragma unroll
                        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                            for (int kn = 0; kn < 8; ++kn) {
                                double jac = (double)v.surf[kn] * (miu - (double)v.surf[ks]) / 8.f / std3;
                                if (ks == kn) jac += 1.f / sd;
                                gsurf[ks] = (float)((double)gsurf[ks] + (double)grad_ns[kn] * jac);
                            }
                        }
                    }
                    scatter8(G.grad_surface, G.mask, G.mask2, v.lk, gsurf);
                    if (G.grad_fake_sample_std != nullptr) {
                        const float gstd = curr_grad_rwalpha * alpha * sqrf(dist) * reweight * trunc_reweight /
                                           (g.fake_sample_std * g.fake_sample_std * g.fake_sample_std);
                        atomicAdd(G.grad_fake_sample_std, gstd);
                    }
                }
            }
        }
        // the reference tests this once per voxel, after all of its samples (:2893-2895)
        if (found && (!REC || rec_last_of_voxel) && expf(log_transmit) < o.stop_thresh) {
            active = false;
            qn = 0;
            n_rec = 0;
        }
      }
        qn = 0;
    }
}

// test_cubic_root_grad_kernel (test_cuda.cu:32-106), one row per thread
__global__ void k_test_cubic_root_grad(const double* __restrict__ fs, const int* __restrict__ st_ids,
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
    float g[4] = {grad_fs[4 * i], grad_fs[4 * i + 1], grad_fs[4 * i + 2], grad_fs[4 * i + 3]};
    if (type == CUBIC_NO_ROOT || st_ids[i] >= root_num) {
        g[0] = g[1] = g[2] = g[3] = 0.f;
    } else {
        cubic_root_grad_vieta(type, st_ids[i], f, g);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) grad_fs[4 * i + j] = g[j];
}

}  // namespace asurf
