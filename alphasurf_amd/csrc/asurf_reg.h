// Per-iteration regularisers of the surface grid: one thread per
// (listed cell, channel), forward-difference stencils on the link grid, gradients added straight
// into the parameter gradients with atomics, touched rows marked in mask_out.
//   k_tv_grad_sparse               loss_kernel.cu:739-808   (TV on raw opacity / any [N,C] plane)
//   k_surf_tv_grad_sparse          loss_kernel.cu:810-895   (TV on the surface scalar)
//   k_tv_value                     loss_kernel.cu:73-117    (the dense TV value, a reduction)
//   k_surface_normal_grad_sparse   loss_kernel.cu:398-436 + render_util.cuh:1823-2130
//   k_alpha_surf_sparsify          loss_kernel.cu:665-738
//   k_surface_eikonal_grad_sparse  the reference evaluates this loss with PyTorch autograd
//                                  (svox2.py:5433-5472); same forward differences, analytic gradient
// All of them are streaming kernels over rand_cells: HBM-bound, no reuse to stage.
#pragma once
#include "asurf_device.h"

namespace asurf {

struct RegGrid {
    const int32_t* __restrict__ links;
    int size[3];
    int offx, offy;
};

__device__ __forceinline__ void reg_decode(const RegGrid& G, int xyz, int& x, int& y, int& z) {
    z = xyz % G.size[2];
    const int xy = xyz / G.size[2];
    y = xy % G.size[1];
    x = xy / G.size[1];
}

__device__ __forceinline__ void reg_add(float* grad, uint8_t* mask, int lnk, int C, int idx, float val,
                                        float idelta) {
    if (lnk >= 0 && val != 0.f) {
        atomicAdd(&grad[(size_t)lnk * C + idx], val * idelta);
        if (mask != nullptr) mask[lnk] = 1;
    }
}

// A block's gradient terms, summed in LDS before they leave (round 4).  The cells of a block are neighbours in storage
// order -- runs along z, column after column -- and every vertex row is a corner of up to eight of them: written straight
// out, the normal loss issued 1.22 M atomic line requests for the bench scene's 1.31 M cells and the surface TV 0.38 M,
// both at 16 G requests/s of the 21 G the chip takes (75 and 25 us).  The table is direct-mapped by element number (rows of
// a z-run are numbered one after the other, so a run fills consecutive slots); a term whose slot belongs to another
// element leaves by itself, as before.  Flushed slot by slot after a barrier: consecutive slots are consecutive
// addresses, an instruction's 64 terms mostly one or two lines.
constexpr int kRegAccSlots = 2048;
struct RegAcc {
    float acc[kRegAccSlots];
    int tag[kRegAccSlots];
};
__device__ __forceinline__ void reg_acc_init(RegAcc& A) {
    for (int i = threadIdx.x; i < kRegAccSlots; i += blockDim.x) {
        A.acc[i] = 0.f;
        A.tag[i] = -1;
    }
    __syncthreads();
}
// elem = row * C + channel (C = 1: the row)
__device__ __forceinline__ void reg_acc_add(RegAcc& A, float* grad, uint8_t* mask, long long elem_ll, int C, float val) {
    const int elem = (int)elem_ll;
    const int slot = elem & (kRegAccSlots - 1);
    int old = elem_ll > 0x7fffffffll ? -2 : A.tag[slot];   // (an element number beyond 31 bits never enters the table)
    if (old != elem) {
        if (old != -2) old = atomicCAS(&A.tag[slot], -1, elem);
        if (old != -1 && old != elem) {
            atomicAdd(&grad[elem_ll], val);
            if (mask != nullptr) mask[C == 1 ? elem_ll : elem_ll / C] = 1;
            return;
        }
    }
    atomicAdd(&A.acc[slot], val);
}
__device__ __forceinline__ void reg_acc_flush(RegAcc& A, float* grad, uint8_t* mask, int C) {
    __syncthreads();
    for (int i = threadIdx.x; i < kRegAccSlots; i += blockDim.x) {
        const int elem = A.tag[i];
        if (elem >= 0) {
            atomicAdd(&grad[elem], A.acc[i]);
            if (mask != nullptr) mask[C == 1 ? elem : elem / C] = 1;
        }
    }
}

// SURF = false: tv_grad_sparse_kernel; SURF = true: surf_tv_grad_sparse_kernel (edge_value for
// missing neighbours, optional boost where all four raw opacities are low).
template <bool SURF>
__global__ void __launch_bounds__(256)
k_tv_grad_sparse(RegGrid G, const float* __restrict__ data, const float* __restrict__ density, int C,
                 int density_C, const int32_t* __restrict__ rand_cells, int start_dim, int end_dim,
                 float scale, long long Q, int ignore_edge, float edge_value, int ignore_last_z,
                 int alpha_dependency, uint8_t* __restrict__ mask_out, float* __restrict__ grad_data) {
    // SURF = false -- the TV on raw opacity, which the training script runs over a run of 1 % of ALL cells, most of them
    // empty -- sums its block's terms in LDS first (RegAcc): 19 -> 12 us for the bench scene's 167 k cells, the few
    // lanes that hold a term no longer issue eight memory instructions each.  The surface TV runs over the non-empty
    // cells: every lane adds four terms, an instruction's 64 rows are neighbours already, and the table's flush only
    // spreads them over more instructions (25 -> 42 us): it adds straight to memory.
    constexpr bool kTable = !SURF;
    __shared__ RegAcc A;
    if constexpr (kTable) reg_acc_init(A);
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nd = end_dim - start_dim;
    int lk4[4] = {-1, -1, -1, -1};   // (!kTable) the four rows and their terms, added below
    float tv4[4] = {0.f, 0.f, 0.f, 0.f};
    int idx_out = 0;
    [&]() {
        if (tid >= Q) return;
        const int idx = (int)(tid % nd) + start_dim;
        idx_out = idx;
        int x, y, z;
        reg_decode(G, rand_cells[tid / nd], x, y, z);
        const int32_t* lp = G.links + ((size_t)x * G.offx + (size_t)y * G.offy + z);
        if (ignore_edge && *lp == 0) return;
        const float sc0 = G.size[0] * (1.f / 256.f), sc1 = G.size[1] * (1.f / 256.f),
                    sc2 = G.size[2] * (1.f / 256.f);  // calculate_ray_scale, :23-59
        // a neighbour outside the grid counts as link 0, as in the reference (:765-770)
        const int l000 = lp[0];
        const int l001 = ((z + 1 < G.size[2]) && (!ignore_last_z || z != G.size[2] - 2)) ? lp[1] : 0;
        const int l010 = (y + 1 < G.size[1]) ? lp[G.offy] : 0;
        const int l100 = (x + 1 < G.size[0]) ? lp[G.offx] : 0;
        if (ignore_last_z && z == G.size[2] - 2) return;
        const float dflt = SURF ? edge_value : 0.f;
        const float v000 = l000 >= 0 ? data[(size_t)l000 * C + idx] : dflt;
        const float null_val = ignore_edge ? v000 : dflt;
        const float v001 = l001 >= 0 ? data[(size_t)l001 * C + idx] : null_val;
        const float v010 = l010 >= 0 ? data[(size_t)l010 * C + idx] : null_val;
        const float v100 = l100 >= 0 ? data[(size_t)l100 * C + idx] : null_val;
        float dx = v100 - v000, dy = v010 - v000, dz = v001 - v000;
        float idelta = scale * rsqrtf(1e-9f + dx * dx + dy * dy + dz * dz);
        if constexpr (SURF) {
            if (alpha_dependency) {  // :866-875
                const float a0 = l000 >= 0 ? density[(size_t)l000 * density_C + idx] : 0.f;
                const float a1 = l001 >= 0 ? density[(size_t)l001 * density_C + idx] : 0.f;
                const float a2 = l010 >= 0 ? density[(size_t)l010 * density_C + idx] : 0.f;
                const float a3 = l100 >= 0 ? density[(size_t)l100 * density_C + idx] : 0.f;
                const float mx = fmaxf(a0, fmaxf(a1, fmaxf(a2, a3)));
                if (mx < 0.1f) idelta /= fmaxf(mx * 10, 1e-1f);
            }
        }
        dx *= sc0;
        dy *= sc1;
        dz *= sc2;
        const int lks[4] = {l000, l001, l010, l100};
        const float vals[4] = {-(dx + dy + dz), dz, dy, dx};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (kTable) {
                if (lks[j] >= 0 && vals[j] != 0.f)
                    reg_acc_add(A, grad_data, mask_out, (long long)lks[j] * C + idx, C, vals[j] * idelta);
            } else if (lks[j] >= 0 && vals[j] != 0.f) {
                lk4[j] = lks[j];
                tv4[j] = vals[j] * idelta;
            }
        }
    }();
    if constexpr (kTable) {
        reg_acc_flush(A, grad_data, mask_out, C);
    } else {
        // Cells mostly follow each other along z, a lane a cell: the z term of the cell below lands on THIS lane's own
        // row and leaves with its own term -- three atomic instructions a wave instead of four (the surface TV was at
        // 16 G atomic requests/s of the chip's 21).
        const int lane = (int)(threadIdx.x & 63);
        const int up_l = __shfl_up(lk4[1], 1);
        const float up_t = __shfl_up(tv4[1], 1);
        const int dn_l = __shfl_down(lk4[0], 1);
        const int own0 = __shfl_down(lk4[1], 0);   // (lk4[1] itself; keeps the three shuffles together)
        // the receiver must be about to add to that row anyway (its own term exists): else the sender keeps its term
        const bool recv = nd == 1 && lane >= 1 && lk4[0] >= 0 && up_l == lk4[0];
        const bool sent = nd == 1 && lane <= 62 && own0 >= 0 && dn_l == own0;
        if (recv) tv4[0] += up_t;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (lk4[j] >= 0 && !(j == 1 && sent)) {
                atomicAdd(&grad_data[(size_t)lk4[j] * C + idx_out], tv4[j]);
                if (mask_out != nullptr) mask_out[lk4[j]] = 1;
            }
        }
    }
}

// tv_kernel (loss_kernel.cu:73-117): the TV *value* over every cell with x < X-1, y < Y-1, z < Z-1 --
// sum over (cell, channel) of sqrt(1e-5 + |scaled forward differences|^2) * scale.  One wave-level
// DPP sum, one LDS pass over the block's four waves, one atomic per block.
__global__ void __launch_bounds__(256)
k_tv_value(RegGrid G, const float* __restrict__ data, int C, int start_dim, int end_dim, float scale,
           long long Q, int ignore_edge, float* __restrict__ out) {
    __shared__ float s_part[4];
    const int nd = end_dim - start_dim;
    const float sc0 = G.size[0] * (1.f / 256.f), sc1 = G.size[1] * (1.f / 256.f),
                sc2 = G.size[2] * (1.f / 256.f);
    float acc = 0.f;
    for (long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x; tid < Q;
         tid += (long long)gridDim.x * blockDim.x) {
        const int idx = (int)(tid % nd) + start_dim;
        const long long xyz = tid / nd;
        const int z = (int)(xyz % (G.size[2] - 1));
        const long long xy = xyz / (G.size[2] - 1);
        const int y = (int)(xy % (G.size[1] - 1));
        const int x = (int)(xy / (G.size[1] - 1));
        const int32_t* lp = G.links + ((size_t)x * G.offx + (size_t)y * G.offy + z);
        const int l000 = lp[0];
        if (ignore_edge && l000 == 0) continue;
        const int l001 = lp[1], l010 = lp[G.offy], l100 = lp[G.offx];
        const float v000 = l000 >= 0 ? data[(size_t)l000 * C + idx] : 0.f;
        const float null_val = ignore_edge ? v000 : 0.f;
        const float v100 = l100 >= 0 ? data[(size_t)l100 * C + idx] : null_val;
        const float v010 = l010 >= 0 ? data[(size_t)l010 * C + idx] : null_val;
        const float v001 = l001 >= 0 ? data[(size_t)l001 * C + idx] : null_val;
        const float dx = (v100 - v000) * sc0, dy = (v010 - v000) * sc1, dz = (v001 - v000) * sc2;
        acc += sqrtf(1e-5f + dx * dx + dy * dy + dz * dz);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (s_part[0] + s_part[1] + s_part[2] + s_part[3]) * scale);
}

// ---- surface normal smoothness -------------------------------------------------------------
struct RegCell {  // the 8 corner rows of a cell and their surface values
    int lk[8];
    float s[8];
    bool exist;
};

__device__ __forceinline__ void reg_cell(const RegGrid& G, const float* __restrict__ surface, int x, int y,
                                         int z, RegCell& c) {
    c.exist = (x < G.size[0] - 1) && (y < G.size[1] - 1) && (z < G.size[2] - 1);
    if (!c.exist) return;
    const int32_t* p = G.links + ((size_t)x * G.offx + (size_t)y * G.offy + z);
    c.lk[0] = p[0];
    c.lk[1] = p[1];
    c.lk[2] = p[G.offy];
    c.lk[3] = p[G.offy + 1];
    c.lk[4] = p[G.offx];
    c.lk[5] = p[G.offx + 1];
    c.lk[6] = p[G.offx + G.offy];
    c.lk[7] = p[G.offx + G.offy + 1];
    int any = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) any |= c.lk[i];
    c.exist = any >= 0;
    if (!c.exist) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) c.s[i] = surface[c.lk[i]];
}

// unnormalised normal: mean forward difference over the cell's faces (__CALC_DX/DY/DZ)
__device__ __forceinline__ void reg_normal(const RegCell& c, float* n) {
    n[0] = ((c.s[4] + c.s[5] + c.s[6] + c.s[7]) - (c.s[0] + c.s[1] + c.s[2] + c.s[3])) / 4;
    n[1] = ((c.s[2] + c.s[3] + c.s[6] + c.s[7]) - (c.s[0] + c.s[1] + c.s[4] + c.s[5])) / 4;
    n[2] = ((c.s[1] + c.s[3] + c.s[5] + c.s[7]) - (c.s[0] + c.s[2] + c.s[4] + c.s[6])) / 4;
}

__device__ __forceinline__ bool reg_all_one_side(const float* s, int n, float lv) {
    bool le = true, ge = true;
    for (int i = 0; i < n; ++i) {
        le = le && (s[i] <= lv);
        ge = ge && (s[i] >= lv);
    }
    return le || ge;
}

// The neighbour of c0 along axis i (bit = 4 >> i): its low face is c0's high face, so only the four corners
// of its high face are loaded.
__device__ __forceinline__ void reg_cell_next(const RegGrid& G, const float* __restrict__ surface, int x, int y,
                                              int z, int bit, const RegCell& c0, RegCell& c) {
    c.exist = (x < G.size[0] - 1) && (y < G.size[1] - 1) && (z < G.size[2] - 1);
    if (!c.exist) return;
    const int32_t* p = G.links + ((size_t)x * G.offx + (size_t)y * G.offy + z);
    const int off[8] = {0, 1, G.offy, G.offy + 1, G.offx, G.offx + 1, G.offx + G.offy, G.offx + G.offy + 1};
    int any = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        c.lk[j] = (j & bit) ? p[off[j]] : c0.lk[j | bit];
        any |= c.lk[j];
    }
    c.exist = any >= 0;
    if (!c.exist) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) c.s[j] = (j & bit) ? surface[c.lk[j]] : c0.s[j | bit];
}

// d(normal component a) / d(corner): +-1/4 by the corner's bit on that axis (_split_add_surface_norm_grad)
__device__ __forceinline__ void reg_split_add(const RegCell& c, const float* gn, float scale,
                                              uint8_t* mask, float* grad) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float g = ((i & 4) ? 0.25f : -0.25f) * gn[0] + ((i & 2) ? 0.25f : -0.25f) * gn[1] +
                        ((i & 1) ? 0.25f : -0.25f) * gn[2];
        const float val = scale * g;
        if (val != 0.f) {
            atomicAdd(&grad[c.lk[i]], val);
            if (mask != nullptr) mask[c.lk[i]] = 1;
        }
    }
}

__global__ void __launch_bounds__(256)
k_surface_normal_grad_sparse(RegGrid G, const float* __restrict__ surface,
                             const int32_t* __restrict__ rand_cells, float lv_set, float scale, long long Q,
                             int con_check, int ignore_empty, int use_l1, uint8_t* __restrict__ mask_out,
                             float* __restrict__ grad_data) {
    __shared__ RegAcc A;
    reg_acc_init(A);
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = tid < Q;
    const int cell = in_range ? rand_cells[tid] : -0x40000000;   // (never next to a listed cell)
    int x = 0, y = 0, z = 0;
    RegCell c0;
    c0.exist = false;
    RegCell cn[3];
    bool active = false;   // the cell exists and has a neighbour that counts
    // The reference issues 16 atomics per used neighbour (8 rows of c0, 8 of the neighbour).  The rows of c0,
    // and the neighbour's low face (which IS c0's high face), are summed in registers and leave once: at
    // most 8 + 3 * 4 atomics per cell instead of 48; a row is marked touched when any of its terms was non-zero.
    float acc0[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    unsigned nz0 = 0u;                       // bit j: some term of acc0[j] was non-zero
    float far_z[4] = {0.f, 0.f, 0.f, 0.f};   // the z neighbour's high face (rows at z + 2), corner (jj << 1) | 1
    unsigned nz_far = 0u;
    if (in_range) {
        reg_decode(G, cell, x, y, z);
        reg_cell(G, surface, x, y, z, c0);
    }
    if (c0.exist) {
        const bool empty0 = ignore_empty ? reg_all_one_side(c0.s, 8, lv_set) : false;
        float n0[3];
        reg_normal(c0, n0);
        // neighbours along x (i = 0), y (1), z (2); the face shared with c0 decides connectivity
        bool use[3];
        int count = 0;
        // The three neighbours are fetched TOGETHER, links first, then values (round 4): one neighbour after the
        // other -- test, links, values, test, ... -- was a chain of nine dependent round trips a cell at five waves a
        // SIMD, 77 us for the 1.3 M cells of the bench scene and the longest kernel of a training iteration.
        bool want[3];
        const int off[8] = {0, 1, G.offy, G.offy + 1, G.offx, G.offx + 1, G.offx + G.offy, G.offx + G.offy + 1};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int bit = 4 >> i;
            cn[i].exist = false;
            want[i] = true;
            if (con_check) {
                // the neighbour's low face along axis i IS c0's high face: whether the level crosses it is known
                // before anything of the neighbour is loaded, and in a band around a surface most faces hold no crossing
                float f[4];
                int k = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j & bit) f[k++] = c0.s[j];
                want[i] = !reg_all_one_side(f, 4, lv_set);
            }
            const int xn = x + (i == 0), yn = y + (i == 1), zn = z + (i == 2);
            want[i] = want[i] && (xn < G.size[0] - 1) && (yn < G.size[1] - 1) && (zn < G.size[2] - 1);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {   // links of the neighbours' high faces
            const int bit = 4 >> i;
            if (want[i]) {
                const int32_t* p = G.links + ((size_t)(x + (i == 0)) * G.offx + (size_t)(y + (i == 1)) * G.offy + (z + (i == 2)));
#pragma unroll
                for (int j = 0; j < 8; ++j) cn[i].lk[j] = (j & bit) ? p[off[j]] : c0.lk[j | bit];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {   // their values
            const int bit = 4 >> i;
            if (want[i]) {
                int any = 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) any |= cn[i].lk[j];
                cn[i].exist = any >= 0;
                if (cn[i].exist) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) cn[i].s[j] = (j & bit) ? surface[cn[i].lk[j]] : c0.s[j | bit];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            bool ok = cn[i].exist;
            if (ok && ignore_empty) ok = !empty0 || !reg_all_one_side(cn[i].s, 8, lv_set);
            use[i] = ok;
            count += ok ? 1 : 0;
        }
        if (count > 0) {
            active = true;
            const float N0 = sqrtf(1e-9f + n0[0] * n0[0] + n0[1] * n0[1] + n0[2] * n0[2]);  // _NORM3
            const float sc = scale * 1.f / count;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (!use[i]) continue;
                float n1[3];
                reg_normal(cn[i], n1);
                const float N1 = sqrtf(1e-9f + n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
                float w[3];  // dLoss / d(L_b), L = n0/N0 - n1/N1
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float L = n0[b] / N0 - n1[b] / N1;
                    w[b] = use_l1 ? ((L > 0.f) ? 1.f : (L == 0.f ? 0.f : -1.f)) : 2.f * L;
                }
                float d0[3], d1[3];
                const float i0 = 1.f / (N0 * N0 * N0), i1 = 1.f / (N1 * N1 * N1);
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float s0 = 0.f, s1 = 0.f;
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        s0 += w[b] * (((a == b) ? 1.f / N0 : 0.f) - n0[a] * n0[b] * i0);
                        s1 += w[b] * (((a == b) ? -1.f / N1 : 0.f) + n1[a] * n1[b] * i1);
                    }
                    d0[a] = s0;
                    d1[a] = s1;
                }
                const int bit = 4 >> i;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float g0 = ((j & 4) ? 0.25f : -0.25f) * d0[0] + ((j & 2) ? 0.25f : -0.25f) * d0[1] +
                                     ((j & 1) ? 0.25f : -0.25f) * d0[2];
                    const float v0 = sc * g0;
                    if (v0 != 0.f) {
                        acc0[j] += v0;
                        nz0 |= 1u << j;
                    }
                    const float g1 = ((j & 4) ? 0.25f : -0.25f) * d1[0] + ((j & 2) ? 0.25f : -0.25f) * d1[1] +
                                     ((j & 1) ? 0.25f : -0.25f) * d1[2];
                    const float v1 = sc * g1;
                    if (v1 != 0.f) {
                        if (!(j & bit)) {
                            acc0[j | bit] += v1;
                            nz0 |= 1u << (j | bit);
                        } else if (i == 2) {   // (j & 1: the rows at z + 2 -- held back for the merge along z below)
                            far_z[j >> 1] += v1;
                            nz_far |= 1u << (j >> 1);
                        } else {
                            reg_acc_add(A, grad_data, mask_out, cn[i].lk[j], 1, v1);
                        }
                    }
                }
            }
        }
    }
    // Merge along z.  The cells of the list mostly come as runs along z (the non-empty cells in storage order), a
    // lane a cell: the four rows at level z of lane's cell also receive the high-face sums of the cell below (lane - 1)
    // and the far-face sums of the cell two below (lane - 2).  Those travel across the wave and leave with the owner's
    // atomic: 12 atomics a cell inside a run instead of 20 (the normal loss is the longest kernel of an iteration at
    // the training script's batch size).  A lane keeps what nobody takes: the run's last cells, the wave's last lanes.
    const int lane = (int)(threadIdx.x & 63);
    const int cell_m1 = __shfl_up(cell, 1), cell_m2 = __shfl_up(cell, 2);
    const int cell_p1 = __shfl_down(cell, 1), cell_p2 = __shfl_down(cell, 2);
    const int iact = active ? 1 : 0;
    const int act_p1 = __shfl_down(iact, 1), act_p2 = __shfl_down(iact, 2);
    const bool recv1 = active && lane >= 1 && cell_m1 == cell - 1 && z >= 1;
    const bool recv2 = active && lane >= 2 && cell_m2 == cell - 2 && z >= 2;
    const bool send1 = active && lane <= 62 && cell_p1 == cell + 1 && z + 1 < G.size[2] && act_p1 != 0;
    const bool send2 = active && lane <= 61 && cell_p2 == cell + 2 && z + 2 < G.size[2] && act_p2 != 0;
    const unsigned nz_hi = ((nz0 >> 1) & 1u) | ((nz0 >> 2) & 2u) | ((nz0 >> 3) & 4u) | ((nz0 >> 4) & 8u);
    const int flags = (int)(nz_hi | (nz_far << 4));
    const int flags_m1 = __shfl_up(flags, 1), flags_m2 = __shfl_up(flags, 2);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const float from_hi = __shfl_up(acc0[2 * jj + 1], 1);
        const float from_far = __shfl_up(far_z[jj], 2);
        if (recv1) {
            acc0[2 * jj] += from_hi;
            nz0 |= (unsigned)((flags_m1 >> jj) & 1) << (2 * jj);
        }
        if (recv2) {
            acc0[2 * jj] += from_far;
            nz0 |= (unsigned)((flags_m2 >> (4 + jj)) & 1) << (2 * jj);
        }
    }
    if (active) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (((nz0 >> j) & 1u) && (!(j & 1) || !send1)) reg_acc_add(A, grad_data, mask_out, c0.lk[j], 1, acc0[j]);
        }
        if (!send2) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if ((nz_far >> jj) & 1u) reg_acc_add(A, grad_data, mask_out, cn[2].lk[2 * jj + 1], 1, far_z[jj]);
            }
        }
    }
    reg_acc_flush(A, grad_data, mask_out, 1);
}

// ---- log-alpha / surface sparsification, loss_kernel.cu:665-738 ------------------------------
__global__ void __launch_bounds__(256)
k_alpha_surf_sparsify(RegGrid G, const float* __restrict__ alpha, const float* __restrict__ surf,
                      const int32_t* __restrict__ rand_cells, float scale_alpha, float scale_surf,
                      int surf_decrease, float surf_thresh, float alpha_bound, float surf_bound, long long Q,
                      uint8_t* __restrict__ mask_out, float* __restrict__ grad_alpha,
                      float* __restrict__ grad_surf) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= Q) return;
    int x, y, z;
    reg_decode(G, rand_cells[tid], x, y, z);
    const int l = G.links[(size_t)x * G.offx + (size_t)y * G.offy + z];
    if (l < 0) return;
    if (mask_out != nullptr) mask_out[l] = 1;
    const float a = alpha[l];
    const float safe = 1.f / fmaxf(a, 1e-8f);
    if (a > alpha_bound) atomicAdd(&grad_alpha[l], scale_alpha * safe);
    const float s = surf[l];
    const bool reg = surf_decrease ? (s > surf_bound) : (s < surf_bound);
    if (reg && a < surf_thresh) atomicAdd(&grad_surf[l], surf_decrease ? (scale_surf * safe) : (-scale_surf * safe));
}

// ---- eikonal: scaling / n_cells * sum (1 - |forward difference / h|^2)^2, svox2.py:5433-5472 ---
__global__ void __launch_bounds__(256)
k_surface_eikonal_grad_sparse(RegGrid G, const float* __restrict__ surface,
                              const int32_t* __restrict__ rand_cells, float scale, float inv_h2, long long Q,
                              uint8_t* __restrict__ mask_out, float* __restrict__ grad_data,
                              float* __restrict__ loss_out) {
    float loss = 0.f;
    // grid-stride: the loss is reduced per block and added with ONE atomic per block (a single
    // address serialises its atomics; one per wave of a 1.3 M-cell launch costs more than the kernel)
    for (long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x; tid < Q;
         tid += (long long)gridDim.x * blockDim.x) {
        int x, y, z;
        reg_decode(G, rand_cells[tid], x, y, z);
        const int32_t* lp = G.links + ((size_t)x * G.offx + (size_t)y * G.offy + z);
        const int l000 = lp[0];
        if (l000 >= 0) {
            // a missing neighbour takes the centre value: zero difference, no gradient
            const int l100 = (x + 1 < G.size[0]) ? lp[G.offx] : -1;
            const int l010 = (y + 1 < G.size[1]) ? lp[G.offy] : -1;
            const int l001 = (z + 1 < G.size[2]) ? lp[1] : -1;
            const float s0 = surface[l000];
            const float dx = (l100 >= 0 ? surface[l100] : s0) - s0;
            const float dy = (l010 >= 0 ? surface[l010] : s0) - s0;
            const float dz = (l001 >= 0 ? surface[l001] : s0) - s0;
            const float g2 = (dx * dx + dy * dy + dz * dz) * inv_h2;
            loss += scale * (1.f - g2) * (1.f - g2);
            const float k = -4.f * scale * (1.f - g2) * inv_h2;  // d loss / d(difference) / difference
            const float gx = k * dx, gy = k * dy, gz = k * dz;
            if (gx != 0.f) atomicAdd(&grad_data[l100], gx);
            if (gy != 0.f) atomicAdd(&grad_data[l010], gy);
            if (gz != 0.f) atomicAdd(&grad_data[l001], gz);
            const float gc = -(gx + gy + gz);
            if (gc != 0.f) atomicAdd(&grad_data[l000], gc);
            if (mask_out != nullptr) {
                mask_out[l000] = 1;
                if (gx != 0.f) mask_out[l100] = 1;
                if (gy != 0.f) mask_out[l010] = 1;
                if (gz != 0.f) mask_out[l001] = 1;
            }
        }
    }
    if (loss_out != nullptr) {
        __shared__ float s_part[256 / 64];
        const float tot = wave_sum(loss);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = tot;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float b = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
            if (b != 0.f) atomicAdd(loss_out, b);
        }
    }
}

}  // namespace asurf
