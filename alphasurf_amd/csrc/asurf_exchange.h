// asurf_exchange.h -- device side of the multi-GPU gradient exchange (alphasurf_amd/dist.py; no reference counterpart).
//
// The rows a batch touched are a fraction of the grid (a fifth for one 800 x 800 view of the bench scene), and the
// dense exchange moves all of it: 152 MB per rank and step at config 2.  The exchange keyed by rows works on the UNION
// of the ranks' touched rows: the masks are all-reduced (max) first, so every rank holds the same union mask and
// derives the SAME ordered list of rows from it (k_rows_count / k_rows_scan / k_rows_fill: per-block counts, one block
// scanning them, ordered fill -- no atomics, the order must not depend on timing); each rank packs its own values
// of those rows into a dense [U, D + 2] buffer (rows it did not touch itself are zeros), ONE ring all-reduce sums the
// buffers, and the sums are written back.  The host learns U once (the buffer's size); nothing else crosses to the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asurf {

constexpr int kRowsPerBlock = 4096;   // rows a block of the compaction counts (256 threads x 16)

__global__ void __launch_bounds__(256)
k_rows_count(const uint8_t* __restrict__ mask, long long n_rows, int* __restrict__ block_counts) {
    __shared__ int s_wave[4];
    const int t = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * kRowsPerBlock;
    int c = 0;
#pragma unroll
    for (int j = 0; j < kRowsPerBlock / 256; ++j) {
        const long long row = row0 + (long long)t * (kRowsPerBlock / 256) + j;
        c += (row < n_rows && mask[row]) ? 1 : 0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((t & 63) == 0) s_wave[t >> 6] = c;
    __syncthreads();
    if (t == 0) block_counts[blockIdx.x] = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
}

// exclusive scan of the block counts in place (one block), total -> *count
__global__ void __launch_bounds__(1024)
k_rows_scan(int* __restrict__ block_counts, int n_blocks, int* __restrict__ count) {
    __shared__ int s_part[1024];
    const int t = threadIdx.x;
    const int per = (n_blocks + 1023) / 1024;
    const int lo = min(t * per, n_blocks), hi = min(lo + per, n_blocks);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += block_counts[i];
    s_part[t] = sum;
    __syncthreads();
    // Hillis-Steele over the 1024 partial sums
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = t >= d ? s_part[t - d] : 0;
        __syncthreads();
        s_part[t] += v;
        __syncthreads();
    }
    int run = s_part[t] - sum;
    for (int i = lo; i < hi; ++i) {
        const int c = block_counts[i];
        block_counts[i] = run;
        run += c;
    }
    if (t == 1023) *count = s_part[1023];
}

__global__ void __launch_bounds__(256)
k_rows_fill(const uint8_t* __restrict__ mask, long long n_rows, const int* __restrict__ block_offsets,
            int* __restrict__ rows_out) {
    __shared__ int s_wave[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const long long row0 = (long long)blockIdx.x * kRowsPerBlock + (long long)t * (kRowsPerBlock / 256);
    unsigned flags = 0;
#pragma unroll
    for (int j = 0; j < kRowsPerBlock / 256; ++j) {
        const long long row = row0 + j;
        if (row < n_rows && mask[row]) flags |= 1u << j;
    }
    const int c = __popc(flags);
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    int at = block_offsets[blockIdx.x] + incl - c;
    for (int k = 0; k < w; ++k) at += s_wave[k];
#pragma unroll
    for (int j = 0; j < kRowsPerBlock / 256; ++j)
        if (flags & (1u << j)) rows_out[at++] = (int)(row0 + j);
}

// packed[u][0 .. D) = grad_sh[row], packed[u][D] = grad_density[row], packed[u][D + 1] = grad_surface[row]
__global__ void __launch_bounds__(256)
k_rows_pack(const int* __restrict__ rows, long long n_list, int D, const float* __restrict__ grad_sh,
            const float* __restrict__ grad_density, const float* __restrict__ grad_surface, float* __restrict__ packed) {
    const int W = D + 2;
    const long long total = n_list * W;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long u = t / W;
        const int c = (int)(t - u * W);
        const long long row = rows[u];
        packed[t] = c < D ? grad_sh[row * D + c] : (c == D ? grad_density[row] : (grad_surface ? grad_surface[row] : 0.f));
    }
}

// ADD: the sums are added to what the buffers hold (a second partial batch of the same step, dist.PipelinedReduce)
template <bool ADD>
__global__ void __launch_bounds__(256)
k_rows_unpack(const int* __restrict__ rows, long long n_list, int D, const float* __restrict__ packed,
              float* __restrict__ grad_sh, float* __restrict__ grad_density, float* __restrict__ grad_surface,
              uint8_t* __restrict__ mask) {
    const int W = D + 2;
    const long long total = n_list * W;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long u = t / W;
        const int c = (int)(t - u * W);
        const long long row = rows[u];
        const float v = packed[t];
        if (c < D) grad_sh[row * D + c] = ADD ? grad_sh[row * D + c] + v : v;
        else if (c == D) {
            grad_density[row] = ADD ? grad_density[row] + v : v;
            if (mask) mask[row] = 1;
        } else if (grad_surface) grad_surface[row] = ADD ? grad_surface[row] + v : v;
    }
}

}  // namespace asurf
