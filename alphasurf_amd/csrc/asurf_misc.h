// Grid maintenance kernels: what prune_grid / accelerate call into.
//   k_dilate            misc_kernel.cu:24-58    3x3x3 binary dilation, edges clamped
//   k_accel_dist_set    misc_kernel.cu:113-144  marks, level by level, the 2^k-cells that hold a vertex
//   k_accel_dist_prop   misc_kernel.cu:147-180  empty links become -(1 + number of empty pyramid levels)
// One voxel per thread, streaming.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asurf {

__global__ void __launch_bounds__(256)
k_dilate(const uint8_t* __restrict__ grid, uint8_t* __restrict__ out, int sx, int sy, int sz) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)sx * sy * sz) return;
    const int z = (int)(tid % sz);
    const int xy = (int)(tid / sz);
    const int y = xy % sy, x = xy / sy;
    const int xs[3] = {max(x - 1, 0), x, min(x + 1, sx - 1)};
    const int ys[3] = {max(y - 1, 0), y, min(y + 1, sy - 1)};
    const int zs[3] = {max(z - 1, 0), z, min(z + 1, sz - 1)};
    uint8_t any = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) any |= grid[((size_t)xs[i] * sy + ys[j]) * sz + zs[k]];
    out[tid] = any ? 1 : 0;
}

__host__ __device__ inline int div2_ceil(int x) { return ((x - 1) >> 1) + 1; }  // cuda_util.cuh int_div2_ceil

template <bool SET>
__global__ void __launch_bounds__(256)
k_accel_dist(int32_t* __restrict__ links, uint8_t* __restrict__ tmp, int sx0, int sy0, int sz0) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)sx0 * sy0 * sz0) return;
    int sx = sx0, sy = sy0, sz = sz0;
    int z = (int)(tid % sz);
    const int xy = (int)(tid / sz);
    int y = xy % sy, x = xy / sy;
    uint8_t* base = tmp;
    const int val = links[tid];
    if (SET ? (val >= 0) : (val < 0)) {
        int result = -1;
        while (sx > 1 && sy > 1 && sz > 1) {
            x >>= 1;
            y >>= 1;
            z >>= 1;
            sx = div2_ceil(sx);
            sy = div2_ceil(sy);
            sz = div2_ceil(sz);
            const size_t idx = ((size_t)x * sy + y) * sz + z;
            if (SET) {
                base[idx] = 1;
            } else {
                if (base[idx]) break;
                result -= 1;
            }
            base += (size_t)sx * sy * sz;
        }
        if (!SET) links[tid] = result;
    }
}

}  // namespace asurf
