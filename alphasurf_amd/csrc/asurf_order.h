// asurf_order.h -- ray ordering inside the fused call (asurf_order.hip; rocPRIM stays out of asurf_api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace asurf_order {
size_t tmp_bytes32(long long n);
// perm[i] = the caller's index of the ray that comes i-th (rays of one origin together, along a Morton curve of the
// direction); tmp: tmp_bytes32(n) bytes of device scratch
// key_bits: 30 (origin hash + direction) or 24 (direction only: every ray of the batch has the same origin)
int order32(const float* origins, const float* dirs, long long n, unsigned* perm, void* tmp, size_t tmp_bytes,
            hipStream_t s, int key_bits);
// The same ordering as a sort of its own (count per strip, scan, scatter, strip-local bitonic sort that also gathers):
// perm AND the rays / targets in the new order; tmp: tmp_bytes_strips(n) bytes
size_t tmp_bytes_strips(long long n);
int order_strips(const float* origins, const float* dirs, const float* gt, long long n, unsigned* perm, float* o_s,
                 float* d_s, float* gt_s, void* tmp, size_t tmp_bytes, hipStream_t s);
void gather(const float* origins, const float* dirs, const float* gt, const unsigned* perm, long long n, float* o_s,
            float* d_s, float* gt_s, hipStream_t s);
void scatter_rgb(const float* rgb_s, const unsigned* perm, long long n, float* rgb, hipStream_t s);
// the rays of a row-major image of `width` pixels a row (n a multiple of it) in the order of their 8 x 8 pixel tiles:
// perm and the rays / targets in that order
void tile_order(const float* origins, const float* dirs, const float* gt, long long n, int width, unsigned* perm, float* o_s,
                float* d_s, float* gt_s, hipStream_t s);
// out[3 b + 0 / 1 / 2], b < kProbeBlocks: coherent pairs, pairs looked at, pairs with different origins among block b's
// share of a sample of 8192 consecutive-ray pairs (the caller adds the blocks up)
constexpr int kProbeBlocks = 32;
void probe(const float* origins, const float* dirs, long long n, int* out, hipStream_t s);
}  // namespace asurf_order
