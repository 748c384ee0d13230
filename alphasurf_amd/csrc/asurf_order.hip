// Ray ordering for coherence (an extension of this build; the reference renders a batch in the order given).
// The walk and the gather kernels run one ray per lane: a wave is as slow as its slowest lane and its loads
// coalesce only when neighbouring lanes are neighbouring rays.  Training batches are shuffled.  One key kernel
// and one radix sort give a permutation that groups the rays of a camera (equal origins hash alike) and
// orders each group along a Morton curve of the unit direction; the caller gathers the rays with it, renders,
// and scatters the colours back.
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
