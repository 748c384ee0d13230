// Rate of float atomics by scope and pattern (MI355X): rows of 27 contiguous floats vs single scattered floats;
// agent scope (what atomicAdd() is) vs workgroup scope (performed in the XCD's own L2).
//   hipcc --offload-arch=gfx950 -O3 -o atomics atomics.hip && ./atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SCOPE, int WIDTH, bool OWN>
__global__ void k(float* a, unsigned n_rows, int iters, int stride_f) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);  // XCC_ID, 4 bits
    unsigned s = wave * 2654435761u + 12345u;
    for (int it = 0; it < iters; ++it) {
        // WIDTH lanes share a row; 64 / WIDTH rows per instruction (WIDTH = 1: every lane its own row)
        const int grp = lane / WIDTH, e = lane - grp * WIDTH;
        unsigned h = (s + grp * 40503u) * 2246822519u;
        h ^= h >> 15;
        unsigned row = h % n_rows;
        if (OWN) row = (row & ~7u) | xcc;  // rows this XCD owns
        if (row >= n_rows) row = xcc;
        if (grp * WIDTH + WIDTH <= 64)
            __hip_atomic_fetch_add(a + (size_t)row * stride_f + e, 1.f, __ATOMIC_RELAXED, SCOPE);
        s = s * 1664525u + 1013904223u;
    }
}

// MODE 0: scattered byte stores, 1: scattered float loads (summed), 2: scattered plain float load + store
template <int MODE>
__global__ void k2(float* a, unsigned n_rows, int iters, float* sink) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    unsigned s = wave * 2654435761u + 12345u;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        unsigned h = (s + lane * 40503u) * 2246822519u;
        h ^= h >> 15;
        const unsigned row = h % n_rows;
        if (MODE == 0) reinterpret_cast<unsigned char*>(a)[(size_t)row * 64] = 1;
        if (MODE == 1) acc += a[(size_t)row * 16];
        if (MODE == 2) a[(size_t)row * 16] += 1.f;
        s = s * 1664525u + 1013904223u;
    }
    if (acc == 12345.f) *sink = acc;
}
template <int MODE>
void run2(const char* name, float* a, unsigned n_rows) {
    const int blocks = 2048, iters = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k2<MODE><<<blocks, 256>>>(a, n_rows, 8, a);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k2<MODE><<<blocks, 256>>>(a, n_rows, iters, a);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 4 * iters * 64;
    printf("%-44s %8.3f ms  %8.1f G lane-ops/s\n", name, ms, ops / ms * 1e-6);
}

template <int SCOPE, int WIDTH, bool OWN>
void run(const char* name, float* a, unsigned n_rows, int stride_f) {
    const int blocks = 2048, iters = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<SCOPE, WIDTH, OWN><<<blocks, 256>>>(a, n_rows, 8, stride_f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<SCOPE, WIDTH, OWN><<<blocks, 256>>>(a, n_rows, iters, stride_f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double instr = (double)blocks * 4 * iters;
    const double lanes = instr * (64 / WIDTH) * WIDTH;
    const double rows = instr * (64 / WIDTH);
    printf("%-44s %8.3f ms  %8.1f G lane-atomics/s  %8.1f G row-transactions/s\n", name, ms, lanes / ms * 1e-6,
           rows / ms * 1e-6);
}

int main() {
    const unsigned n_rows = 1300000;
    float* a;
    (void)hipMalloc(&a, (size_t)n_rows * 27 * 4);
    (void)hipMemset(a, 0, (size_t)n_rows * 27 * 4);
    constexpr int AG = __HIP_MEMORY_SCOPE_AGENT, WG = __HIP_MEMORY_SCOPE_WORKGROUP;
    run<AG, 27, false>("agent, rows of 27 (2 per instr)", a, n_rows, 27);
    run<WG, 27, false>("workgroup, rows of 27", a, n_rows, 27);
    run<WG, 27, true>("workgroup, rows of 27, XCD-owned rows", a, n_rows, 27);
    run<AG, 27, true>("agent, rows of 27, XCD-owned rows", a, n_rows, 27);
    run<AG, 1, false>("agent, scattered floats (64 per instr)", a, n_rows, 1);
    run<WG, 1, false>("workgroup, scattered floats", a, n_rows, 1);
    run<WG, 1, true>("workgroup, scattered floats, XCD-owned", a, n_rows, 1);
    run<AG, 1, true>("agent, scattered floats, XCD-owned", a, n_rows, 1);
    run<AG, 8, false>("agent, rows of 8", a, n_rows, 27);
    run<WG, 8, true>("workgroup, rows of 8, XCD-owned", a, n_rows, 27);
    // small footprint (fits L2): 64k rows
    run<AG, 27, false>("agent, rows of 27, 64k rows", a, 65536, 27);
    run<WG, 27, true>("workgroup, rows of 27, 64k rows, owned", a, 65536, 27);
    run<AG, 1, false>("agent, scattered, 64k floats", a, 65536, 1);
    run<WG, 1, true>("workgroup, scattered, 64k floats, owned", a, 65536, 1);
    run<AG, 2, false>("agent, pairs of adjacent floats", a, n_rows, 2);
    run<AG, 4, false>("agent, quads of adjacent floats", a, n_rows, 4);
    run<AG, 16, false>("agent, 16 adjacent floats (64 B aligned)", a, n_rows, 16);
    run<AG, 32, false>("agent, 32 adjacent floats (128 B aligned)", a, n_rows / 2, 32);
    run<AG, 64, false>("agent, 64 adjacent floats (256 B aligned)", a, n_rows / 4, 64);
    run2<0>("scattered byte stores", a, 1300000 * 27 / 16);
    run2<1>("scattered float loads", a, 1300000 * 27 / 16);
    run2<2>("scattered plain load + store", a, 1300000 * 27 / 16);
    return 0;
}
