// VALU issue rate of a SIMD by the number of waves that share it (MI355X): how many cycles a wave64 vector
// instruction occupies the SIMD -- the constant every "VALU-issue bound" estimate in DESIGN.md is priced with.
//   hipcc --offload-arch=gfx950 -O3 -o issue issue.hip && ./issue
// Each wave runs a loop of 64 independent v_fma_f32 (8 accumulators x 8) -- no memory, no LDS, no dependent chain
// shorter than 8 instructions.  Blocks of 64 * W threads, one block per CU-quarter... simply: grid = 256 CUs x 4 SIMDs
// worth of waves x W, so that every SIMD of the chip holds W waves (the dispatcher spreads equal blocks evenly).
// Output: for W = 1, 2, 4, 8 the cycles per wave-instruction per SIMD at the measured time and a 2.4 GHz clock, and
// the in-kernel clock (s_memtime / s_memrealtime) so that the figure can be corrected for the clock the chip held.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(512) k_issue(float* out, int iters, unsigned long long* clk) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 1.0000001f, c = 1e-9f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            // (inline asm: the compiler would pack pairs into v_pk_fma_f32, another instruction with another cost)
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n"
                         "v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n"
                         "v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                         : "v"(m), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678f) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = t1 - t0;
        clk[1] = r1 - r0;
    }
}

int main() {
    float* out;
    unsigned long long* clk;
    (void)hipMalloc(&out, 4);
    (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 20000;                       // 64 fma per iteration
    for (int W : {1, 2, 4, 8}) {
        // blocks of 256 threads = one wave per SIMD of a CU; W blocks per CU
        const int blocks = 256 * W;
        k_issue<<<blocks, 256>>>(out, 100, clk);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        k_issue<<<blocks, 256>>>(out, iters, clk);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double inst_per_wave = (double)iters * 64.0;
        const double wave_inst_per_simd = inst_per_wave * W;          // W waves on every SIMD
        const double ghz = (double)h[0] / ((double)h[1] * 10.0);      // s_memrealtime ticks at 100 MHz
        const double cyc_24 = ms * 1e-3 * 2.4e9 / wave_inst_per_simd;
        const double cyc_real = cyc_24 * ghz / 2.4;                   // ... at the clock the chip really held
        printf("waves/SIMD %d: %.3f ms, %.2f cycles per wave-instruction per SIMD at a nominal 2.4 GHz; in-kernel clock "
               "%.2f GHz -> %.2f cycles at that clock; one wave sees %.2f cycles between its own instructions\n",
               W, ms, cyc_24, ghz, cyc_real, (double)h[0] / inst_per_wave);
    }
    return 0;
}
