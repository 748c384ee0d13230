// VALU issue cost by instruction class (MI355X): cycles a wave64 vector instruction of each class occupies its SIMD.
// issue.hip prices v_fma_f32 alone; the kernels of this repository that are "VALU-issue bound" (k_sweep: integer and
// compare work; the regularisers: IEEE divisions) run other classes.
//   hipcc --offload-arch=gfx950 -O3 -o issue_mix issue_mix.hip && ./issue_mix
// Each wave runs a loop of 64 independent instructions of ONE class (8 accumulators x 8), no memory, four waves a SIMD.
// Findings (profiles/r4_micro_issue_mix.txt):
//   * f32 fma / add / mul, integer add / sub, two-operand bit-wise logic, v_mov: 2.5 - 3.1 cycles; shifts, bit fields,
//     compares, selects, v_min / v_max (f32, i32), three-operand integer forms (v_lshl_add_u32, v_or3_b32,
//     v_mad_u32_u24), v_bcnt, v_readlane, integer multiplies, conversions, v_floor, 64-bit shifts, packed fma:
//     4.2 - 4.7; f64: 4.8 - 5.2; rcp / sqrt: 8.3 - 8.6.
//     variants/isa_mix.py prices a kernel's static mix with these (bench.py's valu_mix_frac).
//   * v_cndmask_b32 in its SHORT encoding (VOP2, `_e32`: the condition is the implicit VCC) costs 4.4 cycles the first
//     time it reads a freshly written VCC and ~18 every further time: a compare followed by eight selects on the same
//     condition takes 136 cycles a group, the same eight selects in the long encoding (VOP3, `_e64`, VCC or any SGPR pair
//     named as an operand) 38.  (The lines "v_cndmask_b32 ..." without a compare in front -- VCC written once before the
//     loop -- show the slow path alone: 23 cycles an instruction.)  The compiler picks the short encoding whenever the
//     condition sits in VCC; struct selects (`c ? a : b` over several fields) are runs of exactly that.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define S_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define S_ADD32(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define S_MUL32(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define S_ADDU(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define S_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define S_LSHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define S_BFE(i) "v_bfe_u32 %" #i ", %" #i ", 3, 9\n"
#define S_CNDM(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define S_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define S_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define S_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define S_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define S_CVT(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define S_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define S_ADD64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define S_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define S_LSHL64(i) "v_lshlrev_b64 %" #i ", 1, %" #i "\n"
#define S_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define S_PAIRV(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define S_PAIRS(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %8\n v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[20:21]\n"
#define S_PAIRS2(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %8\n s_nop 1\n v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[20:21]\n"
#define S_TRIV(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %8\n s_xor_b64 vcc, s[20:21], s[22:23]\n v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define S_TRIS(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %8\n s_xor_b64 s[24:25], s[20:21], s[22:23]\n v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[24:25]\n"
#define S_CND1(i) "v_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define S_CND1S(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[20:21]\n"
#define S_CND1V64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %9, vcc\n"
#define S_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define S_MAXF(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define S_MAXI(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define S_FLOOR(i) "v_floor_f32 %" #i ", %" #i "\n"
#define S_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define S_OR3(i) "v_or3_b32 %" #i ", %" #i ", %8, %9\n"
#define S_BCNT(i) "v_bcnt_u32_b32 %" #i ", %" #i ", %8\n"
#define S_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define S_RDLANE(i) "v_readlane_b32 s20, %" #i ", 3\n"
#define S_SUBU(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define S_CNDS(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n"
#define S_CNDK(i) "v_cndmask_b32 %" #i ", 0, %" #i ", vcc\n"
#define S_CNDD(i) "v_cndmask_b32 %" #i ", %9, %8, vcc\n"

#define RUNGX(PRE, S) asm volatile(PRE REP8(S) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc", "scc", "s20", "s21", "s24", "s25")
#define RUNG(PRE) asm volatile(PRE REP8(S_CND1) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c) : "vcc", "scc", "s20", "s21", "s24", "s25")

template <int KIND>
__global__ void __launch_bounds__(256) k_mix(float* out, int iters) {
    float m = 1.0000001f, c = 1e-9f;
    unsigned um = 0x7fffffffu;
    double dm = 1.0000001, dc = 1e-9;
    if constexpr (KIND < 13 || KIND >= 18) {
        asm volatile("s_mov_b64 s[20:21], 0x5555\n s_mov_b64 s[22:23], 0x0f0f\n s_mov_b64 vcc, 0x3333" ::: "s20", "s21", "s22", "s23", "vcc");
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#define RUN32(S, X, Y) asm volatile(REP8(S) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(X), "v"(Y) : "vcc", "scc", "s20", "s21", "s24", "s25")
                if constexpr (KIND == 0) RUN32(S_FMA32, m, c);
                if constexpr (KIND == 1) RUN32(S_ADD32, c, c);
                if constexpr (KIND == 2) RUN32(S_MUL32, m, c);
                if constexpr (KIND == 3) RUN32(S_ADDU, um, c);
                if constexpr (KIND == 4) RUN32(S_AND, um, c);
                if constexpr (KIND == 5) RUN32(S_LSHL, um, c);
                if constexpr (KIND == 6) RUN32(S_BFE, um, c);
                if constexpr (KIND == 7) RUN32(S_CNDM, m, c);
                if constexpr (KIND == 8) RUN32(S_CMP, m, c);
                if constexpr (KIND == 9) RUN32(S_MULLO, um, c);
                if constexpr (KIND == 10) RUN32(S_RCP, m, c);
                if constexpr (KIND == 11) RUN32(S_SQRT, m, c);
                if constexpr (KIND == 12) RUN32(S_CVT, m, c);
                if constexpr (KIND == 18) RUN32(S_CNDS, m, c);
                if constexpr (KIND == 40) RUN32(S_MOV, m, c);
                if constexpr (KIND == 41) RUN32(S_MAXF, m, c);
                if constexpr (KIND == 42) RUN32(S_MAXI, um, c);
                if constexpr (KIND == 43) RUN32(S_FLOOR, m, c);
                if constexpr (KIND == 44) RUN32(S_LSHLADD, um, c);
                if constexpr (KIND == 45) RUN32(S_OR3, um, c);
                if constexpr (KIND == 46) RUN32(S_BCNT, um, c);
                if constexpr (KIND == 47) RUN32(S_MAD24, um, c);
                if constexpr (KIND == 48) RUN32(S_RDLANE, um, c);
                if constexpr (KIND == 49) RUN32(S_SUBU, um, c);
                if constexpr (KIND == 24) RUN32(S_TRIV, m, c);
                if constexpr (KIND == 26) RUNG("v_cmp_lt_f32 vcc, %0, %8\n");
                if constexpr (KIND == 27) RUNG("v_cmp_lt_f32_e64 s[20:21], %0, %8\n s_xor_b64 vcc, s[20:21], s[22:23]\n");
                if constexpr (KIND == 29) RUNGX("v_cmp_lt_f32_e64 s[20:21], %0, %8\n", S_CND1S);
                if constexpr (KIND == 30) RUNGX("v_cmp_lt_f32 vcc, %0, %8\n", S_CND1V64);
                if constexpr (KIND == 28) RUNG("s_xor_b64 vcc, s[20:21], s[22:23]\n");
                if constexpr (KIND == 25) RUN32(S_TRIS, m, c);
                if constexpr (KIND == 21) RUN32(S_PAIRV, m, c);
                if constexpr (KIND == 22) RUN32(S_PAIRS, m, c);
                if constexpr (KIND == 23) RUN32(S_PAIRS2, m, c);
                if constexpr (KIND == 19) RUN32(S_CNDK, m, c);
                if constexpr (KIND == 20) RUN32(S_CNDD, m, c);
            }
        }
        const float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        if (s == 12345.678f) out[0] = s;
    } else {
        double a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#define RUN64(S) asm volatile(REP8(S) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(dm), "v"(dc))
                if constexpr (KIND == 13) RUN64(S_FMA64);
                if constexpr (KIND == 14) RUN64(S_ADD64);
                if constexpr (KIND == 15) RUN64(S_MUL64);
                if constexpr (KIND == 16) RUN64(S_LSHL64);
                if constexpr (KIND == 17) RUN64(S_PKFMA);
            }
        }
        const double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        if (s == 12345.678) out[0] = (float)s;
    }
}

template <int KIND>
static void run(const char* name, float* out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 4000, W = 4;
    const int blocks = 256 * W;   // blocks of 256 threads = one wave per SIMD of a CU; W blocks per CU
    k_mix<KIND><<<blocks, 256>>>(out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_mix<KIND><<<blocks, 256>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("  (%s: %s)\n", name, hipGetErrorString(err));
    const double wave_inst_per_simd = (double)iters * 64.0 * W;
    if (name[0] == 'G') printf("%-16s %.3f ms  %.1f cycles per GROUP per SIMD\n", name, ms, 8.0 * ms * 1e-3 * 2.4e9 / wave_inst_per_simd);
    else printf("%-16s %.3f ms  %.2f cycles per wave-instruction per SIMD (2.4 GHz nominal, %d waves a SIMD)\n", name, ms,
           ms * 1e-3 * 2.4e9 / wave_inst_per_simd, W);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 4);
    run<0>("v_fma_f32", out);
    run<1>("v_add_f32", out);
    run<2>("v_mul_f32", out);
    run<3>("v_add_u32", out);
    run<4>("v_and_b32", out);
    run<5>("v_lshlrev_b32", out);
    run<6>("v_bfe_u32", out);
    run<7>("v_cndmask_b32", out);
    run<8>("v_cmp_lt_f32", out);
    run<9>("v_mul_lo_u32", out);
    run<10>("v_rcp_f32", out);
    run<11>("v_sqrt_f32", out);
    run<12>("v_cvt_f32_i32", out);
    run<40>("v_mov_b32", out);
    run<41>("v_max_f32", out);
    run<42>("v_max_i32", out);
    run<43>("v_floor_f32", out);
    run<44>("v_lshl_add_u32", out);
    run<45>("v_or3_b32", out);
    run<46>("v_bcnt_u32_b32", out);
    run<47>("v_mad_u32_u24", out);
    run<48>("v_readlane_b32", out);
    run<49>("v_sub_u32", out);
    run<13>("v_fma_f64", out);
    run<14>("v_add_f64", out);
    run<15>("v_mul_f64", out);
    run<16>("v_lshlrev_b64", out);
    run<17>("v_pk_fma_f32", out);
    run<7>("v_cndmask_b32 (vcc set)", out);
    run<18>("v_cndmask_b32_e64 sgpr", out);
    run<19>("v_cndmask 0,v,vcc", out);
    run<20>("v_cndmask d,a,b (no dep)", out);
    run<21>("PAIR cmp->vcc + cndmask vcc (per pair)", out);
    run<22>("PAIR cmp->sgpr + cndmask sgpr (per pair)", out);
    run<23>("PAIR sgpr with s_nop 1 (per pair)", out);
    run<24>("TRIPLE cmp->sgpr, s_xor->vcc, cndmask vcc (per triple)", out);
    run<25>("TRIPLE cmp->sgpr, s_xor->sgpr, cndmask sgpr (per triple)", out);
    run<26>("GROUP v_cmp->vcc + 8 cndmask vcc (per group of 9)", out);
    run<27>("GROUP v_cmp->sgpr, s_xor->vcc + 8 cndmask vcc (per group of 10)", out);
    run<28>("GROUP s_xor->vcc + 8 cndmask vcc (per group of 9)", out);
    run<29>("GROUP v_cmp->sgpr + 8 cndmask_e64 sgpr (per group of 9)", out);
    run<30>("GROUP v_cmp->vcc + 8 cndmask_e64 vcc (per group of 9)", out);
    return 0;
}
