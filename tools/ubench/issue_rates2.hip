// Micro-benchmark, part 2: sustained issue cost (cycles per wave-instruction per SIMD at 8 waves per
// SIMD) of the candidate instructions for the window-end arithmetic of the --LD kernel (integer
// three-operand forms, fp64, DPP moves).  All operands are VGPRs (an SGPR operand halves the rate of
// the fast ops, see issue_rates.hip).  Four independent chains per kind.
//   hipcc --offload-arch=gfx950 -O2 -o issue_rates2 tools/ubench/issue_rates2.hip && ./issue_rates2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 2000
#define REP 8          // asm blocks per iteration, 8 instructions each

// 8 instructions on 4 chains: INSN(d) expands to the text of one instruction writing %d
#define BLOCK4(T0, T1, T2, T3) T0 "\n" T1 "\n" T2 "\n" T3 "\n" T0 "\n" T1 "\n" T2 "\n" T3

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7;
    uint32_t m = seed | 0x55aa55aa, n = seed * 9 + 1;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, dm = 1.0000001;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#define I3(op) asm volatile(BLOCK4(op " %0, %0, %4, %5", op " %1, %1, %4, %5", op " %2, %2, %4, %5", op " %3, %3, %4, %5") \
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(n))
#define I2(op) asm volatile(BLOCK4(op " %0, %4, %0", op " %1, %4, %1", op " %2, %4, %2", op " %3, %4, %3") \
                            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m))
#define D2(op) asm volatile(BLOCK4(op " %0, %0, %4", op " %1, %1, %4", op " %2, %2, %4", op " %3, %3, %4") \
                            : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm))
            if (KIND == 0) I3("v_lshl_add_u32");
            else if (KIND == 1) I3("v_add3_u32");
            else if (KIND == 2) I3("v_mad_u32_u24");
            else if (KIND == 3) I3("v_mad_i32_i24");
            else if (KIND == 4) I2("v_lshlrev_b32");
            else if (KIND == 5) I2("v_sub_u32");
            else if (KIND == 6) I3("v_add_lshl_u32");
            else if (KIND == 7) I3("v_lshl_or_b32");
            else if (KIND == 8) I3("v_and_or_b32");
            else if (KIND == 9) I3("v_or3_b32");
            else if (KIND == 10) I3("v_bfe_u32");
            else if (KIND == 11) I3("v_alignbit_b32");
            else if (KIND == 12) I3("v_perm_b32");
            else if (KIND == 13) I3("v_bfi_b32");
            else if (KIND == 14) I3("v_xad_u32");
            else if (KIND == 15) I2("v_mul_u32_u24");
            else if (KIND == 16) asm volatile(BLOCK4("v_mul_lo_u32 %0, %0, %4", "v_mul_lo_u32 %1, %1, %4", "v_mul_lo_u32 %2, %2, %4", "v_mul_lo_u32 %3, %3, %4")
                                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            else if (KIND == 17) I3("v_sad_u32");
            else if (KIND == 18) I2("v_xor_b32");
            else if (KIND == 19) asm volatile(BLOCK4("v_mov_b32 %0, %4", "v_mov_b32 %1, %4", "v_mov_b32 %2, %4", "v_mov_b32 %3, %4")
                                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            else if (KIND == 20) asm volatile(BLOCK4("v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
                                                     "v_mov_b32_dpp %2, %4 row_mirror row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %4 row_bcast:15 row_mask:0xa bank_mask:0xf")
                                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            else if (KIND == 21) asm volatile(BLOCK4("v_add_u32_dpp %0, %4, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %1, %4, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
                                                     "v_add_u32_dpp %2, %4, %2 row_mirror row_mask:0xf bank_mask:0xf", "v_add_u32_dpp %3, %4, %3 row_mirror row_mask:0xf bank_mask:0xf")
                                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            else if (KIND == 22) D2("v_add_f64");
            else if (KIND == 23) D2("v_mul_f64");
            else if (KIND == 24) asm volatile(BLOCK4("v_fma_f64 %0, %0, %4, %0", "v_fma_f64 %1, %1, %4, %1", "v_fma_f64 %2, %2, %4, %2", "v_fma_f64 %3, %3, %4, %3")
                                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm));
            else if (KIND == 25) asm volatile(BLOCK4("v_ldexp_f64 %0, %0, %4", "v_ldexp_f64 %1, %1, %4", "v_ldexp_f64 %2, %2, %4", "v_ldexp_f64 %3, %3, %4")
                                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(n & 1));
            else if (KIND == 26) asm volatile(BLOCK4("v_cndmask_b32 %0, %0, %4, vcc", "v_cndmask_b32 %1, %1, %4, vcc", "v_cndmask_b32 %2, %2, %4, vcc", "v_cndmask_b32 %3, %3, %4, vcc")
                                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m) : "vcc");
            else if (KIND == 27) asm volatile(BLOCK4("v_bcnt_u32_b32 %0, %4, 0", "v_bcnt_u32_b32 %1, %4, 0", "v_bcnt_u32_b32 %2, %4, 0", "v_bcnt_u32_b32 %3, %4, 0")
                                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            else if (KIND == 28) I3("v_dot4_u32_u8");
            else if (KIND == 29) I3("v_sad_u8");
            else if (KIND == 30) I3("v_mad_u32_u16");
            else if (KIND == 31) I2("v_min_u32");
            else if (KIND == 32) asm volatile(BLOCK4("v_pk_mov_b32 %0, %4, %4", "v_pk_mov_b32 %1, %4, %4", "v_pk_mov_b32 %2, %4, %4", "v_pk_mov_b32 %3, %4, %4")
                                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm));
            else if (KIND == 33) asm volatile(BLOCK4("v_lshlrev_b64 %0, 1, %0", "v_lshlrev_b64 %1, 1, %1", "v_lshlrev_b64 %2, 1, %2", "v_lshlrev_b64 %3, 1, %3")
                                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
            else if (KIND == 34) asm volatile(BLOCK4("v_mov_b64 %0, %4", "v_mov_b64 %1, %4", "v_mov_b64 %2, %4", "v_mov_b64 %3, %4")
                                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm));
            else if (KIND == 35) asm volatile(BLOCK4("v_mov_b64_dpp %0, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf", "v_mov_b64_dpp %1, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf",
                                                     "v_mov_b64_dpp %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf", "v_mov_b64_dpp %3, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf")
                                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm));
            else if (KIND == 36) asm volatile(BLOCK4("v_fmac_f64_dpp %0, %4, %0 row_newbcast:1 row_mask:0xf bank_mask:0xf", "v_fmac_f64_dpp %1, %4, %1 row_newbcast:1 row_mask:0xf bank_mask:0xf",
                                                     "v_fmac_f64_dpp %2, %4, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf", "v_fmac_f64_dpp %3, %4, %3 row_newbcast:1 row_mask:0xf bank_mask:0xf")
                                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(dm));
            else if (KIND == 37) asm volatile(BLOCK4("v_permlane32_swap_b32 %0, %1", "v_permlane32_swap_b32 %2, %3", "v_permlane32_swap_b32 %0, %1", "v_permlane32_swap_b32 %2, %3")
                                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            else if (KIND == 38) asm volatile(BLOCK4("v_permlane16_swap_b32 %0, %1", "v_permlane16_swap_b32 %2, %3", "v_permlane16_swap_b32 %0, %1", "v_permlane16_swap_b32 %2, %3")
                                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(d0 + d1 + d2 + d3);
}

template <int KIND>
double run(uint32_t *out)
{
    const int w = 8, blocks = 256 * w;       // 256 CUs x 8 blocks of 4 waves = 8 waves per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 2u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / ((double)w * ITER * REP * 8);
}

template <int K0, int K1>
void run_all(uint32_t *out, const char *const *names)
{
    if constexpr (K0 < K1) {
        const double t = run<K0>(out);
        printf("%-22s %.2f ns per wave-instruction per SIMD  (%.2f cycles @2.4GHz)\n", names[K0], t * 1e9, t * 2.4e9);
        fflush(stdout);
        run_all<K0 + 1, K1>(out, names);
    }
}

int main()
{
    uint32_t *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    static const char *names[] = {"v_lshl_add_u32", "v_add3_u32", "v_mad_u32_u24", "v_mad_i32_i24", "v_lshlrev_b32", "v_sub_u32", "v_add_lshl_u32",
                                  "v_lshl_or_b32", "v_and_or_b32", "v_or3_b32", "v_bfe_u32", "v_alignbit_b32", "v_perm_b32", "v_bfi_b32", "v_xad_u32",
                                  "v_mul_u32_u24", "v_mul_lo_u32", "v_sad_u32", "v_xor_b32", "v_mov_b32", "v_mov_b32 dpp", "v_add_u32 dpp", "v_add_f64",
                                  "v_mul_f64", "v_fma_f64", "v_ldexp_f64", "v_cndmask_b32", "v_bcnt_u32_b32 (+0)", "v_dot4_u32_u8", "v_sad_u8",
                                  "v_mad_u32_u16", "v_min_u32", "v_pk_mov_b32", "v_lshlrev_b64", "v_mov_b64", "v_mov_b64 dpp newbcast",
                                  "v_fmac_f64 dpp newbcast", "v_permlane32_swap", "v_permlane16_swap"};
    run_all<0, 39>(out, names);
    return 0;
}
