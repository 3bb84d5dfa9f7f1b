// Micro-benchmark: sustained issue cost (cycles per wave-instruction per SIMD) of the instruction
// kinds the LD kernel is made of, on gfx950.  Each kernel runs a long unrolled stream of one kind
// with 8 independent chains; launched with W waves per SIMD.  Build & run: see tools/ubench/run.sh
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define REP 64
#define ITER 2000

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t m = seed | 0x55aa55aa;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    uint32_t s0 = seed, s1 = seed * 3, s2 = seed * 5, s3 = seed * 7;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (KIND == 0) {        // v_and_b32
                asm volatile("v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n"
                             "v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m));
            } else if (KIND == 1) { // v_bcnt_u32_b32 accumulate
                asm volatile("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                             "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if (KIND == 2) { // v_bitop3_b32 (and3)
                asm volatile("v_bitop3_b32 %0, %8, %0, %9 bitop3:0x80\n v_bitop3_b32 %1, %8, %1, %9 bitop3:0x80\n v_bitop3_b32 %2, %8, %2, %9 bitop3:0x80\n v_bitop3_b32 %3, %8, %3, %9 bitop3:0x80\n"
                             "v_bitop3_b32 %4, %8, %4, %9 bitop3:0x80\n v_bitop3_b32 %5, %8, %5, %9 bitop3:0x80\n v_bitop3_b32 %6, %8, %6, %9 bitop3:0x80\n v_bitop3_b32 %7, %8, %7, %9 bitop3:0x80"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(m), "v"(a0));
            } else if (KIND == 3) { // v_readlane_b32
                asm volatile("v_readlane_b32 %0, %4, 1\n v_readlane_b32 %1, %4, 2\n v_readlane_b32 %2, %4, 3\n v_readlane_b32 %3, %4, 4\n"
                             "v_readlane_b32 %0, %4, 5\n v_readlane_b32 %1, %4, 6\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 8"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(a0));
            } else if (KIND == 4) { // v_mul_f64
                asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4\n"
                             "v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(1.0000001));
            } else if (KIND == 5) { // s_and_b32
                asm volatile("s_and_b32 %0, %0, %4\n s_and_b32 %1, %1, %4\n s_and_b32 %2, %2, %4\n s_and_b32 %3, %3, %4\n"
                             "s_and_b32 %0, %0, %4\n s_and_b32 %1, %1, %4\n s_and_b32 %2, %2, %4\n s_and_b32 %3, %3, %4"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(m) : "scc");
            } else if (KIND == 6) { // mixed: 4 valu + 4 salu interleaved
                asm volatile("v_and_b32 %0, %8, %0\n s_and_b32 %4, %4, %8\n v_and_b32 %1, %8, %1\n s_and_b32 %5, %5, %8\n"
                             "v_and_b32 %2, %8, %2\n s_and_b32 %6, %6, %8\n v_and_b32 %3, %8, %3\n s_and_b32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(m) : "scc");
            } else if (KIND == 7) { // dependent pair chain: and -> bcnt on one register
                asm volatile("v_and_b32 %1, %2, %0\n v_bcnt_u32_b32 %0, %1, %0\n v_and_b32 %1, %2, %0\n v_bcnt_u32_b32 %0, %1, %0\n"
                             "v_and_b32 %1, %2, %0\n v_bcnt_u32_b32 %0, %1, %0\n v_and_b32 %1, %2, %0\n v_bcnt_u32_b32 %0, %1, %0"
                             : "+v"(a0), "+v"(a1) : "s"(m));
            } else if (KIND == 9) { // v_and_b32, VGPR operands only
                asm volatile("v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n"
                             "v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
            } else if (KIND == 10) { // v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3\n"
                             "v_fma_f32 %0, %0, %4, %0\n v_fma_f32 %1, %1, %4, %1\n v_fma_f32 %2, %2, %4, %2\n v_fma_f32 %3, %3, %4, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            } else if (KIND == 11) { // v_pk_add_u16
                asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4\n"
                             "v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            } else if (KIND == 12) { // v_add_f32 (VOP2, VGPR only)
                asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n"
                             "v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            } else if (KIND == 13) { // v_pk_fma_f32 (64-bit operands)
                asm volatile("v_pk_fma_f32 %0, %0, %2, %0\n v_pk_fma_f32 %1, %1, %2, %1\n v_pk_fma_f32 %0, %0, %2, %0\n v_pk_fma_f32 %1, %1, %2, %1\n"
                             "v_pk_fma_f32 %0, %0, %2, %0\n v_pk_fma_f32 %1, %1, %2, %1\n v_pk_fma_f32 %0, %0, %2, %0\n v_pk_fma_f32 %1, %1, %2, %1"
                             : "+v"(d0), "+v"(d1) : "v"(d2));
            } else if (KIND == 14) { // v_add_u32 VGPR only
                asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n"
                             "v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            } else if (KIND == 15) { // v_bitop3_b32, VGPR operands only
                asm volatile("v_bitop3_b32 %0, %8, %0, %9 bitop3:0x80\n v_bitop3_b32 %1, %8, %1, %9 bitop3:0x80\n v_bitop3_b32 %2, %8, %2, %9 bitop3:0x80\n v_bitop3_b32 %3, %8, %3, %9 bitop3:0x80\n"
                             "v_bitop3_b32 %4, %8, %4, %9 bitop3:0x80\n v_bitop3_b32 %5, %8, %5, %9 bitop3:0x80\n v_bitop3_b32 %6, %8, %6, %9 bitop3:0x80\n v_bitop3_b32 %7, %8, %7, %9 bitop3:0x80"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(seed));
            } else if (KIND == 16) { // the LD kernel's plane block: 3 and + 4 bitop3 + 7 bcnt (14 instr), VGPR only; counted as 8 per asm -> scale 14/8
                uint32_t t0, t1, t2;
                asm volatile("v_and_b32 %8, %11, %12\n v_and_b32 %9, %11, %13\n v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %9, %1\n"
                             "v_bitop3_b32 %10, %11, %12, %13 bitop3:0x80\n v_bcnt_u32_b32 %2, %10, %2\n"
                             "v_bitop3_b32 %10, %8, %14, %14 bitop3:0x80\n v_bcnt_u32_b32 %3, %10, %3\n"
                             "v_bitop3_b32 %10, %9, %14, %14 bitop3:0x80\n v_bcnt_u32_b32 %4, %10, %4\n"
                             "v_bitop3_b32 %10, %8, %15, %15 bitop3:0x80\n v_bcnt_u32_b32 %5, %10, %5\n"
                             "v_and_b32 %10, %9, %15\n v_bcnt_u32_b32 %6, %10, %6"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(t0), "=&v"(t1), "=&v"(t2)
                             : "v"(m), "v"(seed), "v"(seed * 3), "v"(seed * 5), "v"(seed * 7));
            } else if (KIND == 17) { // v_lshl_add_u32 VGPR only
                asm volatile("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4\n"
                             "v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
            } else if (KIND == 8) { // s_nop 0
                asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0 + s1 + s2 + s3 + (uint32_t)(d0 + d1 + d2 + d3);
}

template <int KIND>
double run(int waves_per_simd, uint32_t *out)
{
    const int blocks = 256 * waves_per_simd;     // 256 CUs x (waves_per_simd blocks of 4 waves)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 2u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // instructions per SIMD = waves_per_simd * ITER * REP
    return ms * 1e-3 / ((double)waves_per_simd * ITER * REP);   // seconds per wave-instruction per SIMD
}

int main()
{
    uint32_t *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    const char *names[] = {"v_and_b32 (sgpr mask)", "v_bcnt_u32_b32", "v_bitop3_b32", "v_readlane_b32", "v_mul_f64", "s_and_b32", "v_and+s_and mixed", "and->bcnt dependent", "s_nop 0",
                           "v_and_b32 (vgpr only)", "v_fma_f32", "v_pk_add_u16", "v_add_f32", "v_pk_fma_f32", "v_add_u32 (vgpr only)",
                           "v_bitop3 (vgpr only)", "plane block x14/8", "v_lshl_add_u32 (vgpr)"};
    for (int w : {1, 2, 4, 8}) {
        double t[18] = {run<0>(w, out), run<1>(w, out), run<2>(w, out), run<3>(w, out), run<4>(w, out), run<5>(w, out), run<6>(w, out), run<7>(w, out), run<8>(w, out),
                        run<9>(w, out), run<10>(w, out), run<11>(w, out), run<12>(w, out), run<13>(w, out), run<14>(w, out),
                        run<15>(w, out), run<16>(w, out), run<17>(w, out)};
        for (int i = 0; i < 18; ++i)
            printf("waves/SIMD=%d  %-22s %.2f ns per wave-instruction per SIMD  (%.2f cycles @2.4GHz)\n", w, names[i], t[i] * 1e9, t[i] * 2.4e9);
        fflush(stdout);
    }
    return 0;
}
