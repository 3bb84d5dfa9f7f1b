// Micro-benchmark, part 6: the 1:1 mix of v_and_b32 (2.3 cycles alone) and v_bcnt_u32_b32 (4.2 alone) costs
// 3.8-4.0 cycles per instruction back to back (mix_order.hip), not the 3.25 average of its parts.  Here: the
// same dependent pairs with something scalar between the two instructions of a pair or between pairs
// (s_nop 0, s_nop 1, a SALU instruction), 4 and 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o nop_mix tools/ubench/nop_mix.hip && ./nop_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 2000
#define REP 4

// one dependent pair through temporary t (operand 8..15), counter c (0..7); x = %17, m = %16
#define P_PLAIN(c, t) "v_and_b32 %" #t ", %17, %16\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n"
#define P_NOP_IN(c, t) "v_and_b32 %" #t ", %17, %16\n s_nop 0\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n"
#define P_NOP_AFTER(c, t) "v_and_b32 %" #t ", %17, %16\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n s_nop 0\n"
#define P_NOP_BOTH(c, t) "v_and_b32 %" #t ", %17, %16\n s_nop 0\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n s_nop 0\n"
#define P_NOP1_IN(c, t) "v_and_b32 %" #t ", %17, %16\n s_nop 1\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n"
#define P_SALU_IN(c, t) "v_and_b32 %" #t ", %17, %16\n s_add_u32 %18, %18, 1\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n"
#define P_SALU_AFTER(c, t) "v_and_b32 %" #t ", %17, %16\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n s_add_u32 %18, %18, 1\n"
// the bcnt first, then the and of the NEXT pair (distance 2 through two temporaries), nop after the and
#define P_BCNT_ONLY(c, t) "v_bcnt_u32_b32 %" #c ", %17, %" #c "\n"
#define P_BCNT_NOP(c, t) "v_bcnt_u32_b32 %" #c ", %17, %" #c "\n s_nop 0\n"
#define P_AND_ONLY(c, t) "v_and_b32 %" #t ", %17, %16\n"
#define P_AND_NOP(c, t) "v_and_b32 %" #t ", %17, %16\n s_nop 0\n"


// groups of four vector instructions between scalar ones, and other orders
#define Q_FSFS_N(c, d, t, u) "v_and_b32 %" #t ", %17, %16\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n v_and_b32 %" #u ", %17, %16\n v_bcnt_u32_b32 %" #d ", %" #u ", %" #d "\n s_nop 0\n"
#define Q_FFSS_N(c, d, t, u) "v_and_b32 %" #t ", %17, %16\n v_and_b32 %" #u ", %17, %16\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n v_bcnt_u32_b32 %" #d ", %" #u ", %" #d "\n s_nop 0\n"
#define Q_FF_N_SS_N(c, d, t, u) "v_and_b32 %" #t ", %17, %16\n v_and_b32 %" #u ", %17, %16\n s_nop 0\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n v_bcnt_u32_b32 %" #d ", %" #u ", %" #d "\n s_nop 0\n"
#define Q_SF_N(c, d, t, u) "v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n v_and_b32 %" #t ", %17, %16\n s_nop 0\n v_bcnt_u32_b32 %" #d ", %" #u ", %" #d "\n v_and_b32 %" #u ", %17, %16\n s_nop 0\n"
#define Q_F_N_SF_N_S(c, d, t, u) "v_and_b32 %" #t ", %17, %16\n s_nop 0\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n v_and_b32 %" #u ", %17, %16\n s_nop 0\n v_bcnt_u32_b32 %" #d ", %" #u ", %" #d "\n"
#define Q_BITOP(c, d, t, u) "v_bitop3_b32 %" #t ", %17, %16, %" #u " bitop3:0x80\n s_nop 0\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n v_bitop3_b32 %" #t ", %17, %16, %" #u " bitop3:0x80\n s_nop 0\n v_bcnt_u32_b32 %" #d ", %" #t ", %" #d "\n"
#define Q_BITOP_PLAIN(c, d, t, u) "v_bitop3_b32 %" #t ", %17, %16, %" #u " bitop3:0x80\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n v_bitop3_b32 %" #t ", %17, %16, %" #u " bitop3:0x80\n v_bcnt_u32_b32 %" #d ", %" #t ", %" #d "\n"
#define Q_SETPRIO(c, d, t, u) "v_and_b32 %" #t ", %17, %16\n s_setprio 0\n v_bcnt_u32_b32 %" #c ", %" #t ", %" #c "\n v_and_b32 %" #u ", %17, %16\n s_setprio 0\n v_bcnt_u32_b32 %" #d ", %" #u ", %" #d "\n"
#define FOURQ(Q) Q(0, 1, 8, 9) Q(2, 3, 10, 11) Q(4, 5, 12, 13) Q(6, 7, 14, 15)
#define EIGHT(P) P(0, 8) P(1, 9) P(2, 10) P(3, 11) P(4, 12) P(5, 13) P(6, 14) P(7, 15)
// every pair through ONE temporary (like the compiler's code in the --LD kernel)
#define EIGHT1(P) P(0, 8) P(1, 8) P(2, 8) P(3, 8) P(4, 8) P(5, 8) P(6, 8) P(7, 8)

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    uint32_t t0 = 1, t1 = 2, t2 = 3, t3 = 4, t4 = 5, t5 = 6, t6 = 7, t7 = 8;
    uint32_t x = threadIdx.x * 2654435761u + seed, m = seed | 0x55aa55aa, sc = seed;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#define ARGS : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), \
               "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7) : "v"(m), "v"(x), "s"(sc)
#define ARGS_S : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), \
                 "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7) : "v"(m), "v"(x), "s"(sc) : "scc"
            if (KIND == 0) asm volatile(EIGHT(P_PLAIN) EIGHT(P_PLAIN) ARGS);
            else if (KIND == 1) asm volatile(EIGHT(P_NOP_IN) EIGHT(P_NOP_IN) ARGS);
            else if (KIND == 2) asm volatile(EIGHT(P_NOP_AFTER) EIGHT(P_NOP_AFTER) ARGS);
            else if (KIND == 3) asm volatile(EIGHT(P_NOP_BOTH) EIGHT(P_NOP_BOTH) ARGS);
            else if (KIND == 4) asm volatile(EIGHT(P_NOP1_IN) EIGHT(P_NOP1_IN) ARGS);
            else if (KIND == 5) asm volatile(EIGHT(P_SALU_IN) EIGHT(P_SALU_IN) ARGS_S);
            else if (KIND == 6) asm volatile(EIGHT(P_SALU_AFTER) EIGHT(P_SALU_AFTER) ARGS_S);
            else if (KIND == 7) asm volatile(EIGHT1(P_PLAIN) EIGHT1(P_PLAIN) ARGS);
            else if (KIND == 8) asm volatile(EIGHT1(P_NOP_IN) EIGHT1(P_NOP_IN) ARGS);
            else if (KIND == 9) asm volatile(EIGHT(P_BCNT_ONLY) EIGHT(P_BCNT_ONLY) EIGHT(P_BCNT_ONLY) EIGHT(P_BCNT_ONLY) ARGS);
            else if (KIND == 10) asm volatile(EIGHT(P_BCNT_NOP) EIGHT(P_BCNT_NOP) EIGHT(P_BCNT_NOP) EIGHT(P_BCNT_NOP) ARGS);
            else if (KIND == 11) asm volatile(EIGHT(P_AND_ONLY) EIGHT(P_AND_ONLY) EIGHT(P_AND_ONLY) EIGHT(P_AND_ONLY) ARGS);
            else if (KIND == 12) asm volatile(EIGHT(P_AND_NOP) EIGHT(P_AND_NOP) EIGHT(P_AND_NOP) EIGHT(P_AND_NOP) ARGS);
            else if (KIND == 13) asm volatile(FOURQ(Q_FSFS_N) FOURQ(Q_FSFS_N) ARGS);
            else if (KIND == 14) asm volatile(FOURQ(Q_FFSS_N) FOURQ(Q_FFSS_N) ARGS);
            else if (KIND == 15) asm volatile(FOURQ(Q_FF_N_SS_N) FOURQ(Q_FF_N_SS_N) ARGS);
            else if (KIND == 16) asm volatile(FOURQ(Q_SF_N) FOURQ(Q_SF_N) ARGS);
            else if (KIND == 17) asm volatile(FOURQ(Q_F_N_SF_N_S) FOURQ(Q_F_N_SF_N_S) ARGS);
            else if (KIND == 18) asm volatile(FOURQ(Q_BITOP) FOURQ(Q_BITOP) ARGS);
            else if (KIND == 19) asm volatile(FOURQ(Q_BITOP_PLAIN) FOURQ(Q_BITOP_PLAIN) ARGS);
            else if (KIND == 20) asm volatile(FOURQ(Q_SETPRIO) FOURQ(Q_SETPRIO) ARGS);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + t0 + t1 + t2 + t3 + t4 + t5 + t6 + t7;
}

template <int KIND>
double run(uint32_t *out, int w)
{
    const int blocks = 256 * w;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 2u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / ((double)w * ITER * REP * 32);       // 32 vector instructions per block
}

int main()
{
    uint32_t *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    const char *names[] = {"and bcnt (pairs, 8 temps)", "and NOP bcnt", "and bcnt NOP", "and NOP bcnt NOP", "and NOP1 bcnt",
                           "and SALU bcnt", "and bcnt SALU", "and bcnt (one temp)", "and NOP bcnt (one temp)",
                           "bcnt only", "bcnt NOP", "and only", "and NOP", "[and bcnt and bcnt] NOP", "[and and bcnt bcnt] NOP", "[and and] NOP [bcnt bcnt] NOP", "[bcnt and] NOP", "and NOP [bcnt and] NOP bcnt", "bitop3 NOP bcnt", "bitop3 bcnt", "and SETPRIO bcnt"};
    for (int w : {4, 8}) {
        const double t[] = {run<0>(out, w), run<1>(out, w), run<2>(out, w), run<3>(out, w), run<4>(out, w), run<5>(out, w), run<6>(out, w),
                            run<7>(out, w), run<8>(out, w), run<9>(out, w), run<10>(out, w), run<11>(out, w), run<12>(out, w), run<13>(out, w), run<14>(out, w), run<15>(out, w), run<16>(out, w), run<17>(out, w), run<18>(out, w), run<19>(out, w), run<20>(out, w)};
        for (int i = 0; i < 21; ++i)
            printf("waves/SIMD=%d  %-34s %.2f ns per vector instruction per SIMD  (%.2f cycles @2.4GHz)\n", w, names[i], t[i] * 1e9, t[i] * 2.4e9);
    }
    return 0;
}
