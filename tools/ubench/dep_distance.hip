// Micro-benchmark, part 3: does a dependent v_and -> v_bcnt pair cost more than the two instructions
// issued apart?  Streams of {and, bcnt} with the consumer 1, 2, 4 or 7 instructions behind its
// producer, 8 waves per SIMD.  (The --LD kernel's counting block was and,bcnt,and,bcnt,... on one
// temporary register.)
//   hipcc --offload-arch=gfx950 -O2 -o dep_distance tools/ubench/dep_distance.hip && ./dep_distance
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 2000
#define REP 8

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    uint32_t x = threadIdx.x * 2654435761u + seed, y = x * 7 + 1;
    uint32_t m0 = seed | 0x55aa55aa, m1 = seed * 3 | 0x0f0f, m2 = seed * 5 | 0x3333, m3 = seed * 7 | 0xff00ff;
    uint32_t t0, t1, t2, t3, t4, t5, t6;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
            if (KIND == 0)          // distance 1, one temporary (8 instructions)
                asm volatile("v_and_b32 %4, %5, %7\n v_bcnt_u32_b32 %0, %4, %0\n v_and_b32 %4, %6, %7\n v_bcnt_u32_b32 %1, %4, %1\n"
                             "v_and_b32 %4, %5, %8\n v_bcnt_u32_b32 %2, %4, %2\n v_and_b32 %4, %6, %8\n v_bcnt_u32_b32 %3, %4, %3"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0) : "v"(x), "v"(y), "v"(m0), "v"(m1));
            else if (KIND == 1)     // distance 2, two temporaries
                asm volatile("v_and_b32 %4, %6, %8\n v_and_b32 %5, %7, %8\n v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n"
                             "v_and_b32 %4, %6, %9\n v_and_b32 %5, %7, %9\n v_bcnt_u32_b32 %2, %4, %2\n v_bcnt_u32_b32 %3, %5, %3"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1) : "v"(x), "v"(y), "v"(m0), "v"(m1));
            else if (KIND == 2)     // distance 4, four temporaries
                asm volatile("v_and_b32 %4, %8, %10\n v_and_b32 %5, %9, %10\n v_and_b32 %6, %8, %11\n v_and_b32 %7, %9, %11\n"
                             "v_bcnt_u32_b32 %0, %4, %0\n v_bcnt_u32_b32 %1, %5, %1\n v_bcnt_u32_b32 %2, %6, %2\n v_bcnt_u32_b32 %3, %7, %3"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                             : "v"(x), "v"(y), "v"(m0), "v"(m1));
            else if (KIND == 3)     // the plane block, producers first: 7 and/bitop3, then 7 bcnt (14 instructions)
                asm volatile("v_and_b32 %7, %14, %16\n v_and_b32 %8, %15, %16\n v_bitop3_b32 %9, %14, %15, %16 bitop3:0x80\n"
                             "v_bitop3_b32 %10, %14, %16, %17 bitop3:0x80\n v_bitop3_b32 %11, %15, %16, %17 bitop3:0x80\n"
                             "v_bitop3_b32 %12, %14, %16, %18 bitop3:0x80\n v_bitop3_b32 %13, %15, %16, %18 bitop3:0x80\n"
                             "v_bcnt_u32_b32 %0, %7, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %9, %2\n v_bcnt_u32_b32 %3, %10, %3\n"
                             "v_bcnt_u32_b32 %4, %11, %4\n v_bcnt_u32_b32 %5, %12, %5\n v_bcnt_u32_b32 %6, %13, %6"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "=&v"(t0), "=&v"(t1), "=&v"(t2),
                               "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6)
                             : "v"(x), "v"(y), "v"(m0), "v"(m2), "v"(m3));
            else if (KIND == 4)     // the plane block as the compiler wrote it: producer, consumer, ... on one temporary
                asm volatile("v_and_b32 %7, %8, %10\n v_bcnt_u32_b32 %0, %7, %0\n v_and_b32 %7, %9, %10\n v_bcnt_u32_b32 %1, %7, %1\n"
                             "v_bitop3_b32 %7, %8, %9, %10 bitop3:0x80\n v_bcnt_u32_b32 %2, %7, %2\n"
                             "v_bitop3_b32 %7, %8, %10, %11 bitop3:0x80\n v_bcnt_u32_b32 %3, %7, %3\n"
                             "v_bitop3_b32 %7, %9, %10, %11 bitop3:0x80\n v_bcnt_u32_b32 %4, %7, %4\n"
                             "v_bitop3_b32 %7, %8, %10, %12 bitop3:0x80\n v_bcnt_u32_b32 %5, %7, %5\n"
                             "v_bitop3_b32 %7, %9, %10, %12 bitop3:0x80\n v_bcnt_u32_b32 %6, %7, %6"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "=&v"(t0)
                             : "v"(x), "v"(y), "v"(m0), "v"(m2), "v"(m3));
            else if (KIND == 5)     // bcnt only, 8 independent
                asm volatile("v_bcnt_u32_b32 %0, %8, %0\n v_bcnt_u32_b32 %1, %8, %1\n v_bcnt_u32_b32 %2, %8, %2\n v_bcnt_u32_b32 %3, %8, %3\n"
                             "v_bcnt_u32_b32 %4, %8, %4\n v_bcnt_u32_b32 %5, %8, %5\n v_bcnt_u32_b32 %6, %8, %6\n v_bcnt_u32_b32 %7, %8, %7"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(x));
            else if (KIND == 6)     // fast/slow alternating, all independent: and (unused result), bcnt
                asm volatile("v_and_b32 %4, %8, %10\n v_bcnt_u32_b32 %0, %8, %0\n v_and_b32 %5, %9, %10\n v_bcnt_u32_b32 %1, %9, %1\n"
                             "v_and_b32 %6, %8, %11\n v_bcnt_u32_b32 %2, %8, %2\n v_and_b32 %7, %9, %11\n v_bcnt_u32_b32 %3, %9, %3"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
                             : "v"(x), "v"(y), "v"(m0), "v"(m1));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}

template <int KIND>
double run(uint32_t *out, int per_asm)
{
    const int w = 8, blocks = 256 * w;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 2u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3 / ((double)w * ITER * REP * per_asm);
}

int main()
{
    uint32_t *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    const char *names[] = {"and->bcnt distance 1 (one temp)", "and->bcnt distance 2", "and->bcnt distance 4", "plane block, producers first",
                           "plane block, pairs on one temp", "bcnt only", "and + bcnt independent"};
    const double t[] = {run<0>(out, 8), run<1>(out, 8), run<2>(out, 8), run<3>(out, 14), run<4>(out, 14), run<5>(out, 8), run<6>(out, 8)};
    for (int i = 0; i < 7; ++i)
        printf("%-34s %.2f ns per wave-instruction per SIMD  (%.2f cycles @2.4GHz)\n", names[i], t[i] * 1e9, t[i] * 2.4e9);
    return 0;
}
