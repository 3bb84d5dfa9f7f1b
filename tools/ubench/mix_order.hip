// Micro-benchmark, part 4: a 1:1 mix of a fast (v_and_b32, 2.4 cycles alone) and a slow (v_bcnt_u32_b32,
// 4.2 alone) VALU instruction in different orders, all independent, 8 waves per SIMD: does any grouping
// reach the 3.3-cycle average of the two?
//   hipcc --offload-arch=gfx950 -O2 -o mix_order tools/ubench/mix_order.hip && ./mix_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 2000
#define REP 8
#define F(d, s) "v_and_b32 %" #d ", %" #s ", %16\n"
#define S(d, s) "v_bcnt_u32_b32 %" #d ", %" #s ", %" #d "\n"

template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
    uint32_t t0 = 1, t1 = 2, t2 = 3, t3 = 4, t4 = 5, t5 = 6, t6 = 7, t7 = 8;
    uint32_t x = threadIdx.x * 2654435761u + seed, m = seed | 0x55aa55aa;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#define ARGS : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), \
               "+v"(t4), "+v"(t5), "+v"(t6), "+v"(t7) : "v"(m), "v"(x)
            // counters c0..c7 = operands 0..7, temporaries t0..t7 = 8..15, m = 16, x = 17; F writes a temporary from
            // (x & m) -- never read by the S of the same block (S counts x), so everything is independent
            if (KIND == 0) asm volatile(F(8,17) S(0,17) F(9,17) S(1,17) F(10,17) S(2,17) F(11,17) S(3,17) F(12,17) S(4,17) F(13,17) S(5,17) F(14,17) S(6,17) F(15,17) S(7,17) ARGS);
            else if (KIND == 1) asm volatile(F(8,17) F(9,17) S(0,17) S(1,17) F(10,17) F(11,17) S(2,17) S(3,17) F(12,17) F(13,17) S(4,17) S(5,17) F(14,17) F(15,17) S(6,17) S(7,17) ARGS);
            else if (KIND == 2) asm volatile(F(8,17) F(9,17) F(10,17) F(11,17) S(0,17) S(1,17) S(2,17) S(3,17) F(12,17) F(13,17) F(14,17) F(15,17) S(4,17) S(5,17) S(6,17) S(7,17) ARGS);
            else if (KIND == 3) asm volatile(F(8,17) F(9,17) F(10,17) F(11,17) F(12,17) F(13,17) F(14,17) F(15,17) S(0,17) S(1,17) S(2,17) S(3,17) S(4,17) S(5,17) S(6,17) S(7,17) ARGS);
            else if (KIND == 4) asm volatile(F(8,17) F(9,17) F(10,17) F(11,17) F(12,17) F(13,17) F(14,17) F(15,17) F(8,17) F(9,17) F(10,17) F(11,17) F(12,17) F(13,17) F(14,17) F(15,17) ARGS);
            else if (KIND == 5) asm volatile(S(0,17) S(1,17) S(2,17) S(3,17) S(4,17) S(5,17) S(6,17) S(7,17) S(0,17) S(1,17) S(2,17) S(3,17) S(4,17) S(5,17) S(6,17) S(7,17) ARGS);
            // dependent forms: S counts the temporary the F before it wrote
            else if (KIND == 6) asm volatile(F(8,17) S(0,8) F(9,17) S(1,9) F(10,17) S(2,10) F(11,17) S(3,11) F(12,17) S(4,12) F(13,17) S(5,13) F(14,17) S(6,14) F(15,17) S(7,15) ARGS);
            else if (KIND == 7) asm volatile(F(8,17) F(9,17) F(10,17) F(11,17) S(0,8) S(1,9) S(2,10) S(3,11) F(12,17) F(13,17) F(14,17) F(15,17) S(4,12) S(5,13) S(6,14) S(7,15) ARGS);
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
    return ms * 1e-3 / ((double)w * ITER * REP * 16);
}

int main()
{
    uint32_t *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    const char *names[] = {"F S F S ... (independent)", "FF SS ...", "FFFF SSSS ...", "8F 8S", "F only", "S only", "F S dependent pairs", "FFFF SSSS dependent"};
    for (int w : {4, 8}) {
        const double t[] = {run<0>(out, w), run<1>(out, w), run<2>(out, w), run<3>(out, w), run<4>(out, w), run<5>(out, w), run<6>(out, w), run<7>(out, w)};
        for (int i = 0; i < 8; ++i)
            printf("waves/SIMD=%d  %-28s %.2f ns per wave-instruction per SIMD  (%.2f cycles @2.4GHz)\n", w, names[i], t[i] * 1e9, t[i] * 2.4e9);
    }
    return 0;
}
