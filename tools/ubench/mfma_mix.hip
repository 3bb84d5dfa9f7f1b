// Micro-benchmark, part 5: can the matrix pipe take integer dot products off the vector ALU in a kernel
// that is bound by VALU issue?  Per iteration NM integer MFMAs (independent accumulators) and NV plain VALU
// instructions (v_and_b32 / v_bcnt_u32_b32 alternating), alone and together, 4 and 8 waves per SIMD.
//   KIND 0: v_mfma_i32_4x4x4_16b_i8   (16 blocks of 4x4x4: per lane 4 rows x 4 weight columns)
//   KIND 1: v_mfma_i32_16x16x64_i8    (gfx950)
//   KIND 2: v_mfma_i32_16x16x32_i8
//   hipcc --offload-arch=gfx950 -O2 -o mfma_mix tools/ubench/mfma_mix.hip && ./mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define ITER 2000

typedef int v4i __attribute__((ext_vector_type(4)));

template <int KIND, int NM, int NV>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    v4i acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        acc[i] = v4i{0, 0, 0, 0};
    uint32_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    uint32_t x = threadIdx.x * 2654435761u + seed, m = seed | 0x55aa55aa;
    v4i a4 = {(int)x, (int)(x * 3), (int)(x * 5), (int)(x * 7)}, b4 = {(int)m, (int)(m * 3), (int)(m * 5), (int)(m * 7)};
    long a8 = ((long)x << 32) | m, b8 = ((long)m << 32) | x;
    for (int it = 0; it < ITER; ++it) {
        constexpr int N = NM > NV ? NM : NV;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            // spread the MFMAs evenly among the vector instructions
            if (NM && (i * NM) / N != ((i + 1) * NM) / N) {
                const int j = ((i * NM) / N) & 7;
                if (KIND == 0)
                    acc[j] = __builtin_amdgcn_mfma_i32_4x4x4i8((int)x, (int)m, acc[j], 0, 0, 0);
                else if (KIND == 1)
                    acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a4, b4, acc[j], 0, 0, 0);
                else
                    acc[j] = __builtin_amdgcn_mfma_i32_16x16x32_i8(a8, b8, acc[j], 0, 0, 0);
            }
            if (NV && (i * NV) / N != ((i + 1) * NV) / N) {
                const int j = ((i * NV) / N);
                if (j & 1)
                    asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(c[(j >> 1) & 7]) : "v"(t[(j >> 1) & 7]));
                else
                    asm volatile("v_and_b32 %0, %1, %2" : "=v"(t[(j >> 1) & 7]) : "v"(x), "v"(m));
            }
        }
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        s += c[i] + t[i] + acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int NM, int NV>
double run(uint32_t *out, int w)
{
    const int blocks = 256 * w;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, NM, NV>), dim3(blocks), dim3(256), 0, 0, out, 1u);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, NM, NV>), dim3(blocks), dim3(256), 0, 0, out, 2u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // blocks of 4 waves, 256 CUs: w blocks per CU = w waves per SIMD; cycles per SIMD per iteration of ONE wave
    return ms * 1e-3 / ((double)w * ITER) * 2.4e9;
}

template <int KIND, int NM, int NV>
void report(uint32_t *out, const char *name)
{
    for (int w : {4, 8}) {
        const double m = run<KIND, NM, 0>(out, w), v = run<KIND, 0, NV>(out, w), b = run<KIND, NM, NV>(out, w);
        printf("%-24s waves/SIMD=%d  %2d MFMA alone %7.1f cyc (%.1f each)   %2d VALU alone %7.1f (%.2f each)   together %7.1f  (sum %.1f, max %.1f)\n",
               name, w, NM, m, m / NM, NV, v, v / NV, b, m + v, m > v ? m : v);
    }
}

int main()
{
    uint32_t *out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    report<0, 16, 26>(out, "4x4x4_16b_i8 16:26");
    report<0, 16, 52>(out, "4x4x4_16b_i8 16:52");
    report<0, 8, 48>(out, "4x4x4_16b_i8 8:48");
    report<1, 8, 36>(out, "16x16x64_i8 8:36");
    report<1, 8, 68>(out, "16x16x64_i8 8:68");
    report<1, 4, 48>(out, "16x16x64_i8 4:48");
    report<2, 8, 36>(out, "16x16x32_i8 8:36");
    report<2, 16, 68>(out, "16x16x32_i8 16:68");
    return 0;
}
