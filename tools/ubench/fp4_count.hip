// Weighted bit counts of a 32-row tile word on the matrix cores, for ONE comparison individual.
//
// k_ld_popcount takes 25 (mask, count) pairs per tile word and lane: v_and_b32 + v_bcnt_u32_b32 for every weight plane of
// every sum.  v_mfma_scale_f32_16x16x128_f8f6f4 multiplies a 16 x 128 matrix A by a 128 x 16 matrix B; lane l holds 32
// K-elements of row (A) / column (B) l % 16, namely k = 32 (l / 16) .. +31.  Make A block diagonal:
//     A[m = 4 kb' + type][k = 32 kb + r] = weight_type[r]  if kb == kb'   else 0
// and let every lane supply ITS OWN tile word (bits -> FP4) as "column l % 16, K block l / 16".  Then
//     D[4 kb + type][n] = sum_r weight_type[r] * bit_r(word of lane n + 16 kb)
// and the C/D layout (column = lane % 16, rows 4 (lane / 16) .. +3 in the four registers) hands every lane the four
// weighted sums of its own word: no lane movement, one MFMA instead of 4 sums x 3 planes x 2 instructions.
//
// Bits -> FP4 (e2m1) without shifting: nibble 0001 = 0.5, 0010 = 1, 0100 = 2 (1000 = -0: useless), so
//     dword 0 = x & 0x11111111 (rows 4j, value 0.5)   dword 1 = x & 0x22222222 (rows 4j+1, 1.0)
//     dword 2 = x & 0x44444444 (rows 4j+2, 2.0)        dword 3 = (x >> 3) & 0x11111111 (rows 4j+3, 0.5)
// and A carries w, w/2, w/4, w in FP6 e2m3 (exact for w = 0..7): every product is w/2, the sum comes out halved, exact.
//
// This program (1) checks that layout with random words and weights against the host, (2) times the MFMA alone, the
// expansion alone and a segment-like mix (2 MFMAs + 2 expansions + the 7 instructions of the x0&x1 counts) at 1 and 8
// waves per SIMD, beside the 25 pairs of today's kernel.
//   hipcc --offload-arch=gfx950 -O3 -o fp4_count tools/ubench/fp4_count.hip && ./fp4_count
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// FP6 e2m3 code of w * scale for w = 0..7, scale = 1, 1/2, 1/4 (bias 1; exponent 0 = subnormal m/8)
static uint32_t e2m3(double v)
{
    if (v < 1.0)
        return (uint32_t)(v * 8.0);
    int e = 0;
    while (v >= 2.0 * (1 << e)) ++e;
    return (uint32_t)(((e + 1) << 3) | (int)((v / (1 << e) - 1.0) * 8.0));
}

__device__ __forceinline__ v8i expand(uint32_t x)
{
    v8i b = {0, 0, 0, 0, 0, 0, 0, 0};
    b[0] = x & 0x11111111u;
    b[1] = x & 0x22222222u;
    b[2] = x & 0x44444444u;
    b[3] = (x >> 3) & 0x11111111u;
    return b;
}

// (1) layout check: afrag[lane][6] as the host built it, words[lane]; out[lane][4]
__global__ void k_check(const uint32_t *afrag, const uint32_t *words, float *out)
{
    const uint32_t lane = threadIdx.x;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; ++i)
        a[i] = (int)afrag[lane * 6 + i];
    v8i b = expand(words[lane]);
    v4f c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 2 /* A: fp6 e2m3 */, 4 /* B: fp4 */, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int i = 0; i < 4; ++i)
        out[lane * 4 + i] = c[i];
}

// (2) rates.  MODE 0: MFMAs only (2 per turn); 1: the two expansions only; 2: a segment of the new form (2 expansions, 2 MFMAs,
// the x0&x1 counts); 3: today's 25 pairs
template <int MODE>
__global__ __launch_bounds__(512) void k_rate(const uint32_t *afrag, uint32_t *sink, int iters)
{
    const uint32_t lane = threadIdx.x & 63;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; ++i)
        a[i] = (int)afrag[lane * 6 + i];
    uint32_t x0 = lane * 0x9e3779b9u + blockIdx.x, x1 = x0 * 0x85ebca6bu + 1;
    v4f c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    uint32_t h0 = 0, h1 = 0, h2 = 0;
    uint32_t cnt[25];
    for (int i = 0; i < 25; ++i) cnt[i] = 0;
    const uint32_t m0 = afrag[0] | 0x01010101u, m1 = afrag[1] | 0x10101010u, m2 = afrag[2] | 0x00110011u;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            v8i b0 = {(int)x0, (int)x1, (int)x0, (int)x1, 0, 0, 0, 0};
            c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b0, c0, 2, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b0, c1, 2, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else if (MODE == 1) {
            v8i b0 = expand(x0), b1 = expand(x1);
            h0 += (uint32_t)(b0[0] ^ b0[1] ^ b0[2] ^ b0[3]);
            h1 += (uint32_t)(b1[0] ^ b1[1] ^ b1[2] ^ b1[3]);
        } else if (MODE == 2) {
            v8i b0 = expand(x0), b1 = expand(x1);
            c0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b0, c0, 2, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b1, c1, 2, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            const uint32_t hom = x0 & x1;
            h0 += __popc(hom & m0);
            h1 += __popc(hom & m1);
            h2 += __popc(hom & m2);
        } else {
            const uint32_t hom = x0 & x1;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const uint32_t m = k == 0 ? m0 : k == 1 ? m1 : m2;
                const uint32_t u0 = x0 & m, u1 = x1 & m;
                cnt[7 * k + 0] += __popc(u0);
                cnt[7 * k + 1] += __popc(u1);
                cnt[7 * k + 2] += __popc(hom & m);
                cnt[7 * k + 3] += __popc(u0 & m1);
                cnt[7 * k + 4] += __popc(u1 & m1);
                cnt[7 * k + 5] += __popc(u0 & m2);
                cnt[7 * k + 6] += __popc(u1 & m2);
            }
            cnt[21] += __popc(x0 & m1);
            cnt[22] += __popc(x1 & m1);
            cnt[23] += __popc(x0 & m2);
            cnt[24] += __popc(x1 & m2);
        }
        x0 = x0 * 1664525u + 1013904223u;          // (two more vector instructions per turn, in every mode)
        x1 ^= x0;
    }
    uint32_t r = h0 + h1 + h2;
    for (int i = 0; i < 25; ++i) r += cnt[i];
    for (int i = 0; i < 4; ++i) r += (uint32_t)c0[i] + (uint32_t)c1[i];
    if (r == 0x12345678u)
        sink[0] = r;
}

template <int MODE>
static void rate(const char *what, const uint32_t *d_a, uint32_t *d_sink, int waves_per_simd)
{
    const int iters = 20000;
    const int cus = 256;
    dim3 grid(cus), block(256 * waves_per_simd > 512 ? 512 : 256 * waves_per_simd);
    if (waves_per_simd == 8)
        grid = dim3(cus * 4);                      // 4 workgroups of 8 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_rate<MODE><<<grid, block>>>(d_a, d_sink, 100);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k_rate<MODE><<<grid, block>>>(d_a, d_sink, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // nominal 2.4 GHz; turns per SIMD = iters x waves per SIMD
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * waves_per_simd);
    printf("%-58s %d waves/SIMD: %7.1f cycles per turn and wave (at 2.4 GHz), %.3f ms\n", what, waves_per_simd, cyc, ms);
}

int main()
{
    // ---- (1) layout
    std::vector<uint32_t> w(4 * 32), words(64), afrag(64 * 6, 0);
    srand(7);
    for (auto &v : w) v = rand() % 8;
    for (auto &v : words) v = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    const double scale[4] = {1.0, 0.5, 0.25, 1.0};
    for (int lane = 0; lane < 64; ++lane) {
        const int m = lane % 16, kb = lane / 16;
        if (m / 4 != kb)
            continue;                       // zero row block
        const int type = m % 4;
        // element k = 8 d + j of the lane's 32  <->  tile row r = 4 j + d
        unsigned __int128 bits_lo = 0;      // 192 bits: low 128 here, high 64 below
        uint64_t bits_hi = 0;
        for (int k = 0; k < 32; ++k) {
            const int d = k / 8, j = k % 8, r = 4 * j + d;
            const uint32_t code = e2m3(w[type * 32 + r] * scale[d]);
            const int pos = 6 * k;
            if (pos < 128) {
                bits_lo |= (unsigned __int128)code << pos;
                if (pos + 6 > 128)
                    bits_hi |= (uint64_t)code >> (128 - pos);
            } else {
                bits_hi |= (uint64_t)code << (pos - 128);
            }
        }
        for (int i = 0; i < 4; ++i)
            afrag[lane * 6 + i] = (uint32_t)(bits_lo >> (32 * i));
        afrag[lane * 6 + 4] = (uint32_t)bits_hi;
        afrag[lane * 6 + 5] = (uint32_t)(bits_hi >> 32);
    }
    uint32_t *d_a, *d_w, *d_sink;
    float *d_out;
    CHECK(hipMalloc(&d_a, afrag.size() * 4)); CHECK(hipMalloc(&d_w, 64 * 4)); CHECK(hipMalloc(&d_out, 64 * 4 * 4));
    CHECK(hipMalloc(&d_sink, 4));
    CHECK(hipMemcpy(d_a, afrag.data(), afrag.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_w, words.data(), 64 * 4, hipMemcpyHostToDevice));
    k_check<<<1, 64>>>(d_a, d_w, d_out);
    std::vector<float> out(64 * 4);
    CHECK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane)
        for (int type = 0; type < 4; ++type) {
            uint32_t s = 0;
            for (int r = 0; r < 32; ++r)
                s += ((words[lane] >> r) & 1) * w[type * 32 + r];
            if (out[lane * 4 + type] != 0.5f * (float)s) {
                if (bad < 8)
                    printf("lane %d type %d: %g, expected %g\n", lane, type, out[lane * 4 + type], 0.5 * s);
                ++bad;
            }
        }
    printf("layout check (own word's four weighted sums back in the lane's own registers, halved): %s (%d of 256 wrong)\n",
           bad ? "FAILED" : "ok", bad);
    // ---- (2) rates
    for (int wps : {1, 8}) {
        rate<0>("2 MFMA 16x16x128 fp6 x fp4", d_a, d_sink, wps);
        rate<1>("2 expansions (6 and, 2 shift) + 6 xor/add", d_a, d_sink, wps);
        rate<2>("segment, new: 2 expansions + 2 MFMA + x0&x1 counts (7)", d_a, d_sink, wps);
        rate<3>("segment, today: 25 (and, bcnt) pairs", d_a, d_sink, wps);
    }
    return bad != 0;
}
