// Check of the operand / result layout of v_mfma_i32_32x32x32_i8 as k_ld_mfma (ibdg_ld_mfma.hip) uses it, with
// exact integer data and an ASYMMETRIC operand pair:
//   A (rows = target haplotypes): lane l holds 16 bytes of row l&31, k-slots (l>>5, j), j = 0..15
//   B (columns = background individuals): lane l holds 16 bytes of column l&31, the same k-slots
//   D: lane l, register r holds D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
// Which k a slot (l>>5, j) is does not matter as long as A and B agree (a dot product does not care about
// the order of its terms); the test fills slot (h, j) of both with k = 16h + j.
// Also: v_permlane32_swap (values of lanes 32..63 brought to lanes 0..31).
//   hipcc --offload-arch=gfx950 -O2 -o mfma_i8_layout tools/ubench/mfma_i8_layout.hip && ./mfma_i8_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__global__ void k(const int8_t *A, const int8_t *B, int *D, uint32_t *swapped)
{
    const unsigned l = threadIdx.x, r = l & 31, h = l >> 5;
    v4i a, b;
    for (int q = 0; q < 4; ++q) {
        uint32_t wa = 0, wb = 0;
        for (int j = 0; j < 4; ++j) {
            const int kk = 16 * h + 4 * q + j;
            wa |= (uint32_t)(uint8_t)A[r * 32 + kk] << (8 * j);     // A[row r][k]
            wb |= (uint32_t)(uint8_t)B[kk * 32 + r] << (8 * j);     // B[k][col r]
        }
        a[q] = (int)wa;
        b[q] = (int)wb;
    }
    v16i acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc, 0, 0, 0);
    for (int reg = 0; reg < 16; ++reg)
        D[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = acc[reg];
    // lanes 0..31 fetch the value of lane l+32
    uint32_t mine = 1000 + l, other = mine;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(other), "+v"(mine));
    // after the swap: `other` of lanes 32..63 <-> `mine` of lanes 0..31
    swapped[l] = mine;
    swapped[64 + l] = other;
}

int main()
{
    int8_t hA[1024], hB[1024];
    srand(7);
    for (int i = 0; i < 1024; ++i) {
        hA[i] = (int8_t)(rand() % 51);            // weights 0..50
        hB[i] = (int8_t)(rand() & 1);
    }
    int8_t *A, *B; int *D; uint32_t *S;
    (void)hipMalloc(&A, 1024); (void)hipMalloc(&B, 1024); (void)hipMalloc(&D, 4096); (void)hipMalloc(&S, 512);
    (void)hipMemcpy(A, hA, 1024, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hB, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, D, S);
    int hD[1024]; uint32_t hS[128];
    (void)hipMemcpy(hD, D, 4096, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hS, S, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            int ref = 0;
            for (int kk = 0; kk < 32; ++kk)
                ref += hA[i * 32 + kk] * hB[kk * 32 + j];
            bad += ref != hD[i * 32 + j];
        }
    printf("v_mfma_i32_32x32x32_i8 layout: %d of 1024 results differ\n", bad);
    printf("permlane32_swap: mine[0]=%u mine[31]=%u mine[32]=%u mine[63]=%u | other[0]=%u other[31]=%u other[32]=%u other[63]=%u\n",
           hS[0], hS[31], hS[32], hS[63], hS[64], hS[64 + 31], hS[64 + 32], hS[64 + 63]);
    return bad != 0;
}
