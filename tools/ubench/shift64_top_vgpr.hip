// Micro-benchmark behind DESIGN.md s4.4 (round 4): is a 64-bit vector shift whose shift amount sits in the LAST VGPR of
// the wave's allocation read wrongly on gfx950?  The site-preparation kernel that kept its rows' records in registers
// (24 VGPRs, the lane number in v23) produced, in some waves of workgroups that share a CU with others,
//     v_lshlrev_b64 v[0:1], v23, 1        ; 1 << lane
//     v_lshlrev_b64 v[14:15], v23, -1     ; -1 << lane      <- came out as -1 << (low word of v[0:1])
// i.e. the second shift took its amount from v0, the register "behind" v23 in a 24-register allocation (v24 wraps to v0).
// This program runs exactly that pair with the amount in v23 (the allocation's last register: the kernel clobbers
// nothing above it) and, as the control, in v22 with the same allocation; many workgroups per CU, optional barrier.
//   hipcc --offload-arch=gfx950 -O2 -o shift64_top_vgpr tools/ubench/shift64_top_vgpr.hip && ./shift64_top_vgpr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// the same pair with the amount in v15 of a 24-register allocation (index 7 mod 8, but not the last), and in v23 of a
// 32-register allocation (v31 clobbered): which of the two conditions is it?
template <int WHICH>
__global__ __launch_bounds__(256) void k2(unsigned long long *bad, unsigned long long *bad_like_v0, int iters)
{
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long wrong = 0, like_v0 = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t one, mask;
        if (WHICH == 0)
            asm volatile("v_mov_b32 v15, %2\n\t"
                         "v_lshlrev_b64 v[0:1], v15, 1\n\t"
                         "v_lshlrev_b64 %1, v15, -1\n\t"
                         "v_mov_b64 %0, v[0:1]"
                         : "=&v"(one), "=&v"(mask) : "v"(lane) : "v0", "v1", "v15", "v23");
        else
            asm volatile("v_mov_b32 v23, %2\n\t"
                         "v_lshlrev_b64 v[0:1], v23, 1\n\t"
                         "v_lshlrev_b64 %1, v23, -1\n\t"
                         "v_mov_b64 %0, v[0:1]"
                         : "=&v"(one), "=&v"(mask) : "v"(lane) : "v0", "v1", "v23", "v31");
        if (mask != (~0ull << lane)) {
            ++wrong;
            if (mask == (~0ull << ((uint32_t)one & 63)))
                ++like_v0;
        }
    }
    if (wrong) {
        atomicAdd(bad, wrong);
        atomicAdd(bad_like_v0, like_v0);
    }
}

// the siblings: v_lshrrev_b64 / v_ashrrev_i64 of the constant 0x8000000000000000 by the lane number held in v23 (last of 24)
template <int ARITH>
__global__ __launch_bounds__(256) void k3(unsigned long long *bad, unsigned long long *bad_like_v0, int iters)
{
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long wrong = 0, like_v0 = 0;
    const uint64_t top = 0x8000000000000000ull;
    for (int it = 0; it < iters; ++it) {
        uint64_t one, res;
        if (ARITH)
            asm volatile("v_mov_b32 v23, %2\n\t"
                         "v_lshlrev_b64 v[0:1], v23, 1\n\t"
                         "v_ashrrev_i64 %1, v23, %3\n\t"
                         "v_mov_b64 %0, v[0:1]"
                         : "=&v"(one), "=&v"(res) : "v"(lane), "v"(top) : "v0", "v1", "v22", "v23");
        else
            asm volatile("v_mov_b32 v23, %2\n\t"
                         "v_lshlrev_b64 v[0:1], v23, 1\n\t"
                         "v_lshrrev_b64 %1, v23, %3\n\t"
                         "v_mov_b64 %0, v[0:1]"
                         : "=&v"(one), "=&v"(res) : "v"(lane), "v"(top) : "v0", "v1", "v22", "v23");
        const uint64_t want = ARITH ? (uint64_t)((int64_t)top >> lane) : top >> lane;
        const uint32_t a0 = (uint32_t)one & 63;
        const uint64_t v0way = ARITH ? (uint64_t)((int64_t)top >> a0) : top >> a0;
        if (res != want) {
            ++wrong;
            if (res == v0way)
                ++like_v0;
        }
    }
    if (wrong) {
        atomicAdd(bad, wrong);
        atomicAdd(bad_like_v0, like_v0);
    }
}

template <int TOP, int BARRIER>
__global__ __launch_bounds__(256) void k(unsigned long long *bad, unsigned long long *bad_like_v0, int iters)
{
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long wrong = 0, like_v0 = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t one, mask;
        if (TOP)
            asm volatile("v_mov_b32 v23, %2\n\t"
                         "v_lshlrev_b64 v[0:1], v23, 1\n\t"
                         "v_lshlrev_b64 %1, v23, -1\n\t"
                         "v_mov_b64 %0, v[0:1]"
                         : "=&v"(one), "=&v"(mask) : "v"(lane) : "v0", "v1", "v22", "v23");
        else
            asm volatile("v_mov_b32 v22, %2\n\t"
                         "v_lshlrev_b64 v[0:1], v22, 1\n\t"
                         "v_lshlrev_b64 %1, v22, -1\n\t"
                         "v_mov_b64 %0, v[0:1]"
                         : "=&v"(one), "=&v"(mask) : "v"(lane) : "v0", "v1", "v22", "v23");
        if (mask != (~0ull << lane)) {
            ++wrong;
            if (mask == (~0ull << ((uint32_t)one & 63)))
                ++like_v0;
        }
        if (BARRIER)
            __syncthreads();
    }
    if (wrong) {
        atomicAdd(bad, wrong);
        atomicAdd(bad_like_v0, like_v0);
    }
}

template <int TOP, int BARRIER>
static void run(const char *what, unsigned long long *d)
{
    (void)hipMemset(d, 0, 16);
    int nv = 0;
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k<TOP, BARRIER>));
    nv = fa.numRegs;
    hipLaunchKernelGGL((k<TOP, BARRIER>), dim3(256 * 16), dim3(256), 0, 0, d, d + 1, 2000);
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-44s VGPRs %d: %llu wrong masks of %llu (%llu of them = -1 << low6(v0))\n", what, nv, h[0],
           256ull * 16 * 256 * 2000, h[1]);
}

template <int WHICH>
static void run2(const char *what, unsigned long long *d)
{
    (void)hipMemset(d, 0, 16);
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k2<WHICH>));
    hipLaunchKernelGGL((k2<WHICH>), dim3(256 * 16), dim3(256), 0, 0, d, d + 1, 2000);
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-44s VGPRs %d: %llu wrong masks of %llu (%llu of them = -1 << low6(v0))\n", what, fa.numRegs, h[0],
           256ull * 16 * 256 * 2000, h[1]);
}

template <int ARITH>
static void run3(const char *what, unsigned long long *d)
{
    (void)hipMemset(d, 0, 16);
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(k3<ARITH>));
    hipLaunchKernelGGL((k3<ARITH>), dim3(256 * 16), dim3(256), 0, 0, d, d + 1, 2000);
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-44s VGPRs %d: %llu wrong results of %llu (%llu of them = shifted by low6(v0))\n", what, fa.numRegs, h[0],
           256ull * 16 * 256 * 2000, h[1]);
}

int main()
{
    unsigned long long *d;
    (void)hipMalloc(&d, 16);
    run<1, 0>("amount in v23 (last of the allocation)", d);
    run<0, 0>("amount in v22 (control)", d);
    run<1, 1>("amount in v23, barrier per turn", d);
    run<0, 1>("amount in v22, barrier per turn (control)", d);
    run2<0>("amount in v15, 24 registers allocated", d);
    run2<1>("amount in v23, 32 registers allocated", d);
    run3<0>("v_lshrrev_b64, amount in v23 (last of 24)", d);
    run3<1>("v_ashrrev_i64, amount in v23 (last of 24)", d);
    (void)hipFree(d);
    return 0;
}
