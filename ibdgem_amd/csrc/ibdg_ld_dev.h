// ibdg_ld_dev.h -- device helpers shared by the --LD kernels (ibdg_ld_popcount.hip, ibdg_ld_mfma.hip):
// tile words of the transposed panel, fixed-order wave sums, the single-instruction integer forms of the
// window end, LDS reads in assembly, the table product.
#pragma once
#include "ibdg_kernels.h"

#include <hip/hip_runtime.h>

namespace ibdg {

// the two haplotype words of one individual for one tile (wave-uniform address -> scalar load)
__device__ __forceinline__ uint2 tile_words(const uint4 *__restrict__ base, uint32_t tile)
{
    const uint2 *p = reinterpret_cast<const uint2 *>(base + (size_t)(tile >> 1) * 64);
    return p[tile & 1];
}

// v + (v of the lane selected by a DPP control): the cross-lane step of a wave reduction with
// data-parallel-primitive moves instead of ds_bpermute (no LDS traffic, no lane-index arithmetic).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int plo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, true);
    const int phi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, true);
    return v + __hiloint2double(phi, plo);
}

// Sum over the 64 lanes in a fixed order; the total ends up in lane 63.
__device__ __forceinline__ double wave_sum_to_lane63(double v)
{
    v = dpp_add<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);     // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);     // row_mirror: every lane of a 16-lane row holds the row total
    v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2 and 3
    return v;
}

#ifdef IBDG_DEBUG_TGT
#include <cstdio>
#define IBDG_CHECK_TGT(tgt, bound, where)                                                          \
    do {                                                                                           \
        if ((tgt) >= (bound)) {                                                                    \
            if ((threadIdx.x & 63) == 0)                                                           \
                printf("BAD TARGET %u (bound %u) in %s block %u,%u,%u\n", (unsigned)(tgt), (unsigned)(bound), where, blockIdx.x, blockIdx.y, blockIdx.z); \
            (tgt) = 0;                                                                             \
        }                                                                                          \
    } while (0)
#define IBDG_CHECK_IDX(idx, bound, where)                                                          \
    do {                                                                                           \
        if ((idx) >= (bound))                                                                      \
            printf("BAD INDEX %u (bound %u) in %s block %u,%u,%u\n", (unsigned)(idx), (unsigned)(bound), where, blockIdx.x, blockIdx.y, blockIdx.z); \
    } while (0)
#else
#define IBDG_CHECK_TGT(tgt, bound, where) do { } while (0)
#define IBDG_CHECK_IDX(idx, bound, where) do { } while (0)
#endif

// IBD0 of window w for comparison individual tgt from the ONE pass over the site list that keeps what does not depend on the
// comparison individual (src/ibdgem.c:714-715, :743: its own exclusion is all that does): the chunks' sums p2c[w][chunk][2]
// with, in place of the chunk the individual sits in, that chunk's 63 other weighted products p2w[w][lanes] -- masked, not
// subtracted: the own term can dominate the sum.  The additions are those of a launch that counts the IBD0 terms itself, in
// its order -- a chunk's 64 products meet in the balanced tree over the lane number's bits that wave_sum2 and
// wave_sum_to_lane63 share, the chunks' sums one lane a chunk and then in that tree again (k_ld_finalize) --, so the result is
// the same BITS whichever form a run took.  The total ends up in lane 63.
__device__ __forceinline__ double ibd0_from_pass(const double *__restrict__ p2c, const double *__restrict__ p2w, uint32_t lanes,
                                                 uint32_t n_chunks, uint32_t w, uint32_t tgt, uint32_t lane)
{
    const uint32_t c_own = tgt >> 6;
    const double *pc = p2c + (size_t)w * n_chunks * 2;
    double v = p2w[(size_t)w * lanes + 64 * (size_t)c_own + lane];
    v = wave_sum_to_lane63(lane == (tgt & 63) ? 0.0 : v);
    const double own = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                                        __builtin_amdgcn_readlane(__double2loint(v), 63));
    double t0 = 0.0;
    for (uint32_t c = lane; c < n_chunks; c += 64)
        t0 += c == c_own ? own : pc[2 * c];
    return wave_sum_to_lane63(t0);
}

// The same additions for lane 63 alone: without the row masks of the two row_bcast steps the other rows take sums nobody
// reads, and the steps need no zeroed destination (four v_mov less per sum).  Lane 63 holds the bits wave_sum_to_lane63 gives.
__device__ __forceinline__ double wave_sum_lane63_only(double v)
{
    v = dpp_add<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xf>(v);     // row_half_mirror
    v = dpp_add<0x140, 0xf>(v);     // row_mirror
    v = dpp_add<0x142, 0xf>(v);     // row_bcast:15: every row takes the total of the row before it (row 0: nothing)
    v = dpp_add<0x143, 0xf>(v);     // row_bcast:31: lane 31 holds rows 0 + 1 by now, row 3 rows 2 + 3
    return v;
}

// v + (v of lane ^ X within the 32-lane half) through the LDS crossbar (ds_swizzle, bit mode): no VALU
// move, no LDS memory -- the exchange is issued on the LDS port beside other waves' arithmetic.
template <int X>
__device__ __forceinline__ double swz_add(double v)
{
    constexpr int pat = (X << 10) | 0x1f;        // and 0x1f, or 0, xor X
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pat);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pat);
    return v + __hiloint2double(hi, lo);
}

// (a << SH) + b and a * M + c as the single instructions they are (v_lshl_add_u32, v_mad_i32_i24): the
// window end is a chain of these, and hipcc otherwise splits them into shifts and three-operand adds
// (56 integer instructions per window where 38 do).  a < 2^23 for the multiply (exponents are sums of
// at most a few thousand reads; the host does not offer this kernel beyond that).
template <int SH>
__device__ __forceinline__ uint32_t lshl_add(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(SH), "v"(b));
    return d;
}

template <int M>
__device__ __forceinline__ uint32_t mad24(uint32_t a, uint32_t c)
{
    static_assert(M >= -16 && M <= 64, "inline constants only; other multipliers go through mad24r");
    uint32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "n"(M), "v"(c));
    return d;
}

// the same with the multiplier in a register (-32 is not an inline constant)
__device__ __forceinline__ uint32_t mad24r(uint32_t a, uint32_t m, uint32_t c)
{
    uint32_t d;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(m), "v"(c));
    return d;
}

// sum_k v[k] << k by Horner's rule: KP-1 instructions
template <int KP>
__device__ __forceinline__ uint32_t planes_sum(const uint32_t (&v)[KP])
{
    uint32_t s = v[KP - 1];
#pragma unroll
    for (int k = KP - 2; k >= 0; --k)
        s = lshl_add<1>(s, v[k]);
    return s;
}

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ uint4 lds_read_b128(uint32_t addr)
{
    uint4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}

__device__ __forceinline__ uint2 lds_read_b64(uint32_t addr)
{
    uint2 v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}

__device__ __forceinline__ void lds_read2(uint4 &w0, uint4 &w1, uint32_t addr0, uint32_t addr1)
{
    asm volatile("ds_read_b128 %0, %2\n\t"
                 "ds_read_b128 %1, %3\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(w0), "=&v"(w1)
                 : "v"(addr0), "v"(addr1)
                 : "memory");
}

// eight power-table entries (16 B each) in one round trip (ds_read_b96 of the 12 bytes that matter was
// tried: 8-16 % slower in all three kernels)
__device__ __forceinline__ void lds_read_pow8(uint4 (&p)[8], const uint32_t (&ad)[8])
{
    asm volatile("ds_read_b128 %0, %8\n\t"
                 "ds_read_b128 %1, %9\n\t"
                 "ds_read_b128 %2, %10\n\t"
                 "ds_read_b128 %3, %11\n\t"
                 "ds_read_b128 %4, %12\n\t"
                 "ds_read_b128 %5, %13\n\t"
                 "ds_read_b128 %6, %14\n\t"
                 "ds_read_b128 %7, %15\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]),
                   "=&v"(p[7])
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7])
                 : "memory");
}

// rho^E2 * sigma^E3 * 2^eK from two table entries {m (2 words), e, pad}; the mantissa of K' is
// applied once per window in k_ld_finalize
__device__ __forceinline__ double ld_value(int eK, const uint4 &p1, const uint4 &p2)
{
    const double m1 = __hiloint2double((int)p1.y, (int)p1.x), m2 = __hiloint2double((int)p2.y, (int)p2.x);
    return __builtin_ldexp(m1 * m2, eK + (int)p1.z + (int)p2.z);
}

}  // namespace ibdg
