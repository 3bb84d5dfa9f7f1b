// ibdg_ld_mfma.hip -- the --LD loop for MANY comparison individuals against one panel (BASELINE.json
// configs[4]: hundreds of them in one run): the sums that depend on the comparison individual as integer
// matrix products on the matrix cores.
//
// Of the nine weighted popcounts per background individual and window (header of ibdg_ld_popcount.hip) four
// depend on the comparison individual t:
//     G(x,t) = <x & t, cov> = sum over the window's rows r of  cov_r t_r x_r       (x in {x0,x1}, t in {t0,t1})
// For ONE comparison individual that is element-wise work (and + popcount, k_ld_popcount).  For T of them it
// is a product of two matrices over the rows of the window,
//     G[2T target haplotypes][2N background haplotypes] = (cov_r t_r)[2T][rows] . (x_r)[rows][2N],
// i.e. a dense integer contraction -- the one place on this path where the matrix cores apply.  One
// v_mfma_i32_32x32x32_i8 takes the 32 rows of a tile as K: 32 target-haplotype rows (15 comparison
// individuals and one pair of rows that carries the weights themselves, which gives C(x) and A(x) of the
// header's algebra) against 32 background individuals, 32 768 multiply-adds in the time of ~9 vector
// instructions, where the counting kernels spend 24 vector instructions per comparison individual on the
// same tile.  The integers are the same integers as the counting kernels'; what follows them (the window
// end below) is arranged for many comparison individuals per background individual.
//
// Work split: a wave = 32 background individuals (half a chunk of the transposed panel) x one run of
// windows x one group of IBDG_TG = 15 comparison individuals; 4 waves per workgroup.
//   per segment ((window, 32-row tile), as in k_ld_popcount):
//     B operand: the lane's individual's two tile words, 16 rows per lane half, bit -> byte by a shift and the
//                mask 0x01010101 per dword (rows 4 kb + d + 8 j; a 256-entry table in LDS was slower);
//     A operand: 16 bytes per lane of the segment's target image (k_win_target_g: weight of row r for target
//                haplotype m, rows outside the window 0), a coalesced 1 KiB load two segments ahead;
//     two MFMAs (x0 and x1) accumulate into 2 x 16 registers;  C(x0 & x1) by and + popcount as before.
//   per window: target-haplotype row m of the operand is haplotype (m >> 2) & 1 of slot (m & 3) + 4 (m >> 3), so
//     result register r of lane half h holds G(x, t_h) of comparison individual r for the lane's background
//     individual: a lane finishes its individual against ONE haplotype of all 16 slots, the two halves of the wave
//     the two haplotypes.  The B operand carries 16 for a set bit, so a register IS 16 G(x,t) -- the byte offset of
//     tau^G in a table of plain doubles that sits at LDS address 0: no instruction forms a look-up address (a window whose
//     powers leave the double range takes the {mantissa, exponent} table beside it).
//     The exponents of a product are (header of ibdg_ld_popcount.hip)
//         E2 = AT - <t,alt> - A(x) + G(x,t),   E3 = <t,cov> + C(x) - 2 G(x,t),
//     so with tau = rho / sigma^2 = 4 eps (1-eps) the product factors into three parts,
//         K' rho^E2 sigma^E3 = [K' rho^(AT - A(x)) sigma^C(x)] . [rho^-<t,alt> sigma^<t,cov>] . tau^G(x,t)
//                            =            V_x                 .            U_t              . tau^G,
//     V_x once per (background haplotype, window) in the lane, U_t once per (comparison haplotype, window) from
//     k_win_slot_g, and ONE table look-up per product (the unfactored form needs two and eight address
//     computations per comparison individual): Q(x0,t) + Q(x1,t) = mU_t . (V_x0 tau^G0 + V_x1 tau^G1 scaled by
//     2^eU_t), everything as mantissa and integer exponent until one ldexp per product.  mU_t does not depend on the
//     background individual, so it is applied AFTER the sum over the background: a lane's addend for slot r is
//     ldexp(mV0 mtau0, eV0 + etau0 + eU) + ldexp(mV1 mtau1, eV1 + etau1 + eU) -- three vector instructions per
//     product -- and the two haplotypes' sums meet as mU_t0 S_0 + mU_t1 S_1 in the lanes that store.  A lane's
//     addends go to the wave's LDS strip, eight slots per turn, and the sums over the 32 lanes of each half come
//     from reading the strip transposed (eight values per lane, then two exchange steps): 13 vector instructions
//     where a reduction in registers took 51.  The two halves of a chunk are added by k_ld_finalize
//     (halves = 1).  The association of the products and the order of the sums differ from the counting kernels',
//     so the results agree with theirs (and the oracle's) to ~1e-14, not bit for bit; the bar is 1e-10.
//     Nothing in the loops goes through the scalar memory path: the run's segment records are staged into
//     LDS once per workgroup (a scalar load per segment cost more than the segment's arithmetic).
// Operand / result layout of the instruction: tools/ubench/mfma_i8_layout.hip (checked on the device).
#include "ibdg_kernels.h"
#include "ibdg_ld_dev.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <type_traits>

namespace ibdg {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr uint32_t TG = IBDG_TG;     // comparison individuals per group
constexpr uint32_t PSEUDO = 15;      // the slot whose two rows are the weights: row 30 = cov, row 31 = alt

// Result register r of lane half h holds row (r & 3) + 8 (r >> 2) + 4 h of the product (tools/ubench/
// mfma_i8_layout.hip).  Operand row m is therefore given to
//     target haplotype (m >> 2) & 1 of slot (m & 3) + 4 (m >> 3):
// register r = slot r, lane half = haplotype.
__device__ __forceinline__ uint32_t row_slot(uint32_t m) { return (m & 3) + 4 * (m >> 3); }
__device__ __forceinline__ uint32_t row_hap(uint32_t m) { return (m >> 2) & 1; }

// ---------------------------------------------------------------------------
// Per group of comparison individuals: (1) the A operand of every segment,
//     aimg[group][segment][lane l] = 16 bytes: row m = l & 31 (slot row_slot(m), target haplotype row_hap(m)),
//     byte j of dword d = weight of tile row 4 (l >> 5) + d + 8 j (the order in which a shift and the mask
//     0x01010101 take the background bits out of a tile word):  cov_r where the target haplotype carries the
//     alt allele on a row of the segment, else 0; slot 15: cov_r (row 27) and alt_r (row 31) themselves;
// (2) per window and slot the four table offsets 16<t0,cov> 16<t1,cov> 16(AT-<t0,alt>) 16(AT-<t1,alt>)
//     (k_win_target's window constants).
// ---------------------------------------------------------------------------
// (a wave takes IBDG_WTG_SEGS consecutive segments: the slots' individuals and their tile pointers once, the segments'
//  tile words all requested before the first is used)
#ifndef IBDG_WTG_SEGS
#define IBDG_WTG_SEGS 4
#endif
__global__ __launch_bounds__(256) void k_win_target_g(MfmaArgs a)
{
    constexpr uint32_t NSEG = IBDG_WTG_SEGS;
    const unsigned grp = blockIdx.y;
    const uint32_t left = a.n_targets - grp * TG, cnt = left < TG ? left : TG;
    const size_t wv = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t s0 = __builtin_amdgcn_readfirstlane((uint32_t)(wv * NSEG));
    if (s0 >= a.n_segs)
        return;
    const uint32_t l = threadIdx.x & 63, m = l & 31, kb = l >> 5, q = row_slot(m), th = row_hap(m);
    const bool use_alt = q == PSEUDO && th;
    const bool real = q != PSEUDO && q < cnt;
    const uint4 *tt = nullptr;
    if (real) {
        const uint32_t tgt = a.targets[a.t_base + grp * TG + q];
        tt = reinterpret_cast<const uint4 *>(a.t32) + (size_t)(tgt >> 6) * a.n_pairs * 64 + (tgt & 63);
    }
    uint32_t selw[NSEG];
#pragma unroll
    for (uint32_t j = 0; j < NSEG; ++j) {
        const uint32_t s = s0 + j < a.n_segs ? s0 + j : a.n_segs - 1;
        IBDG_CHECK_IDX(a.segs[s].tile, 2 * a.n_pairs, "k_win_target_g tile");
        uint32_t sel = q == PSEUDO ? 0xffffffffu : 0u;     // rows of the tile that count in this row of the operand
        if (real) {
            const uint2 w = tile_words(tt, a.segs[s].tile);
            sel = th ? w.y : w.x;
        }
        selw[j] = sel >> (4 * kb);
    }
#pragma unroll
    for (uint32_t j = 0; j < NSEG; ++j) {
        const uint32_t s = s0 + j;
        if (s >= a.n_segs)
            break;
        const Seg &S = a.segs[s];
        const uint32_t sel = selw[j];
        const uint32_t ncov = (S.flags >> 16) & 0xff, nalt = S.flags >> 24, np = ncov > nalt ? ncov : nalt;
        // Where no row of the tile has 16 reads or more (nearly every segment) dword d of the image carries the weights times
        // 8 >> d: the kernel then takes the background bits of its B operand with four masks and no shifts -- values 1, 2, 4, 8
        // by dword -- and every product is 8 w all the same (8 x 15 fits the signed byte).  Otherwise the plain weights, and
        // the kernel shifts every bit to the value 8 (seg_wide).
        const bool wide = np > 4;
        uint32_t out[4] = {0, 0, 0, 0};
        // (all sixteen masks at once, wave-uniform: loaded inside the loop over the planes they cost a memory round trip each)
        uint32_t pl[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
            pl[k] = use_alt ? S.alt[k] : S.cov[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if ((uint32_t)k < np) {
                const uint32_t f = (pl[k] >> (4 * kb)) & sel;
#pragma unroll
                for (int d = 0; d < 4; ++d)
                    out[d] += ((f >> d) & 0x01010101u) << (wide ? k : k + 3 - d);       // weights <= 127 (<= 120): no carry between bytes
            }
        }
        a.aimg[((size_t)grp * a.n_segs + s) * 64 + l] = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

// (2) of the above: a wave per window, lane l = slot l & 15, segments l >> 4, l >> 4 + 4, ... of the window.
// U_t = rho^-<t,alt> sigma^<t,cov> of the slot's two haplotypes as {mantissa, exponent}: the quotient of two
// table mantissas (a correctly rounded division) and the difference of their exponents.
__global__ __launch_bounds__(256) void k_win_slot_g(MfmaArgs a)
{
    const unsigned grp = blockIdx.y;
    const uint32_t left = a.n_targets - grp * TG, cnt = left < TG ? left : TG;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (w >= a.n_win)
        return;
    const uint32_t l = threadIdx.x & 63, q = l & 15;
    uint32_t a0cov = 0, a1cov = 0, a0alt = 0, a1alt = 0;
    const bool real = q < cnt && q != PSEUDO;
    if (real) {
        const uint32_t tgt = a.targets[a.t_base + grp * TG + q];
        const uint4 *tt = reinterpret_cast<const uint4 *>(a.t32) + (size_t)(tgt >> 6) * a.n_pairs * 64 + (tgt & 63);
        const uint32_t s1 = a.wconst[w + 1].seg_begin;
        for (uint32_t s = a.wconst[w].seg_begin + (l >> 4); s < s1; s += 4) {
            const Seg S = a.segs[s];                 // the whole record at once (five 16-byte loads in flight)
            const uint2 at = tile_words(tt, S.tile);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a0cov += (uint32_t)__popc(at.x & S.cov[k]) << k;
                a1cov += (uint32_t)__popc(at.y & S.cov[k]) << k;
                a0alt += (uint32_t)__popc(at.x & S.alt[k]) << k;
                a1alt += (uint32_t)__popc(at.y & S.alt[k]) << k;
            }
        }
    }
#pragma unroll
    for (int m = 16; m < 64; m <<= 1) {
        a0cov += __shfl_xor(a0cov, m);
        a1cov += __shfl_xor(a1cov, m);
        a0alt += __shfl_xor(a0alt, m);
        a1alt += __shfl_xor(a1alt, m);
    }
    // the largest <t,cov> of the group's haplotypes in this window: no G(x,t) = <x & t, cov> of the window exceeds it
    uint32_t gmax = real ? (a0cov > a1cov ? a0cov : a1cov) : 0u;
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
        const uint32_t o = __shfl_xor(gmax, m);
        gmax = o > gmax ? o : gmax;
    }
    if (l < 16) {
        // per window 32 uint4: [0..3] the exponents of U_t0 of slots 0..15, [4..7] of U_t1, [8 + q] the two mantissas of slot q
        // (slot 15 holds no individual: its U_t0 exponent entry carries gmax)
        uint4 um = make_uint4(0, 0, 0, 0);                // empty slots: U = 0
        int e0 = q == PSEUDO ? (int)gmax : 0, e1 = 0;
        if (real) {
            const PowEntry r0 = a.pow_1me[a0alt], r1 = a.pow_1me[a1alt], s0 = a.pow_eps[a0cov], s1 = a.pow_eps[a1cov];
            const double m0 = s0.m / r0.m, m1 = s1.m / r1.m;
            um = make_uint4((uint32_t)__double2loint(m0), (uint32_t)__double2hiint(m0), (uint32_t)__double2loint(m1),
                            (uint32_t)__double2hiint(m1));
            e0 = s0.e - r0.e;
            e1 = s1.e - r1.e;
        }
        uint4 *dst = a.wc_slot + ((size_t)grp * a.n_win + w) * 32;
        dst[8 + q] = um;
        reinterpret_cast<int *>(dst)[q] = e0;
        reinterpret_cast<int *>(dst)[16 + q] = e1;
    }
}

// the value of lane 32 + (l & 31) in every lane (v_permlane32_swap: the upper half of the first operand and
// the lower half of the second change places; the builtin, so that hipcc pads the wait states the
// instruction needs behind the write of its operand -- an asm statement read stale registers)
__device__ __forceinline__ uint32_t from_upper_half(uint32_t v)
{
    return __builtin_amdgcn_permlane32_swap(v, v, false, false)[1];
}

// The partner's value of a 64-bit quantity (lane ^ X within the 32-lane half): for X = 1, 2 a DPP move within
// the quad (vector ALU, which has slack here), beyond that through the LDS crossbar (the port this kernel is
// bound by)
template <int X>
__device__ __forceinline__ double swz_get(double v)
{
    if (X == 1 || X == 2) {
        constexpr int ctrl = X == 1 ? 0xB1 : 0x4E;          // quad_perm [1,0,3,2] / [2,3,0,1]
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true);
        return __hiloint2double(hi, lo);
    }
    constexpr int pat = (X << 10) | 0x1f;
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), pat);
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), pat);
    return __hiloint2double(hi, lo);
}

// An 8-byte store by the lanes of `mask` only, as ONE asm statement (EXEC narrowed and put back inside it).  On purpose out of
// hipcc's sight: a store under a C++ condition is a branch, behind which hipcc cannot count the vector-memory operations in
// flight any more and drains them all (s_waitcnt vmcnt(0)) at the next use of a requested operand -- round 4 therefore
// drained the requests at the START of every window end, where the last one had just been issued.  An operation hipcc does
// not know of only makes its counted waits wait for more, never for less (the queue retires in order).
__device__ __forceinline__ void store_f64_lanes(uint64_t mask, double *p, double v)
{
    uint64_t exec_was;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_and_b64 exec, %0, %3\n\t"
                 "global_store_dwordx2 %1, %2, off\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(exec_was)
                 : "v"(p), "v"(v), "s"(mask)
                 : "memory", "scc");
}

// the same into LDS (the workgroup's sums of a window, MfmaArgs::wg_sum)
__device__ __forceinline__ void lds_store_f64_lanes(uint64_t mask, uint32_t addr, double v)
{
    uint64_t exec_was;
    asm volatile("s_mov_b64 %0, exec\n\t"
                 "s_and_b64 exec, %0, %3\n\t"
                 "ds_write_b64 %1, %2\n\t"
                 "s_mov_b64 exec, %0"
                 : "=&s"(exec_was)
                 : "v"(addr), "v"(v), "s"(mask)
                 : "memory", "scc");
}

constexpr uint32_t SS = 36;          // doubles per row of the reduction strip: 32 lanes + 4 of padding (reads two-way at most)

// (a << 4) + b with b in a scalar register (the tau table's LDS address is the same for the whole workgroup)
__device__ __forceinline__ uint32_t lshl4_add_s(uint32_t a, uint32_t b)
{
    uint32_t d;
    asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(d) : "v"(a), "s"(b));
    return d;
}

__device__ __forceinline__ void reg_swap(uint32_t &a, uint32_t &b)
{
    asm volatile("v_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// the two operand slots change places
__device__ __forceinline__ void slots_swap(uint2 &x0, uint4 &a0, uint2 &x1, uint4 &a1)
{
    reg_swap(x0.x, x1.x);
    reg_swap(x0.y, x1.y);
    reg_swap(a0.x, a1.x);
    reg_swap(a0.y, a1.y);
    reg_swap(a0.z, a1.z);
    reg_swap(a0.w, a1.w);
}

// Four slots of the lane (registers R0 .. R0 + 3 of the two accumulators = the byte offsets of tau^G(x0,t), tau^G(x1,t)
// in the table at LDS address TAU0) per LDS round trip: eight reads in one statement.  At four waves per SIMD a
// round trip is not hidden by other waves.  out[j] = V_x0 tau^G0 + V_x1 tau^G1 scaled by 2^eU of slot R0 + j: the
// reference's Q(x0,t) + Q(x1,t) of ibdgem.c:716-719 without the mantissa of U_t (applied after the sum over the
// background, which it does not depend on).
template <int R0>
__device__ __forceinline__ void comp_quad(const v16i &acc0, const v16i &acc1, uint32_t eu_addr, uint32_t tau16, double mV0,
                                          double mV1, int eV0, int eV1, double (&out)[4])
{
    // the general form: tau^G as {mantissa, exponent}, 16 bytes per entry at tau16 + 2 x (8 G)
    uint4 p[8], eu;              // eu: the exponents of U of the four slots (wave-half uniform address: a broadcast read)
    uint32_t ad[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(ad[2 * j]) : "v"(acc0[R0 + j]), "s"(tau16));
        asm("v_lshl_add_u32 %0, %1, 1, %2" : "=v"(ad[2 * j + 1]) : "v"(acc1[R0 + j]), "s"(tau16));
    }
    asm volatile("ds_read_b128 %0, %9\n\t"
                 "ds_read_b128 %1, %10\n\t"
                 "ds_read_b128 %2, %11\n\t"
                 "ds_read_b128 %3, %12\n\t"
                 "ds_read_b128 %4, %13\n\t"
                 "ds_read_b128 %5, %14\n\t"
                 "ds_read_b128 %6, %15\n\t"
                 "ds_read_b128 %7, %16\n\t"
                 "ds_read_b128 %8, %17 offset:%18\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]), "=&v"(eu)
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]), "v"(eu_addr),
                   "n"(4 * R0)
                 : "memory");
    const int e[4] = {(int)eu.x, (int)eu.y, (int)eu.z, (int)eu.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint4 &p0 = p[2 * j], &p1 = p[2 * j + 1];
        const double q0 = __builtin_ldexp(mV0 * __hiloint2double((int)p0.y, (int)p0.x), eV0 + (int)p0.z + e[j]);
        const double q1 = __builtin_ldexp(mV1 * __hiloint2double((int)p1.y, (int)p1.x), eV1 + (int)p1.z + e[j]);
        out[j] = q0 + q1;
    }
}

// The same for a window all of whose powers tau^G are ordinary doubles (G <= the window's reads, and tau^reads has not
// left the double range: nearly every window): the look-up is 8 bytes at LDS address 8 G -- the accumulator itself -- and
// the product's exponent is the lane's and the slot's alone.  mV tau^G = (mV mtau) 2^etau exactly, so the values are the
// bits of the general form; the LDS port, the busiest unit of this kernel, moves half the bytes.
// The lane's two haplotypes share the slot's exponent, so they are brought to ONE exponent first (mVa = mV0 2^(eV0 - eR),
// mVb = mV1 2^(eV1 - eR), eR the larger of eV0, eV1: what falls out of the double range on the way is < 2^-1000 of the
// other addend) and a slot costs two products, one addition and one ldexp -- 5 vector instructions where two separately
// scaled products cost 7.  Safe: the addend of the larger exponent keeps its mantissa (>= 1/4) and every plain power is
// >= 2^-1000, so that product is >= 2^-1002; whatever the other product loses to the double range on its way down is
// < 2^-1074, i.e. < 2^-72 of the sum.  (The rounding differs from the general form's in the last place; the bar is 1e-10.)
// Round 5: NO exponent in the lane's work per slot.  The slot's exponent eU is the same for every lane of the half, so it is
// applied after the sum over the lanes (with U's mantissa, by the lanes that store); and the lanes' own exponents are brought to
// ONE for the whole wave first -- eRef = the largest eR of a lane with a non-zero multiplicity; mVa, mVb arrive here scaled by
// 2^(eR - eRef) -- so that the 32 addends of a slot add up as plain doubles: out[j] = mVa tau^G0 + mVb tau^G1 (a product and a
// fused multiply-add: 2 vector instructions per slot where the form above took 5), S = sum over the lanes, and the slot's value is
// mU S 2^(eRef + eU).  Safe because the window qualifies for this path only when tau^gmax >= 2^-480 (gmax = the largest <t,cov>
// of the group's haplotypes in the window, k_win_slot_g: no G exceeds it): the lane that sets eRef contributes at least 2^-482 to
// every slot's sum, and whatever a lane's scaling or product loses to the double range is below 2^-1022, i.e. < 2^-540 of the sum.
template <int R0>
__device__ __forceinline__ void comp_quad_fast(const v16i &acc0, const v16i &acc1, double mVa, double mVb, double (&out)[4])
{
    double p[8];
    asm volatile("ds_read_b64 %0, %8\n\t"
                 "ds_read_b64 %1, %9\n\t"
                 "ds_read_b64 %2, %10\n\t"
                 "ds_read_b64 %3, %11\n\t"
                 "ds_read_b64 %4, %12\n\t"
                 "ds_read_b64 %5, %13\n\t"
                 "ds_read_b64 %6, %14\n\t"
                 "ds_read_b64 %7, %15\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7])
                 : "v"(acc0[R0]), "v"(acc1[R0]), "v"(acc0[R0 + 1]), "v"(acc1[R0 + 1]), "v"(acc0[R0 + 2]), "v"(acc1[R0 + 2]),
                   "v"(acc0[R0 + 3]), "v"(acc1[R0 + 3])
                 : "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j)
        out[j] = __builtin_fma(mVa, p[2 * j], mVb * p[2 * j + 1]);
}

// the largest value of v over the wave's lanes, in a scalar register: four exchange-and-max steps within the rows of 16 (DPP),
// then the four rows' maxima through v_readlane
__device__ __forceinline__ int wave_max_i32(int v)
{
#define IBDG_MAX_STEP(CTRL)                                                        \
    {                                                                              \
        const int o = __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);    \
        v = o > v ? o : v;                                                         \
    }
    IBDG_MAX_STEP(0xB1)      // quad_perm [1,0,3,2]
    IBDG_MAX_STEP(0x4E)      // quad_perm [2,3,0,1]
    IBDG_MAX_STEP(0x141)     // row_half_mirror
    IBDG_MAX_STEP(0x140)     // row_mirror
#undef IBDG_MAX_STEP
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const int ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// addend I of the lane into the wave's reduction strip: row (8 h + I), column n
template <int I>
__device__ __forceinline__ void strip_put(uint32_t put_addr, double v)
{
    asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(put_addr), "v"(v), "n"(I * (int)SS * 8) : "memory");
}

// Lane L = 4 s + p adds the elements p, p + 4, ..., p + 28 of row s (a wave's LDS operations execute in order: the
// reads see the writes of strip_put); two exchange steps within the quad finish the row.  Every lane of quad s
// ends up with the sum of row s = the sum over the 32 lanes of half s >> 3 of their addend s & 7.
// the same with one more read in the round trip: the lane's U mantissa (8 bytes at mu_addr)
// ... and its U exponent (4 bytes at eu_addr)
__device__ __forceinline__ double strip_sum_mu(uint32_t get_addr, uint32_t mu_addr, uint32_t eu_addr, double &mu, int &eu)
{
    double r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile("ds_read_b64 %0, %10\n\t"
                 "ds_read_b64 %1, %10 offset:32\n\t"
                 "ds_read_b64 %2, %10 offset:64\n\t"
                 "ds_read_b64 %3, %10 offset:96\n\t"
                 "ds_read_b64 %4, %10 offset:128\n\t"
                 "ds_read_b64 %5, %10 offset:160\n\t"
                 "ds_read_b64 %6, %10 offset:192\n\t"
                 "ds_read_b64 %7, %10 offset:224\n\t"
                 "ds_read_b64 %8, %11\n\t"
                 "ds_read_b32 %9, %12\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7), "=&v"(mu), "=&v"(eu)
                 : "v"(get_addr), "v"(mu_addr), "v"(eu_addr)
                 : "memory");
    double t = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    t = t + swz_get<1>(t);
    t = t + swz_get<2>(t);
    return t;
}

__device__ __forceinline__ double strip_sum(uint32_t get_addr)
{
    double r0, r1, r2, r3, r4, r5, r6, r7;
    asm volatile("ds_read_b64 %0, %8\n\t"
                 "ds_read_b64 %1, %8 offset:32\n\t"
                 "ds_read_b64 %2, %8 offset:64\n\t"
                 "ds_read_b64 %3, %8 offset:96\n\t"
                 "ds_read_b64 %4, %8 offset:128\n\t"
                 "ds_read_b64 %5, %8 offset:160\n\t"
                 "ds_read_b64 %6, %8 offset:192\n\t"
                 "ds_read_b64 %7, %8 offset:224\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                 : "v"(get_addr)
                 : "memory");
    double t = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    t = t + swz_get<1>(t);
    t = t + swz_get<2>(t);
    return t;
}

// cache policy of the tile-word loads (aux of the buffer load: 0 default, 2 = nt): a group streams its chunk's tile words once
#ifndef IBDG_MFMA_TILE_AUX
#define IBDG_MFMA_TILE_AUX 0
#endif
#ifndef IBDG_MFMA_AHEAD
#define IBDG_MFMA_AHEAD 2           /* operand slots = segments a request runs ahead of its use.  3 (round 5, measured at
                                       compile time only): the three window instances' slots meet in phis the register
                                       allocator resolves with copies -- 56 spilled VGPRs, a full drain at every copy */
#endif
#ifndef IBDG_MFMA_DRAIN
#define IBDG_MFMA_DRAIN 0           /* 1: round 4's wait for every request at the start of a window end */
#endif
#ifndef IBDG_MFMA_XCD
#define IBDG_MFMA_XCD 1             /* the workgroups of a run on one XCD (0: in launch order round the XCDs) */
#endif
#ifndef IBDG_MFMA_GRP_MAJOR
#define IBDG_MFMA_GRP_MAJOR 0       /* 1: within a run the groups of individuals in turn, within those the half-chunk groups */
#endif
#ifndef IBDG_MFMA_WAVES
#define IBDG_MFMA_WAVES 8           /* waves = half chunks per workgroup: 8 share one copy of the tables (4: the strips'
                                       LDS leaves room for 3 waves per SIMD only, 12 % slower) */
#endif
// LDS per workgroup: window constants (16 B + 16 slots x 32 B per window), the three power tables,
// the run's segment records, a reduction strip per wave
size_t ld_mfma_lds_bytes(uint32_t win_per_group, uint32_t tab_len, uint32_t max_seg)
{
    return (size_t)win_per_group * 33 * 16 + (size_t)tab_len * 56 + 16 + ((size_t)max_seg + 1) * 32 +
           (size_t)IBDG_MFMA_WAVES * 16 * SS * 8;
}

// MfmaArgs::wg_sum: the eight waves of a workgroup leave a window's sums in LDS and the workgroup adds them up when its run is
// done -- one partial sum per (window, group of eight half chunks, slot) instead of one per half chunk: an eighth of the
// bytes the kernel writes and k_ld_finalize_g reads (354 MB per group of 15 at chr1).  1 KiB of LDS per window of the run:
// taken where two workgroups still fit a CU.
int ld_mfma_wg_sum(uint32_t win_per_group, uint32_t tab_len, uint32_t max_seg)
{
#if IBDG_MFMA_WAVES == 8
    return ld_mfma_lds_bytes(win_per_group, tab_len, max_seg) + (size_t)win_per_group * 1024 <= 80 * 1024;
#else
    return 0;
#endif
}

#ifndef IBDG_MFMA_WAVES_PER_EU
#define IBDG_MFMA_WAVES_PER_EU 4
#endif
__global__ __launch_bounds__(64 * IBDG_MFMA_WAVES) __attribute__((amdgpu_waves_per_eu(IBDG_MFMA_WAVES_PER_EU, IBDG_MFMA_WAVES_PER_EU)))
void k_ld_mfma(MfmaArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    const uint32_t n_half = 2 * a.n_chunks, n_hgroups = (n_half + IBDG_MFMA_WAVES - 1) / IBDG_MFMA_WAVES;
#if IBDG_MFMA_XCD
    // The workgroups of ONE run (its n_hgroups groups of half chunks, times the launch's groups of comparison individuals)
    // are dealt to ONE XCD, next to each other: those of one group read the same target image, 1 KiB per segment, and the
    // same half chunks' workgroups of the launch's OTHER groups read the same tile words -- the XCD's L2 serves all but the
    // first of them.  Consecutive workgroup ids go round the 8 XCDs: workgroup b runs on XCD b % 8 as its (b / 8)-th; XCD x
    // takes the runs x, x + 8, x + 16, ... in that order, within a run the half-chunk groups in turn, within those the groups
    // of individuals (the launch holds 8 * n_hgroups * n_groups * ceil(n_runs / 8) workgroups; those past the last run leave).
    const uint32_t xcd = blockIdx.x & 7, nth = blockIdx.x >> 3, per_run = n_hgroups * a.n_groups;
    const uint32_t run = xcd + 8 * (nth / per_run), in_run = nth % per_run;
#if IBDG_MFMA_GRP_MAJOR
    const uint32_t hgroup = in_run % n_hgroups;
    const unsigned grp = in_run / n_hgroups;
#else
    const uint32_t hgroup = in_run / a.n_groups;
    const unsigned grp = in_run % a.n_groups;
#endif
    if (run >= a.n_runs)
        return;
#else
    const unsigned grp = blockIdx.z;
    const uint32_t run = blockIdx.x / n_hgroups, hgroup = blockIdx.x - run * n_hgroups;
#endif
    const uint32_t w0 = a.run_begin[run], w1 = a.run_begin[run + 1];
    const uint32_t seg0 = a.wconst[w0].seg_begin, seg1 = a.wconst[w1].seg_begin;
    if (seg1 == seg0)
        return;

    // The tau table comes first: the kernel has no static LDS, so the table sits at LDS address 0 and the accumulators --
    // 8 G -- are its entries' addresses as they are (hipcc itself folds the table's address to the constant 0; every
    // parity test of this kernel would fail if that ever changed).
    double *taud = reinterpret_cast<double *>(smem);               // tau^n as plain doubles (0 where it has left their range): LDS address 8 n
    uint4 *tau = reinterpret_cast<uint4 *>(taud + ((a.tab_len + 1) & ~1u));      // tau^n as {mantissa, exponent}
    uint4 *wcc = tau + a.tab_len;                                   // [win_per_group] eK, 16 AT + rho table, sigma table, segment end | all powers plain
    uint4 *wcs = wcc + a.win_per_group;                             // [win_per_group][32]: exponents of U_t0 (4), of U_t1 (4), 16 x two mantissas
    uint4 *tab = wcs + (size_t)a.win_per_group * 32;                // rho^n, sigma^n
    uint4 *rec = tab + 2 * (size_t)a.tab_len;              // [max_seg][2] tile, cov planes | the first six cov masks
    double *strip = reinterpret_cast<double *>(rec + 2 * ((size_t)a.max_seg + 1)) + (size_t)wave * 16 * SS;   // this wave's
    const uint32_t tab1 = (uint32_t)(uintptr_t)(lds_void *)tab, tab2 = tab1 + a.tab_len * 16;
    const uint32_t tau16 = (uint32_t)(uintptr_t)(lds_void *)tau;
    for (uint32_t i = threadIdx.x; i < w1 - w0; i += blockDim.x) {
        const WinConst &W = a.wconst[w0 + i];
        // (no G(x,t) of the window exceeds gmax, the largest <t,cov> of the group's haplotypes in it -- the spare exponent entry of
        // slot 15, k_win_slot_g: when tau^gmax >= 2^-480 all the window's look-ups are ordinary doubles with room to spare for
        // the common scale of the wave's lanes, see comp_quad_fast)
        const uint32_t gmax = reinterpret_cast<const uint32_t *>(a.wc_slot + ((size_t)grp * a.n_win + w0 + i) * 32)[PSEUDO];
        const uint32_t plain = a.plain_tau && gmax < a.tab_len && a.pow_tau[gmax].e >= -480 ? 1u << 31 : 0u;
        wcc[i] = make_uint4((uint32_t)W.eK, 16 * W.alt_total + tab1, tab2, a.wconst[w0 + i + 1].seg_begin | plain);
    }
    {
        const uint4 *src = a.wc_slot + ((size_t)grp * a.n_win + w0) * 32;
        for (uint32_t i = threadIdx.x; i < (w1 - w0) * 32; i += blockDim.x)
            wcs[i] = src[i];
    }
    for (uint32_t i = threadIdx.x; i < a.tab_len; i += blockDim.x) {
        tau[i] = reinterpret_cast<const uint4 *>(a.pow_tau)[i];
        taud[i] = a.pow_tau[i].e >= -1000 ? __builtin_ldexp(a.pow_tau[i].m, a.pow_tau[i].e) : 0.0;
        tab[i] = reinterpret_cast<const uint4 *>(a.pow_1me)[i];
        tab[a.tab_len + i] = reinterpret_cast<const uint4 *>(a.pow_eps)[i];
    }
    // the run's segments: nothing in the loops below goes through the scalar path (a scalar load per segment
    // and its wait cost more than the segment's arithmetic)
    // word 0 of a record: tile (< 2^24) of the segment TWO AHEAD (what the loop requests while it works on this one; the
    // run's last tile at the end), bit 24 cov planes beyond three, bit 25 first, bit 26 last segment of its window
    for (uint32_t i = threadIdx.x; i < seg1 - seg0; i += blockDim.x) {
        const Seg &S = a.segs[seg0 + i];
        const uint32_t ahead = seg0 + i + IBDG_MFMA_AHEAD < seg1 ? seg0 + i + IBDG_MFMA_AHEAD : seg1 - 1;
        const uint32_t first = (i == 0 || a.segs[seg0 + i - 1].last) ? 1u << 25 : 0u;
        const uint32_t deep = ((S.flags >> 16) & 0xff) > 3 ? 1u << 24 : 0u;
        const uint32_t wide = (((S.flags >> 16) & 0xff) > 4 || (S.flags >> 24) > 4) ? 1u << 27 : 0u;       // a weight of 16 or more: k_win_target_g
        rec[2 * i] = make_uint4(a.segs[ahead].tile | deep | wide | first | (S.last ? 1u << 26 : 0u), S.cov[0], S.cov[1], S.cov[2]);
        rec[2 * i + 1] = make_uint4(S.cov[3], S.cov[4], S.cov[5], S.cov[6]);
    }
    __syncthreads();

    const uint32_t hc = hgroup * IBDG_MFMA_WAVES + wave;                          // half chunk of this wave
    const bool wg_sum = a.wg_sum != 0;
    // (wg_sum) [window of the run][wave][16 slots]: the waves' sums, behind the strips
    const uint32_t wsum_base = (uint32_t)(uintptr_t)(lds_void *)(reinterpret_cast<double *>(rec + 2 * ((size_t)a.max_seg + 1)) +
                                                                 (size_t)IBDG_MFMA_WAVES * 16 * SS);
    if (hc >= n_half && !wg_sum)
        return;
    auto run_wave = [&]() {
    const uint32_t c = hc >> 1, n = lane & 31, h = lane >> 5;
    const uint32_t indiv = 64 * c + 32 * (hc & 1) + n;               // the lane's background individual
    const double wgt = a.base_weight[indiv];
    const uint32_t left = a.n_targets - grp * TG, cnt = left < TG ? left : TG;
    // bit r: slot r is the lane's own individual (no individual is in its own background, ibdgem.c:714)
    uint32_t excl = 0;
    for (uint32_t q = 0; q < cnt; ++q)
        if (a.targets[a.t_base + grp * TG + q] == indiv)
            excl |= 1u << q;
    // The operand loads go through buffer instructions: wave-uniform base (a resource descriptor in scalar registers: this
    // wave's chunk of the tiles, this group's target image), a scalar byte offset per segment, a constant 32-bit offset
    // per lane -- no vector instruction forms an address (a per-lane 64-bit pointer cost three 64-bit vector additions
    // per segment).  The offsets stay below 2^31: n_pairs < 2^21 tile pairs per chunk, n_segs < 2^21 segments per group
    // (checked by the host before it picks this kernel).
    const __amdgpu_buffer_rsrc_t xt_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4 *>(reinterpret_cast<const uint4 *>(a.t32) + (size_t)c * a.n_pairs * 64), 0, 0x7fffffff, 0x00020000);
    const uint32_t x_lane = (32 * (hc & 1) + n) * 16;
    const __amdgpu_buffer_rsrc_t ai_rsrc = __builtin_amdgcn_make_buffer_rsrc(a.aimg + (size_t)grp * a.n_segs * 64, 0, 0x7fffffff, 0x00020000);
    const uint32_t a_lane = lane * 16;
    // B operand: 8 where the individual carries the alt allele on tile row 4 h + d + 8 j (byte j of dword d of lane
    // half h): y = x rotated so that that bit sits at 3 + d + 8 j (left by 3 in the lower half of the wave, right by 1
    // in the upper; what wraps around lands where no mask looks), so dword d = (y >> d) & 0x08080808
    const uint32_t rot = h ? 1u : 29u;
    const uint32_t sh4 = 4 * h;          // (segments without a weight of 16 or more: the bits stay where they are in their bytes)

    // the reduction strip: lane (h, n) puts its addend I of a turn at row 8 h + I, column n; lane L = 4 s + p reads row s
    const uint32_t put_addr = (uint32_t)(uintptr_t)(lds_void *)strip + (8 * h * SS + n) * 8;
    const uint32_t get_addr = (uint32_t)(uintptr_t)(lds_void *)strip + ((lane >> 2) * SS + (lane & 3)) * 8;
    // row s = lane >> 2 is haplotype h = s >> 3 of slot (s & 7) + 8 turn; lane p = 0 of the quads of half 0 stores
    const uint32_t st_q = (lane >> 2) & 7;
    const bool st_lane = (lane & 35) == 0;                          // p = 0, h = 0
    // (the lanes that store a turn's sums, as EXEC masks)
    const uint64_t st_ok0 = __builtin_amdgcn_ballot_w64(st_lane && st_q < cnt), st_ok1 = __builtin_amdgcn_ballot_w64(st_lane && st_q + 8 < cnt);      // (slot 15 never: cnt <= 15)
    // Partial sums of this wave (half chunk hc) per window, for k_ld_finalize_g:
    //   t1[group][window][hc][16 slots]  the IBD1 sums -- the eight storing lanes of a turn write 64 consecutive bytes.
    // (The IBD0 sums are not this kernel's: MfmaArgs::p2w / p2c.)
    double *const t1_row = a.part_t1 + (((size_t)grp * a.n_win) * n_half + hc) * 16 + st_q;          // + window * n_half * 16 (+ 8 per turn)
    const uint32_t wsum_lane = wsum_base + wave * 128 + st_q * 8;                                     // + window of the run * 1024 (+ 64 per turn)
    const bool any_excl = __builtin_amdgcn_ballot_w64(excl != 0) != 0;
    const uint32_t n_quads = (cnt + 3) / 4;                          // groups of four slots that hold comparison individuals
    const uint32_t wcs_base = (uint32_t)(uintptr_t)(lds_void *)wcs;
    const uint32_t eu_lane = wcs_base + 64 * h;                      // + 512 per window: the exponents of U_t(h) of slots 0..15
    const uint32_t mu_lane = wcs_base + 128 + 16 * st_q + 8 * h;     // + 512 per window (+ 128 per turn): the mantissa of U_t(h) of the lane's row
    const uint32_t er_lane = wcs_base + 64 * h + 4 * st_q;           // + 512 per window (+ 32 per turn): its exponent

    // The operands of a segment (the lane's two tile words, 16 bytes of the target image) are requested two
    // segments ahead, into two register slots that the segments of the RUN take in turn (segment s: slot s & 1,
    // whatever window it belongs to; a queue that shifts costs six moves per segment): one segment of this kernel
    // is a few hundred cycles of work, a load from L2 or HBM takes a multiple of that.  The loop is straight-line
    // per segment -- every segment requests the one two ahead (at the end of the run: the last one again), its
    // record says which tile that is -- so that hipcc can count the loads in flight (s_waitcnt vmcnt(2)); with the
    // request under a condition, and the slots swapped after windows of an odd number of segments, it drained
    // them all (vmcnt(0)) at every window start and every swap, and a segment paid three LDS round trips.
    uint2 xq0, xq1, xq2;
    uint4 aq0, aq1, aq2;
    auto fetch = [&](uint32_t seg, uint32_t tile, uint2 &xq, uint4 &aq) {
#ifdef IBDG_EXP_SAMETILE            /* timing experiment: every request hits the same lines (what the memory latency costs); 1: both, 2: the tile words only, 3: the target image only */
        if (IBDG_EXP_SAMETILE != 2)
            seg = seg0;
        if (IBDG_EXP_SAMETILE != 3)
            tile = a.segs[seg0].tile;
#endif
        const auto xv = __builtin_amdgcn_raw_buffer_load_b64(xt_rsrc, x_lane, (tile >> 1) * 1024 + (tile & 1) * 8, IBDG_MFMA_TILE_AUX);
        const auto av = __builtin_amdgcn_raw_buffer_load_b128(ai_rsrc, a_lane, seg * 1024, 0);
        xq = make_uint2(xv[0], xv[1]);
        aq = make_uint4(av[0], av[1], av[2], av[3]);
    };
    {
        const uint32_t t0 = a.segs[seg0].tile, s1 = seg0 + 1 < seg1 ? seg0 + 1 : seg0, t1 = a.segs[s1].tile;
        fetch(seg0, t0, xq0, aq0);
        fetch(s1, t1, xq1, aq1);
#if IBDG_MFMA_AHEAD == 3
        const uint32_t s2 = seg0 + 2 < seg1 ? seg0 + 2 : seg1 - 1;
        fetch(s2, a.segs[s2].tile, xq2, aq2);
#endif
    }
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v16i acc0, acc1;
    uint32_t w = w0, s = seg0;
    const uint32_t seg_last = seg1 - 1;
    const uint4 *rec_p = rec;            // the record of the NEXT segment is read while this one is worked on
    uint4 r_cur = rec_p[0];
    // one segment from operand slot (xq, aq), which is refilled for segment s + 2; the first segment of a window
    // starts the sums (zero as the matrix instruction's addend: no 32 registers to clear per window)
    auto segment = [&](uint2 &xq, uint4 &aq, auto first) {
        const uint4 r0 = r_cur;
        const uint32_t ctl = __builtin_amdgcn_readfirstlane(r0.x);
        r_cur = rec_p[2];                    // (behind the run's last record: a spare one, never used)
        const uint2 x = xq;
        const v4i A = {(int)aq.x, (int)aq.y, (int)aq.z, (int)aq.w};
        v4i B0, B1;
        if (__builtin_expect((ctl & (1u << 27)) != 0, 0)) {
            // a row with 16 reads or more in the tile: plain weights in the image, every bit to the value 8
            const uint32_t b0 = __builtin_amdgcn_alignbit(x.x, x.x, rot), b1 = __builtin_amdgcn_alignbit(x.y, x.y, rot);
            B0 = v4i{(int)(b0 & 0x08080808u), (int)((b0 >> 1) & 0x08080808u), (int)((b0 >> 2) & 0x08080808u),
                     (int)((b0 >> 3) & 0x08080808u)};
            B1 = v4i{(int)(b1 & 0x08080808u), (int)((b1 >> 1) & 0x08080808u), (int)((b1 >> 2) & 0x08080808u),
                     (int)((b1 >> 3) & 0x08080808u)};
        } else {
            // the image carries the weights times 8 >> d: the bit of row 4 h + d + 8 j as it stands in its byte, 1 << d
            const uint32_t y0 = x.x >> sh4, y1 = x.y >> sh4;
            B0 = v4i{(int)(y0 & 0x01010101u), (int)(y0 & 0x02020202u), (int)(y0 & 0x04040404u), (int)(y0 & 0x08080808u)};
            B1 = v4i{(int)(y1 & 0x01010101u), (int)(y1 & 0x02020202u), (int)(y1 & 0x04040404u), (int)(y1 & 0x08080808u)};
        }
        // (<x0 & x1, cov> and the product of an individual's OWN genotype factors that needs it -- src/ibdgem.c:715, the IBD0
        // term -- do not depend on the comparison individuals: round 5 takes them from ONE pass of k_ld_popcount per site list
        // and background, MfmaArgs::p2w / p2c, instead of counting them again in every group's launch)
        rec_p += 2;
        ++s;
        if constexpr (decltype(first)::value) {
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B0, zero16, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B1, zero16, 0, 0, 0);
        } else {
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B1, acc1, 0, 0, 0);
        }
        // the slot's refill is requested only now, behind the matrix instructions that read it: issued before them, the
        // old and the new operand would both be live, hipcc would land the new one in other registers and move it over
        // at the end of the loop body -- behind a wait for every load in flight (s_waitcnt vmcnt(0))
#ifndef IBDG_EXP_NOFETCH
        fetch(s + (IBDG_MFMA_AHEAD - 1) < seg_last ? s + (IBDG_MFMA_AHEAD - 1) : seg_last, ctl & 0xffffffu, xq, aq);   // (s has been advanced: the segment IBDG_MFMA_AHEAD ahead)
#endif
    };
    auto window_end_all = [&](const uint4 kc) {
        // ---- the window's end: every lane finishes its individual against one haplotype of the 15 comparison individuals
        // register 15 = the weights' own rows: 16 C(x) in the lower half of the wave, 16 A(x) in the upper
        const auto w0s = __builtin_amdgcn_permlane32_swap((uint32_t)acc0[15], (uint32_t)acc0[15], false, false);
        const auto w1s = __builtin_amdgcn_permlane32_swap((uint32_t)acc1[15], (uint32_t)acc1[15], false, false);
        const uint32_t C0s = 2 * w0s[0], A0s = 2 * w0s[1], C1s = 2 * w1s[0], A1s = 2 * w1s[1];   // 16 C(x0), 16 A(x0), 16 C(x1), 16 A(x1)
        const bool plain = (__builtin_amdgcn_readfirstlane(kc.w) >> 31) != 0;
        const int eK = (int)kc.x;
        double mV0, mV1;
        int eV0, eV1;
        {
            // V_x = K' rho^(AT - A(x)) sigma^C(x) with the lane's background multiplicity folded into its mantissa
            uint4 r0, s0, r1, s1;
            const uint32_t ad3 = kc.y - A0s, ad4 = kc.z + C0s;
            const uint32_t ad5 = kc.y - A1s, ad6 = kc.z + C1s;
            asm volatile("ds_read_b128 %0, %4\n\t"
                         "ds_read_b128 %1, %5\n\t"
                         "ds_read_b128 %2, %6\n\t"
                         "ds_read_b128 %3, %7\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(s0), "=&v"(r1), "=&v"(s1)
                         : "v"(ad3), "v"(ad4), "v"(ad5), "v"(ad6)
                         : "memory");
            mV0 = wgt * (__hiloint2double((int)r0.y, (int)r0.x) * __hiloint2double((int)s0.y, (int)s0.x));
            mV1 = wgt * (__hiloint2double((int)r1.y, (int)r1.x) * __hiloint2double((int)s1.y, (int)s1.x));
            eV0 = eK + (int)r0.z + (int)s0.z;
            eV1 = eK + (int)r1.z + (int)s1.z;
        }
        const uint32_t eu_addr = eu_lane + (w - w0) * 512, mu_addr = mu_lane + (w - w0) * 512, er_addr = er_lane + (w - w0) * 512;
        // (plain path) the lanes' exponents brought to the wave's largest one, eRef; a lane without multiplicity (excluded, or
        // beyond the panel's individuals) has mV = 0 and does not set the scale
        int eRef = 0;
        double mVa = 0.0, mVb = 0.0;
        if (plain) {
            const int eR = eV0 > eV1 ? eV0 : eV1;
            eRef = wave_max_i32(wgt != 0.0 ? eR : -(1 << 28));
            mVa = __builtin_ldexp(mV0, eV0 - eRef);
            mVb = __builtin_ldexp(mV1, eV1 - eRef);
        }
        // The IBD1 addends of the lane's slots (:744-745) go to the strip, eight slots per turn (a short group occupies the
        // first registers only).  In the few waves that hold one of the group's comparison individuals, that lane's
        // addends for itself are left out: no individual is in its own background (ibdgem.c:714).
        auto window_end = [&](auto with_excl) {
            constexpr bool EX = decltype(with_excl)::value;
            // (Round 5: no wait for the operand requests here.  The window end's two stores are asm statements hipcc does not
            // count, store_f64_lanes, so the requests of the window's last segments stay in flight through the whole window
            // end -- round 4 drained them at this point, a memory latency per window: 9 % of the kernel.)
#if IBDG_MFMA_DRAIN
            __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
#define IBDG_QUAD(R0, EU)                                                                                    \
            {                                                                                                \
                double v[4];                                                                                 \
                if (plain)                                                                                   \
                    comp_quad_fast<R0>(acc0, acc1, mVa, mVb, v);                                             \
                else                                                                                         \
                    comp_quad<R0>(acc0, acc1, EU, tau16, mV0, mV1, eV0, eV1, v);                             \
                if (EX) {                                                                                    \
                    v[0] = (excl >> (R0)) & 1 ? 0.0 : v[0];                                                  \
                    v[1] = (excl >> (R0 + 1)) & 1 ? 0.0 : v[1];                                              \
                    v[2] = (excl >> (R0 + 2)) & 1 ? 0.0 : v[2];                                              \
                    v[3] = (excl >> (R0 + 3)) & 1 ? 0.0 : v[3];                                              \
                }                                                                                            \
                strip_put<(R0 & 7)>(put_addr, v[0]);                                                         \
                strip_put<(R0 & 7) + 1>(put_addr, v[1]);                                                     \
                strip_put<(R0 & 7) + 2>(put_addr, v[2]);                                                     \
                strip_put<(R0 & 7) + 3>(put_addr, v[3]);                                                     \
            }
            // a turn's sums: quad s of half h holds S_h of slot (s & 7) + 8 turn; the lanes that store add the two haplotypes:
            // mU_t0 S_0 + mU_t1 S_1 (the reference's ((Q00 + Q01) + Q10) + Q11 up to the association)
#define IBDG_TURN_END(TURN, OK)                                                                              \
            {                                                                                                \
                double mu;                                                                                   \
                int er;                                                                                      \
                const double S = strip_sum_mu(get_addr, mu_addr + 128 * TURN, er_addr + 32 * TURN, mu, er);  \
                /* (plain path: the sums are in units of 2^(eRef + eU of the row's slot and haplotype)) */   \
                const double part = __builtin_ldexp(mu * S, plain ? eRef + er : 0);                          \
                const uint32_t plo = from_upper_half((uint32_t)__double2loint(part));                        \
                const uint32_t phi = from_upper_half((uint32_t)__double2hiint(part));                        \
                const double both = part + __hiloint2double((int)phi, (int)plo);                             \
                if (wg_sum)                                                                                  \
                    lds_store_f64_lanes(OK, wsum_lane + (w - w0) * 1024 + 64 * TURN, both);                  \
                else                                                                                         \
                    store_f64_lanes(OK, t1_row + ((size_t)w * n_half * 16 + 8 * TURN), both);                \
            }
            if (n_quads > 2) {
                IBDG_QUAD(8, eu_addr)
                if (n_quads > 3)
                    IBDG_QUAD(12, eu_addr)
                IBDG_TURN_END(1, st_ok1)
            }
            {
                IBDG_QUAD(0, eu_addr)
                if (n_quads > 1)
                    IBDG_QUAD(4, eu_addr)
                IBDG_TURN_END(0, st_ok0)
            }
#undef IBDG_QUAD
#undef IBDG_TURN_END
        };
        if (any_excl)
            window_end(std::true_type());
        else
            window_end(std::false_type());
        ++w;
    };
#if IBDG_MFMA_AHEAD == 3
    // Round 5: THREE operand slots, taken in turn by the segments of the run (segment s: slot s mod 3), each refilled for the
    // segment three ahead -- a timing build whose requests all hit the same lines ran 9 % faster (profiles/r05_ab_mfma.txt):
    // that much was the latency of requests two segments ahead.  (The registers for the third slot came from the IBD0 terms
    // leaving the kernel: 126 -> 117 -> 123.)  A window that starts from slot k runs the same code with the slots' roles
    // rotated; which instance comes next is a uniform switch -- nothing is in flight across it, every window end waits for
    // all requests (they have landed long before).
    auto window = [&](uint2 &xa, uint4 &aa, uint2 &xb, uint4 &ab, uint2 &xc, uint4 &ac) -> uint32_t {
        const uint4 kc = wcc[w - w0];        // (read once per window: the segment count now, the rest at the window's end)
        const uint32_t n_more = (__builtin_amdgcn_readfirstlane(kc.w) & 0x7fffffffu) - s - 1;     // segments behind the first
        segment(xa, aa, std::true_type());
        uint32_t left = n_more;
        for (; left >= 3; left -= 3) {
            segment(xb, ab, std::false_type());
            segment(xc, ac, std::false_type());
            segment(xa, aa, std::false_type());
        }
        if (left >= 1)
            segment(xb, ab, std::false_type());
        if (left == 2)
            segment(xc, ac, std::false_type());
        window_end_all(kc);
        return left + 1 == 3 ? 0u : left + 1;         // slots the run has moved on by, modulo 3
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);          // (vmcnt(0): the first three segments' operands)
    uint32_t slot = 0;
    while (w < w1) {
        uint32_t adv;
        if (slot == 0)
            adv = window(xq0, aq0, xq1, aq1, xq2, aq2);
        else if (slot == 1)
            adv = window(xq1, aq1, xq2, aq2, xq0, aq0);
        else
            adv = window(xq2, aq2, xq0, aq0, xq1, aq1);
        slot += adv;
        slot = slot >= 3 ? slot - 3 : slot;
    }
#else
    // One window whose first segment finds its operands in slot (xa, aa); its segments take the two slots in turn.  Returns
    // whether it had an odd number of segments: the next window then starts from the other slot -- the same code with the
    // slots' roles exchanged (nothing moves: the loads in flight land where the next segments look for them).
    auto window = [&](uint2 &xa, uint4 &aa, uint2 &xb, uint4 &ab) -> bool {
        const uint4 kc = wcc[w - w0];        // (read once per window: the segment count now, the rest at the window's end)
        const uint32_t n_more = (__builtin_amdgcn_readfirstlane(kc.w) & 0x7fffffffu) - s - 1;     // segments behind the first
        segment(xa, aa, std::true_type());
        for (uint32_t i = n_more >> 1; i > 0; --i) {
            segment(xb, ab, std::false_type());
            segment(xa, aa, std::false_type());
        }
        if (n_more & 1)
            segment(xb, ab, std::false_type());
        window_end_all(kc);
        return !(n_more & 1);
    };
    __builtin_amdgcn_s_waitcnt(0x0F70);          // (vmcnt(0): the first two segments' operands, so that the loop's first wait is a counted one on every path into it)
    while (w < w1) {
        if (window(xq0, aq0, xq1, aq1)) {
            // from slot 1 until another odd window brings the run back to slot 0
            while (w < w1 && !window(xq1, aq1, xq0, aq0)) {
            }
        }
    }
    (void)xq2;
    (void)aq2;
#endif
    };
    if (hc < n_half)
        run_wave();
    if (wg_sum) {
        // the workgroup's sums: its waves' in the order of their half chunks
        __syncthreads();
        const double *ws = reinterpret_cast<const double *>(reinterpret_cast<double *>(rec + 2 * ((size_t)a.max_seg + 1)) +
                                                            (size_t)IBDG_MFMA_WAVES * 16 * SS);
        const uint32_t left_h = n_half - hgroup * IBDG_MFMA_WAVES, nw = left_h < IBDG_MFMA_WAVES ? left_h : IBDG_MFMA_WAVES;
        for (uint32_t i = threadIdx.x; i < (w1 - w0) * 16; i += blockDim.x) {
            const uint32_t wi = i >> 4, q = i & 15;
            double sum = 0.0;
            for (uint32_t v = 0; v < nw; ++v)
                sum += ws[((size_t)wi * IBDG_MFMA_WAVES + v) * 16 + q];
            a.part_t1[(((size_t)grp * a.n_win + w0 + wi) * n_hgroups + hgroup) * 16 + q] = sum;
        }
    }
}

// The value of lane ^ X: within the quad by a DPP move, beyond it through the LDS crossbar, across the halves by
// v_permlane32_swap (valid in the upper half only: what is read there).
template <int X>
__device__ __forceinline__ double lane_xor_get(double v)
{
    if (X == 32) {
        const uint32_t lo = __builtin_amdgcn_permlane32_swap((uint32_t)__double2loint(v), (uint32_t)__double2loint(v), false, false)[0];
        const uint32_t hi = __builtin_amdgcn_permlane32_swap((uint32_t)__double2hiint(v), (uint32_t)__double2hiint(v), false, false)[0];
        return __hiloint2double((int)hi, (int)lo);            // lanes 32..63: the value of lane l - 32
    }
    if (X == 1 || X == 2) {
        constexpr int ctrl = X == 1 ? 0xB1 : 0x4E;            // quad_perm [1,0,3,2] / [2,3,0,1]
        return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true),
                                __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true));
    }
    constexpr int pat = ((X & 31) << 10) | 0x1f;              // ds_swizzle, bit mode: and 0x1f, or 0, xor X
    return __hiloint2double(__builtin_amdgcn_ds_swizzle(__double2hiint(v), pat), __builtin_amdgcn_ds_swizzle(__double2loint(v), pat));
}

// One level of sixteen wave sums at once: the lanes whose bit X is clear go on with sum a, the others with sum b -- each adds
// its partner's (lane ^ X) addend of the sum it keeps.
template <int X>
__device__ __forceinline__ double tree_merge(double a, double b, uint32_t lane)
{
    const bool up = (lane & X) != 0;
    const double keep = up ? b : a, give = up ? a : b;
    return keep + lane_xor_get<X>(give);
}

// The sums over the wave's 64 lanes of SIXTEEN addends per lane, v[0..15], in the additions of wave_sum_to_lane63 -- the
// balanced tree over the lane number's bits 0, 1, ..., 5 -- for all sixteen at once: after the levels of bits 0..3 lane l
// holds the sum of its row of 16 lanes for addend l & 15 (8 + 4 + 2 + 1 merges instead of 16 x 4 steps), bits 4 and 5 add the
// rows.  Lanes 48 + q hold the total of v[q], the same bits as wave_sum_to_lane63(v[q]) leaves in lane 63.
__device__ __forceinline__ double wave_sums16(const double (&v)[16], uint32_t lane)
{
    double r[8], s[4], t[2];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        r[j] = tree_merge<1>(v[2 * j], v[2 * j + 1], lane);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        s[j] = tree_merge<2>(r[2 * j], r[2 * j + 1], lane);
#pragma unroll
    for (int j = 0; j < 2; ++j)
        t[j] = tree_merge<4>(s[2 * j], s[2 * j + 1], lane);
    double u = tree_merge<8>(t[0], t[1], lane);
    u = u + lane_xor_get<16>(u);
    u = u + lane_xor_get<32>(u);
    return u;
}

// The window averages of a group's comparison individuals (src/ibdgem.c:736-753): a wave per (window, group), four windows
// a workgroup.  IBD1: lane 16 j + q adds t1[hc][q] of half chunks hc = j, j + 4, ... in that order, the four part sums of a
// slot meet as ((j0 + j1) + j2) + j3.  IBD0 does not depend on the comparison individual except for its own exclusion
// (:714): for each slot in turn the wave reads, one lane an individual, the 64 products of the chunk the slot's individual
// sits in (p2w of the one pass over the site list; its own lane counts 0), adds them up, and then, one lane a chunk, that
// sum and the other chunks' sums (p2c) -- the additions of ibd0_from_pass (ibdg_ld_dev.h), which are those of a single
// comparison individual's own launch: IBD0 is the same bits whichever kernel served the individual.
__global__ __launch_bounds__(256) void k_ld_finalize_g(MfmaArgs a, const int *__restrict__ n_refpanel, double *__restrict__ win_ll)
{
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6), grp = blockIdx.y, lane = threadIdx.x & 63;
    if (w >= a.n_win)
        return;
    // (partial sums per half chunk, or per workgroup of eight of them: MfmaArgs::wg_sum)
    const uint32_t n_half = a.wg_sum ? (2 * a.n_chunks + IBDG_MFMA_WAVES - 1) / IBDG_MFMA_WAVES : 2 * a.n_chunks, q = lane & 15, j = lane >> 4;
    const uint32_t left = a.n_targets - grp * TG, cnt = left < TG ? left : TG;
    const double *t1 = a.part_t1 + (((size_t)grp * a.n_win + w) * n_half) * 16;
    double acc = 0.0;
    for (uint32_t hc = j; hc < n_half; hc += 4)
        acc += t1[(size_t)hc * 16 + q];
    const double a0 = __shfl(acc, q), a1 = __shfl(acc, q + 16), a2 = __shfl(acc, q + 32), a3 = __shfl(acc, q + 48);
    const double s1 = ((a0 + a1) + a2) + a3;                   // (every lane: the total of slot lane & 15)
    const double *pc = a.p2c + (size_t)w * a.n_chunks * 2;
    const double *pw = a.p2w + (size_t)w * a.lanes;
    const double pc_lane = lane < a.n_chunks ? pc[2 * lane] : 0.0;
    double s0 = 0.0;                                           // slot lane & 15's sum (in the lanes 48..63 at least)
    if (a.n_chunks <= 64) {
        // all slots at once (wave_sums16): first every slot's own chunk without its own individual, one lane an individual;
        // then, one lane a chunk, that sum in place of the chunk's -- the additions of ibd0_from_pass, slot by slot
        uint32_t c_own[16];
        double v[16];
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) {
            c_own[qq] = 0xffffffffu;
            v[qq] = 0.0;
            if ((uint32_t)qq < cnt) {
                const uint32_t tgt = a.targets[a.t_base + grp * TG + qq];      // (the same for the whole wave)
                c_own[qq] = tgt >> 6;
                const double x = pw[64 * (size_t)c_own[qq] + lane];
                v[qq] = lane == (tgt & 63) ? 0.0 : x;
            }
        }
        const double own_all = wave_sums16(v, lane);
#pragma unroll
        for (int qq = 0; qq < 16; ++qq) {
            const double own = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(own_all), 48 + qq),
                                                __builtin_amdgcn_readlane(__double2loint(own_all), 48 + qq));
            v[qq] = lane == c_own[qq] ? own : pc_lane;
        }
        s0 = wave_sums16(v, lane);
    } else
    for (uint32_t qq = 0; qq < cnt; ++qq) {
        const uint32_t tgt = a.targets[a.t_base + grp * TG + qq];      // (the same for the whole wave)
        const uint32_t c_own = tgt >> 6;                       // the chunk (64 individuals) the slot's own individual sits in
        double v = pw[64 * (size_t)c_own + lane];
        v = wave_sum_to_lane63(lane == (tgt & 63) ? 0.0 : v);
        const double own = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                                            __builtin_amdgcn_readlane(__double2loint(v), 63));
        double t0 = lane == c_own ? own : pc_lane;
        for (uint32_t c = lane + 64; c < a.n_chunks; c += 64)  // (panels of more than 4096 individuals)
            t0 += c == c_own ? own : pc[2 * c];
        const double tot = wave_sum_to_lane63(t0);
        const double all = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tot), 63),
                                            __builtin_amdgcn_readlane(__double2loint(tot), 63));
        s0 = q == qq ? all : s0;
    }
    if (lane >= 48 && q < cnt) {                               // (the lanes 48 + slot hold both totals)
        const uint32_t t = a.t_base + grp * TG + q;
        const int nref = n_refpanel[t];
        const double mK = a.wconst[w].mK;             // mantissa of K' (its exponent went into every term)
        double *o = win_ll + ((size_t)t * a.n_win + w) * 3;
        o[0] = (s0 * mK) / (double)nref;
        o[1] = (s1 * mK) / (double)(nref * 4);
    }
}

void launch_ld_finalize_g(const MfmaArgs &a, unsigned n_groups, const int *n_refpanel, double *win_ll, hipStream_t st)
{
    if (a.n_win == 0 || n_groups == 0)
        return;
    hipLaunchKernelGGL(k_ld_finalize_g, dim3((a.n_win + 3) / 4, n_groups), dim3(256), 0, st, a, n_refpanel, win_ll);
}

void launch_win_target_g(const MfmaArgs &a, unsigned n_groups, hipStream_t st)
{
    if (a.n_win == 0 || n_groups == 0)
        return;
    hipLaunchKernelGGL(k_win_target_g, dim3((unsigned)((((size_t)a.n_segs + IBDG_WTG_SEGS - 1) / IBDG_WTG_SEGS * 64 + 255) / 256), n_groups),
                       dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_win_slot_g, dim3((a.n_win + 3) / 4, n_groups), dim3(256), 0, st, a);
}

int launch_ld_mfma(const MfmaArgs &a, unsigned n_groups, hipStream_t st, KernelEvents ev)
{
    if (a.n_win == 0 || n_groups == 0)
        return 0;
    const size_t lds = ld_mfma_lds_bytes(a.win_per_group, a.tab_len, a.max_seg) + (a.wg_sum ? (size_t)a.win_per_group * 1024 : 0);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_ld_mfma), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return 1;
    const uint32_t n_hgroups = (2 * a.n_chunks + IBDG_MFMA_WAVES - 1) / IBDG_MFMA_WAVES;
    MfmaArgs b = a;
    b.n_groups = n_groups;
#if IBDG_MFMA_XCD
    const dim3 grid(8 * n_hgroups * n_groups * ((a.n_runs + 7) / 8), 1, 1);
#else
    const dim3 grid(a.n_runs * n_hgroups, 1, n_groups);
#endif
    hipExtLaunchKernelGGL(k_ld_mfma, grid, dim3(64 * IBDG_MFMA_WAVES), (uint32_t)lds, st, ev.start, ev.stop, 0, b);
    return 0;
}

}  // namespace ibdg
