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
//   per window: result register pair i of lane half h holds G(x, t0), G(x, t1) of comparison individual
//     slot(h, i) for the lane's background individual, so every lane finishes 8 (individual, comparison
//     individual) pairs without any exchange.  The exponents of a product are (header of ibdg_ld_popcount.hip)
//         E2 = AT - <t,alt> - A(x) + G(x,t),   E3 = <t,cov> + C(x) - 2 G(x,t),
//     so with tau = rho / sigma^2 = 4 eps (1-eps) the product factors into three parts,
//         K' rho^E2 sigma^E3 = [K' rho^(AT - A(x)) sigma^C(x)] . [rho^-<t,alt> sigma^<t,cov>] . tau^G(x,t)
//                            =            V_x                 .            U_t              . tau^G,
//     V_x once per (background haplotype, window) in the lane, U_t once per (comparison haplotype, window) from
//     k_win_slot_g, and ONE table look-up per product (the unfactored form needs two and eight address
//     computations per comparison individual): Q(x0,t) + Q(x1,t) = mU_t . (V_x0 tau^G0 + V_x1 tau^G1 scaled by
//     2^eU_t), everything as mantissa and integer exponent until one ldexp per product.  The eight addends of a lane
//     go to the wave's LDS strip as they are finished, and the sums over the 32 lanes of each half come from
//     reading the strip transposed (eight values per lane, then two exchange steps): 13 vector instructions
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

// Result register pair i (registers 2i, 2i+1) of lane half h holds rows (2i & 3) + 8 (2i >> 2) + 4 h (+1) of
// the product, i.e. target-haplotype rows 2 slot, 2 slot + 1 of:
__device__ __forceinline__ uint32_t slot_of(uint32_t h, uint32_t i) { return 4 * (i >> 1) + 2 * h + (i & 1); }

// ---------------------------------------------------------------------------
// Per group of comparison individuals: (1) the A operand of every segment,
//     aimg[group][segment][lane l] = 16 bytes: row m = l & 31 (slot m >> 1, target haplotype m & 1),
//     byte j of dword d = weight of tile row 4 (l >> 5) + d + 8 j (the order in which a shift and the mask
//     0x01010101 take the background bits out of a tile word):  cov_r where the target haplotype carries the
//     alt allele on a row of the segment, else 0; slot 15: cov_r (row 30) and alt_r (row 31) themselves;
// (2) per window and slot the four table offsets 16<t0,cov> 16<t1,cov> 16(AT-<t0,alt>) 16(AT-<t1,alt>)
//     (k_win_target's window constants).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_win_target_g(MfmaArgs a)
{
    const unsigned grp = blockIdx.y;
    const uint32_t left = a.n_targets - grp * TG, cnt = left < TG ? left : TG;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)a.n_segs * 64) {
        // a wave per segment: the segment's masks are wave-uniform (scalar loads)
        const uint32_t s = __builtin_amdgcn_readfirstlane((uint32_t)(i >> 6));
        const uint32_t l = (uint32_t)i & 63, m = l & 31, kb = l >> 5, q = m >> 1, th = m & 1;
        const Seg &S = a.segs[s];
        uint32_t sel = 0;                       // rows of the tile that count in this row of the operand
        const bool use_alt = q == PSEUDO && th;
        if (q == PSEUDO) {
            sel = 0xffffffffu;
        } else if (q < cnt) {
            const uint32_t tgt = a.targets[a.t_base + grp * TG + q];
            const uint4 *tt = reinterpret_cast<const uint4 *>(a.t32) + (size_t)(tgt >> 6) * a.n_pairs * 64 + (tgt & 63);
            const uint2 w = tile_words(tt, S.tile);
            sel = th ? w.y : w.x;
        }
        sel >>= 4 * kb;
        const uint32_t ncov = (S.flags >> 16) & 0xff, nalt = S.flags >> 24, np = ncov > nalt ? ncov : nalt;
        uint32_t out[4] = {0, 0, 0, 0};
        for (uint32_t k = 0; k < np; ++k) {
            const uint32_t f = ((use_alt ? S.alt[k] : S.cov[k]) >> (4 * kb)) & sel;
#pragma unroll
            for (int d = 0; d < 4; ++d)
                out[d] += ((f >> d) & 0x01010101u) << k;                // weights <= 127: no carry between bytes
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
    if (l < 16) {
        uint4 um = make_uint4(0, 0, 0, 0), ue = um;       // empty slots: U = 0
        if (real) {
            const PowEntry r0 = a.pow_1me[a0alt], r1 = a.pow_1me[a1alt], s0 = a.pow_eps[a0cov], s1 = a.pow_eps[a1cov];
            const double m0 = s0.m / r0.m, m1 = s1.m / r1.m;
            um = make_uint4((uint32_t)__double2loint(m0), (uint32_t)__double2hiint(m0), (uint32_t)__double2loint(m1),
                            (uint32_t)__double2hiint(m1));
            ue = make_uint4((uint32_t)(s0.e - r0.e), (uint32_t)(s1.e - r1.e), 0, 0);
        }
        uint4 *dst = a.wc_slot + (((size_t)grp * a.n_win + w) * 16 + q) * 2;
        dst[0] = um;
        dst[1] = ue;
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

// mU0 (q00 + q01) + mU1 (q10 + q11): the reference's ((Q00 + Q01) + Q10) + Q11 (ibdgem.c:716-719, :744-745) up to
// the association, from the slot's U constants and the four tau^G entries
__device__ __forceinline__ double comp_value(const uint4 &um, const uint2 &ue, const uint4 &p0, const uint4 &p1, const uint4 &p2,
                                             const uint4 &p3, double mV0, double mV1, int eV0, int eV1)
{
    const double mU0 = __hiloint2double((int)um.y, (int)um.x), mU1 = __hiloint2double((int)um.w, (int)um.z);
    const int eU0 = (int)ue.x, eU1 = (int)ue.y;
    const double q00 = __builtin_ldexp(mV0 * __hiloint2double((int)p0.y, (int)p0.x), eV0 + (int)p0.z + eU0);   // x0, t0
    const double q01 = __builtin_ldexp(mV1 * __hiloint2double((int)p1.y, (int)p1.x), eV1 + (int)p1.z + eU0);   // x1, t0
    const double q10 = __builtin_ldexp(mV0 * __hiloint2double((int)p2.y, (int)p2.x), eV0 + (int)p2.z + eU1);   // x0, t1
    const double q11 = __builtin_ldexp(mV1 * __hiloint2double((int)p3.y, (int)p3.x), eV1 + (int)p3.z + eU1);   // x1, t1
    return mU0 * (q00 + q01) + mU1 * (q10 + q11);
}

// Two comparison individuals of the lane (register pairs I and I + 1 of the two accumulators) per LDS round trip:
// twelve reads in one statement -- per individual the slot's two mantissas and two exponents (immediate offsets
// from the lane's slot base) and four tau^G entries.  At four waves per SIMD a round trip is not hidden by other
// waves; one per individual (eight per window) cost 11 % more than one per two.
template <int I>
__device__ __forceinline__ void comp_pair(const v16i &acc0, const v16i &acc1, uint32_t slot_base, uint32_t tab_tau, double mV0,
                                          double mV1, int eV0, int eV1, double &ra, double &rb)
{
    constexpr int OFA = 128 * (I >> 1) + 32 * (I & 1);        // slot_of(h, I) * 32 bytes, the 2 h part is in slot_base
    constexpr int OFB = 128 * ((I + 1) >> 1) + 32 * ((I + 1) & 1);
    const uint32_t a0 = lshl4_add_s((uint32_t)acc0[2 * I], tab_tau), a1 = lshl4_add_s((uint32_t)acc1[2 * I], tab_tau);
    const uint32_t a2 = lshl4_add_s((uint32_t)acc0[2 * I + 1], tab_tau), a3 = lshl4_add_s((uint32_t)acc1[2 * I + 1], tab_tau);
    const uint32_t b0 = lshl4_add_s((uint32_t)acc0[2 * I + 2], tab_tau), b1 = lshl4_add_s((uint32_t)acc1[2 * I + 2], tab_tau);
    const uint32_t b2 = lshl4_add_s((uint32_t)acc0[2 * I + 3], tab_tau), b3 = lshl4_add_s((uint32_t)acc1[2 * I + 3], tab_tau);
    uint4 uma, pa0, pa1, pa2, pa3, umb, pb0, pb1, pb2, pb3;
    uint2 uea, ueb;
    asm volatile("ds_read_b128 %0, %12 offset:%21\n\t"
                 "ds_read_b64 %1, %12 offset:%22\n\t"
                 "ds_read_b128 %2, %13\n\t"
                 "ds_read_b128 %3, %14\n\t"
                 "ds_read_b128 %4, %15\n\t"
                 "ds_read_b128 %5, %16\n\t"
                 "ds_read_b128 %6, %12 offset:%23\n\t"
                 "ds_read_b64 %7, %12 offset:%24\n\t"
                 "ds_read_b128 %8, %17\n\t"
                 "ds_read_b128 %9, %18\n\t"
                 "ds_read_b128 %10, %19\n\t"
                 "ds_read_b128 %11, %20\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(uma), "=&v"(uea), "=&v"(pa0), "=&v"(pa1), "=&v"(pa2), "=&v"(pa3), "=&v"(umb), "=&v"(ueb), "=&v"(pb0),
                   "=&v"(pb1), "=&v"(pb2), "=&v"(pb3)
                 : "v"(slot_base), "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "n"(OFA), "n"(OFA + 16),
                   "n"(OFB), "n"(OFB + 16)
                 : "memory");
    ra = comp_value(uma, uea, pa0, pa1, pa2, pa3, mV0, mV1, eV0, eV1);
    rb = comp_value(umb, ueb, pb0, pb1, pb2, pb3, mV0, mV1, eV0, eV1);
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

#ifndef IBDG_MFMA_WAVES
#define IBDG_MFMA_WAVES 8           /* waves = half chunks per workgroup: 8 share one copy of the tables (4: the strips'
                                       LDS leaves room for 3 waves per SIMD only, 12 % slower) */
#endif
// LDS per workgroup: window constants (16 B + 16 slots x 32 B per window), the three power tables,
// the run's segment records, a reduction strip per wave
size_t ld_mfma_lds_bytes(uint32_t win_per_group, uint32_t tab_len, uint32_t max_seg)
{
    return (size_t)win_per_group * 33 * 16 + (size_t)tab_len * 48 + (size_t)max_seg * 32 +
           (size_t)IBDG_MFMA_WAVES * 16 * SS * 8;
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
    const unsigned grp = blockIdx.z;
    const uint32_t n_half = 2 * a.n_chunks, n_hgroups = (n_half + IBDG_MFMA_WAVES - 1) / IBDG_MFMA_WAVES;
    const uint32_t run = blockIdx.x / n_hgroups, hgroup = blockIdx.x - run * n_hgroups;
    const uint32_t w0 = a.run_begin[run], w1 = a.run_begin[run + 1];
    const uint32_t seg0 = a.wconst[w0].seg_begin, seg1 = a.wconst[w1].seg_begin;
    if (seg1 == seg0)
        return;

    uint4 *wcc = reinterpret_cast<uint4 *>(smem);                  // [win_per_group] eK, 16 AT + rho table, sigma table, segment end
    uint4 *wcs = wcc + a.win_per_group;                             // [win_per_group][16 slots][2] U mantissas | U exponents
    uint4 *tab = wcs + (size_t)a.win_per_group * 32;                // rho^n, sigma^n, tau^n
    uint4 *rec = tab + 3 * (size_t)a.tab_len;              // [max_seg][2] tile, cov planes | the first six cov masks
    double *strip = reinterpret_cast<double *>(rec + 2 * (size_t)a.max_seg) + (size_t)wave * 16 * SS;   // this wave's
    const uint32_t tab1 = (uint32_t)(uintptr_t)(lds_void *)tab, tab2 = tab1 + a.tab_len * 16, tab3 = tab2 + a.tab_len * 16;
    for (uint32_t i = threadIdx.x; i < w1 - w0; i += blockDim.x) {
        const WinConst &W = a.wconst[w0 + i];
        wcc[i] = make_uint4((uint32_t)W.eK, 16 * W.alt_total + tab1, tab2, a.wconst[w0 + i + 1].seg_begin);
    }
    {
        const uint4 *src = a.wc_slot + ((size_t)grp * a.n_win + w0) * 32;
        for (uint32_t i = threadIdx.x; i < (w1 - w0) * 32; i += blockDim.x)
            wcs[i] = src[i];
    }
    for (uint32_t i = threadIdx.x; i < 3 * a.tab_len; i += blockDim.x)
        tab[i] = i < a.tab_len ? reinterpret_cast<const uint4 *>(a.pow_1me)[i]
                               : (i < 2 * a.tab_len ? reinterpret_cast<const uint4 *>(a.pow_eps)[i - a.tab_len]
                                                    : reinterpret_cast<const uint4 *>(a.pow_tau)[i - 2 * a.tab_len]);
    // the run's segments: nothing in the loops below goes through the scalar path (a scalar load per segment
    // and its wait cost more than the segment's arithmetic)
    for (uint32_t i = threadIdx.x; i < seg1 - seg0; i += blockDim.x) {
        const Seg &S = a.segs[seg0 + i];
        rec[2 * i] = make_uint4(S.tile | (S.flags & 0xff0000u) << 8, S.cov[0], S.cov[1], S.cov[2]);   // tile < 2^24
        rec[2 * i + 1] = make_uint4(S.cov[3], S.cov[4], S.cov[5], S.cov[6]);
    }
    __syncthreads();

    const uint32_t hc = hgroup * IBDG_MFMA_WAVES + wave;                          // half chunk of this wave
    if (hc >= n_half)
        return;
    const uint32_t c = hc >> 1, n = lane & 31, h = lane >> 5;
    const uint32_t indiv = 64 * c + 32 * (hc & 1) + n;               // the lane's background individual
    const double wgt = a.base_weight[indiv];
    const uint32_t left = a.n_targets - grp * TG, cnt = left < TG ? left : TG;
    // bit i: slot_of(h, i) is the lane's own individual (no individual is in its own background, ibdgem.c:714)
    uint32_t excl = 0;
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
        const uint32_t q = slot_of(h, i);
        if (q < cnt && q != PSEUDO && a.targets[a.t_base + grp * TG + q] == indiv)
            excl |= 1u << i;
    }
    const uint4 *xt = reinterpret_cast<const uint4 *>(a.t32) + (size_t)c * a.n_pairs * 64 + 32 * (hc & 1) + n;   // + pair * 64
    const uint4 *ai = a.aimg + (size_t)grp * a.n_segs * 64 + lane;                                              // + segment * 64
    const uint32_t sh = 4 * h;

    // the reduction strip: lane (h, n) puts its addend I at row 8 h + I, column n; lane L = 4 s + p reads row s
    const uint32_t put_addr = (uint32_t)(uintptr_t)(lds_void *)strip + (8 * h * SS + n) * 8;
    const uint32_t get_addr = (uint32_t)(uintptr_t)(lds_void *)strip + ((lane >> 2) * SS + (lane & 3)) * 8;
    // row s = lane >> 2 belongs to half s >> 3 = h and register pair s & 7: lane p = 0 of the quad stores it
    const uint32_t st_q = slot_of(h, (lane >> 2) & 7);
    const bool st_ok = (lane & 3) == 0 && st_q < cnt && st_q != PSEUDO;
    const size_t st_row = (((size_t)(grp * TG + st_q) * a.n_win) * n_half + hc) * 2;                       // window 0
    const bool any_excl = __builtin_amdgcn_ballot_w64(excl != 0) != 0;
    const uint32_t n_iter = 2 * ((cnt + 3) / 4);                   // register pairs that hold comparison individuals
    const uint32_t wcs_lane = (uint32_t)(uintptr_t)(lds_void *)wcs + 64 * h;     // + 512 per window: slots 2 h, 2 h + 1, ...

    uint32_t s = seg0;
    // The operands of a segment (the lane's two tile words, 16 bytes of the target image) are requested two
    // segments ahead, into two register slots used in turn (a queue that shifts costs six moves per segment):
    // one segment of this kernel is a few hundred cycles of work, a load from L2 or HBM takes a multiple of that
    // (one ahead: 0.40 ms per individual against 0.36; four: 0.40; six: 0.49 -- registers).
    uint2 xq0 = make_uint2(0, 0), xq1 = xq0;
    uint4 aq0 = make_uint4(0, 0, 0, 0), aq1 = aq0;
    auto fetch = [&](uint32_t seg, uint2 &xq, uint4 &aq) {
        const uint32_t tile = __builtin_amdgcn_readfirstlane(rec[2 * (seg - seg0)].x) & 0xffffffu;
        xq = reinterpret_cast<const uint2 *>(xt + (size_t)(tile >> 1) * 64)[tile & 1];
        aq = ai[(size_t)seg * 64];
    };
    fetch(seg0, xq0, aq0);
    if (seg0 + 1 < seg1)
        fetch(seg0 + 1, xq1, aq1);
    const v16i zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t w = w0; w < w1; ++w) {
        const uint4 kc = wcc[w - w0];
        const uint32_t se = __builtin_amdgcn_readfirstlane(kc.w);
        v16i acc0, acc1;
        uint32_t CH;
        // one segment from operand slot (xq, aq), which is refilled for segment s + 2; the first segment of a window
        // starts the sums (zero as the matrix instruction's addend: no 32 registers to clear per window)
        auto segment = [&](uint2 &xq, uint4 &aq, auto first) {
            const uint2 x = xq;
            const v4i A = {(int)aq.x, (int)aq.y, (int)aq.z, (int)aq.w};
            const uint4 r0 = rec[2 * (s - seg0)];
            if (s + 2 < seg1)
                fetch(s + 2, xq, aq);
            // B operand: byte j of dword d of lane half kb = the individual's bit of row 4 kb + d + 8 j -- one shift
            // and one mask per dword, no table
            const uint32_t b0 = x.x >> sh, b1 = x.y >> sh;
            const v4i B0 = {(int)(b0 & 0x01010101u), (int)((b0 >> 1) & 0x01010101u), (int)((b0 >> 2) & 0x01010101u),
                            (int)((b0 >> 3) & 0x01010101u)};
            const v4i B1 = {(int)(b1 & 0x01010101u), (int)((b1 >> 1) & 0x01010101u), (int)((b1 >> 2) & 0x01010101u),
                            (int)((b1 >> 3) & 0x01010101u)};
            const uint32_t hom = x.x & x.y;
            uint32_t ch = (uint32_t)__popc(hom & r0.y) + ((uint32_t)__popc(hom & r0.z) << 1) + ((uint32_t)__popc(hom & r0.w) << 2);
            const uint32_t ncov = __builtin_amdgcn_readfirstlane(r0.x) >> 24;
            if (ncov > 3) {                      // deep rows (cov >= 8): max_cov < 128, seven planes at most
                const uint4 r1 = rec[2 * (s - seg0) + 1];
                ch += ((uint32_t)__popc(hom & r1.x) << 3) + ((uint32_t)__popc(hom & r1.y) << 4) +
                      ((uint32_t)__popc(hom & r1.z) << 5) + ((uint32_t)__popc(hom & r1.w) << 6);
            }
            if constexpr (decltype(first)::value) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B0, zero16, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B1, zero16, 0, 0, 0);
                CH = ch;
            } else {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B1, acc1, 0, 0, 0);
                CH += ch;
            }
        };
        // Segments take the two operand slots in turn, slot 0 first in every window: two per turn of the loop, and
        // after an odd number of segments the slots change places (six register moves per such window instead of
        // six per segment for a queue that shifts).
        if (s < se) {
            segment(xq0, aq0, std::true_type());
            ++s;
            if (s < se) {
                segment(xq1, aq1, std::false_type());
                ++s;
                while (s + 1 < se) {
                    segment(xq0, aq0, std::false_type());
                    ++s;
                    segment(xq1, aq1, std::false_type());
                    ++s;
                }
                if (s < se) {
                    segment(xq0, aq0, std::false_type());
                    ++s;
                    slots_swap(xq0, aq0, xq1, aq1);
                }
            } else {
                slots_swap(xq0, aq0, xq1, aq1);
            }
        } else {                                  // (every window has a covered row, hence a segment)
            acc0 = acc1 = zero16;
            CH = 0;
        }

        // ---- the window's end: every lane finishes its individual against 8 comparison individuals
        const uint32_t C0 = from_upper_half((uint32_t)acc0[14]), a0 = from_upper_half((uint32_t)acc0[15]);
        const uint32_t C1 = from_upper_half((uint32_t)acc1[14]), a1 = from_upper_half((uint32_t)acc1[15]);
        const int eK = (int)kc.x;
        double wP2, mV0, mV1;
        int eV0, eV1;
        {
            // pDg[x0+x1] (ibdgem.c:715): E3 = C0 + C1 - 2 CH, E2 = AT - a0 - a1 + CH;
            // V_x = K' rho^(AT - A(x)) sigma^C(x) with the lane's background multiplicity folded into its mantissa
            uint4 p1, p2, r0, s0, r1, s1;
            const uint32_t ad1 = lshl_add<4>(CH - (a0 + a1), kc.y), ad2 = lshl_add<4>(mad24<-2>(CH, C0 + C1), kc.z);
            const uint32_t ad3 = lshl_add<4>(0u - a0, kc.y), ad4 = lshl_add<4>(C0, kc.z);
            const uint32_t ad5 = lshl_add<4>(0u - a1, kc.y), ad6 = lshl_add<4>(C1, kc.z);
            asm volatile("ds_read_b128 %0, %6\n\t"
                         "ds_read_b128 %1, %7\n\t"
                         "ds_read_b128 %2, %8\n\t"
                         "ds_read_b128 %3, %9\n\t"
                         "ds_read_b128 %4, %10\n\t"
                         "ds_read_b128 %5, %11\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(p1), "=&v"(p2), "=&v"(r0), "=&v"(s0), "=&v"(r1), "=&v"(s1)
                         : "v"(ad1), "v"(ad2), "v"(ad3), "v"(ad4), "v"(ad5), "v"(ad6)
                         : "memory");
            wP2 = wgt * ld_value(eK, p1, p2);                  // :743
            mV0 = wgt * (__hiloint2double((int)r0.y, (int)r0.x) * __hiloint2double((int)s0.y, (int)s0.x));
            mV1 = wgt * (__hiloint2double((int)r1.y, (int)r1.x) * __hiloint2double((int)s1.y, (int)s1.x));
            eV0 = eK + (int)r0.z + (int)s0.z;
            eV1 = eK + (int)r1.z + (int)s1.z;
        }
        const uint32_t slot_base = wcs_lane + (w - w0) * 512;
        // The IBD1 addends of the lane's eight comparison individuals (:744-745) go to the strip as they are finished
        // (a short group occupies the first register pairs only: slots 4j .. 4j+3 are pairs 2j, 2j+1 of the two halves;
        // the upper half's last pair are the weights' own rows, whose slot has U = 0).  In the few waves that hold one
        // of the group's comparison individuals, that lane's addends for itself are left out: no individual is in its
        // own background (ibdgem.c:714).
        auto window_end = [&](auto with_excl) {
            constexpr bool EX = decltype(with_excl)::value;
#define IBDG_COMP2(I)                                                                                        \
            if (I < n_iter) {                                                                                \
                double va, vb;                                                                               \
                comp_pair<I>(acc0, acc1, slot_base, tab3, mV0, mV1, eV0, eV1, va, vb);                       \
                if (EX) {                                                                                    \
                    va = (excl >> I) & 1 ? 0.0 : va;                                                         \
                    vb = (excl >> (I + 1)) & 1 ? 0.0 : vb;                                                   \
                }                                                                                            \
                strip_put<I>(put_addr, va);                                                                  \
                strip_put<I + 1>(put_addr, vb);                                                              \
            }
            IBDG_COMP2(0) IBDG_COMP2(2) IBDG_COMP2(4) IBDG_COMP2(6)
#undef IBDG_COMP2
            const double t1 = strip_sum(get_addr);
            double t0;
            if (!EX) {
                // the IBD0 addends are the same for all comparison individuals: one butterfly over the half
                double s0 = wP2;
                s0 = s0 + swz_get<1>(s0);
                s0 = s0 + swz_get<2>(s0);
                s0 = swz_add<4>(s0);
                s0 = swz_add<8>(s0);
                s0 = swz_add<16>(s0);
                t0 = s0;
            } else {
                // the IBD0 addend of a lane counts for all comparison individuals but itself: a second turn of the strip
#define IBDG_PUT0(I) strip_put<I>(put_addr, (excl >> I) & 1 ? 0.0 : wP2);
                IBDG_PUT0(0) IBDG_PUT0(1) IBDG_PUT0(2) IBDG_PUT0(3) IBDG_PUT0(4) IBDG_PUT0(5) IBDG_PUT0(6) IBDG_PUT0(7)
#undef IBDG_PUT0
                t0 = strip_sum(get_addr);
            }
            if (st_ok)
                *reinterpret_cast<double2 *>(a.partial + st_row + (size_t)w * n_half * 2) = make_double2(t0, t1);
        };
        if (any_excl)
            window_end(std::true_type());
        else
            window_end(std::false_type());
    }
}

void launch_win_target_g(const MfmaArgs &a, unsigned n_groups, hipStream_t st)
{
    if (a.n_win == 0 || n_groups == 0)
        return;
    hipLaunchKernelGGL(k_win_target_g, dim3((unsigned)(((size_t)a.n_segs * 64 + 255) / 256), n_groups), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_win_slot_g, dim3((a.n_win + 3) / 4, n_groups), dim3(256), 0, st, a);
}

int launch_ld_mfma(const MfmaArgs &a, unsigned n_groups, hipStream_t st, KernelEvents ev)
{
    if (a.n_win == 0 || n_groups == 0)
        return 0;
    const size_t lds = ld_mfma_lds_bytes(a.win_per_group, a.tab_len, a.max_seg);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_ld_mfma), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return 1;
    const uint32_t n_hgroups = (2 * a.n_chunks + IBDG_MFMA_WAVES - 1) / IBDG_MFMA_WAVES;
    hipExtLaunchKernelGGL(k_ld_mfma, dim3(a.n_runs * n_hgroups, 1, n_groups), dim3(64 * IBDG_MFMA_WAVES), (uint32_t)lds, st, ev.start,
                          ev.stop, 0, a);
    return 0;
}

}  // namespace ibdg
