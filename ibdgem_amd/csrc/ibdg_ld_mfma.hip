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
// same tile.  The integers are the same integers, so everything after them -- the table products, the
// background sums in the fixed tree of wave_sum2, k_ld_finalize -- gives the bits of the other kernels
// (tests: a batched run equals single runs bit for bit).
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
//     individual) pairs without any exchange: four table products each, the weights, and the sums over the 32
//     lanes of each half by a transposed reduction (reduce8_halves / reduce16_halves: every value through the
//     tree of wave_sum2).  The two halves of a chunk are added by k_ld_finalize (halves = 1) -- the last
//     addition of the 64-lane tree.  Nothing in the loops goes through the scalar memory path: the run's
//     segment records are staged into LDS once per workgroup (a scalar load per segment cost more than the
//     segment's arithmetic).
// Operand / result layout of the instruction: tools/ubench/mfma_i8_layout.hip (checked on the device).
#include "ibdg_kernels.h"
#include "ibdg_ld_dev.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

namespace ibdg {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr uint32_t TG = IBDG_TG;     // comparison individuals per group
#ifndef IBDG_MFMA_PREFETCH
#define IBDG_MFMA_PREFETCH 2
#endif
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

// (2) of the above: a wave per window, lane l = slot l & 15, segments l >> 4, l >> 4 + 4, ... of the window
__global__ __launch_bounds__(256) void k_win_slot_g(MfmaArgs a)
{
    const unsigned grp = blockIdx.y;
    const uint32_t left = a.n_targets - grp * TG, cnt = left < TG ? left : TG;
    const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));
    if (w >= a.n_win)
        return;
    const uint32_t l = threadIdx.x & 63, q = l & 15;
    const uint32_t AT = a.wconst[w].alt_total;
    uint32_t a0cov = 0, a1cov = 0, a0alt = 0, a1alt = 0;
    if (q < cnt && q != PSEUDO) {
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
    if (l < 16)
        a.wc_slot[((size_t)grp * a.n_win + w) * 16 + q] = make_uint4(16 * a0cov, 16 * a1cov, 16 * (AT - a0alt), 16 * (AT - a1alt));
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

// Sixteen sums over the 32 lanes of a half at once.  A butterfly per value costs five exchanges and five
// additions each (80 + 80), every exchange a round trip through the LDS port whose latency this kernel is
// bound by; here every level hands half of the values to the partner lane and keeps the other half --
// 8 + 4 + 2 + 1 exchanges, then one butterfly level for the one value left: 16 exchanges and additions in
// five round trips.  Every value still goes through the tree of the butterfly (level j adds the totals of
// two adjacent blocks of 2^j lanes, own + partner's), so the sums are the same bits.
// Lane n ends up with the total of value 8 (n & 1) + 4 (n >> 1 & 1) + 2 (n >> 2 & 1) + (n >> 3 & 1).
__device__ __forceinline__ double reduce16_halves(const double (&v)[16], uint32_t n)
{
    const bool b0 = n & 1, b1 = n & 2, b2 = n & 4, b3 = n & 8;
    double u[8], x[4], y[2];
#pragma unroll
    for (int k = 0; k < 8; ++k)
        u[k] = (b0 ? v[k + 8] : v[k]) + swz_get<1>(b0 ? v[k] : v[k + 8]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        x[k] = (b1 ? u[k + 4] : u[k]) + swz_get<2>(b1 ? u[k] : u[k + 4]);
#pragma unroll
    for (int k = 0; k < 2; ++k)
        y[k] = (b2 ? x[k + 2] : x[k]) + swz_get<4>(b2 ? x[k] : x[k + 2]);
    double z = (b3 ? y[1] : y[0]) + swz_get<8>(b3 ? y[0] : y[1]);
    z = swz_add<16>(z);
    return z;
}

// The same for eight values: lane n ends up with the total of value 4 (n & 1) + 2 (n >> 1 & 1) + (n >> 2 & 1).
__device__ __forceinline__ double reduce8_halves(const double (&v)[8], uint32_t n)
{
    const bool b0 = n & 1, b1 = n & 2, b2 = n & 4;
    double u[4], x[2];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        u[k] = (b0 ? v[k + 4] : v[k]) + swz_get<1>(b0 ? v[k] : v[k + 4]);
#pragma unroll
    for (int k = 0; k < 2; ++k)
        x[k] = (b1 ? u[k + 2] : u[k]) + swz_get<2>(b1 ? u[k] : u[k + 2]);
    double y = (b2 ? x[1] : x[0]) + swz_get<4>(b2 ? x[0] : x[1]);
    y = swz_add<8>(y);
    y = swz_add<16>(y);
    return y;
}

// LDS per workgroup: window constants (16 B + 16 slots x 16 B per window), the two power tables,
// the run's segment records
size_t ld_mfma_lds_bytes(uint32_t win_per_group, uint32_t tab_len, uint32_t max_seg)
{
    return (size_t)win_per_group * 17 * 16 + (size_t)tab_len * 32 + (size_t)max_seg * 32;
}

#ifndef IBDG_MFMA_WAVES
#define IBDG_MFMA_WAVES 4           /* waves = half chunks per workgroup (8: 3 % slower; 5 or 6 waves per SIMD spill) */
#endif
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

    uint4 *wcc = reinterpret_cast<uint4 *>(smem);                  // [win_per_group] eK, 16 AT + rho table, sigma table
    uint4 *wcs = wcc + a.win_per_group;                             // [win_per_group][16] per slot
    uint4 *tab = wcs + (size_t)a.win_per_group * 16;                // rho^n then sigma^n
    uint4 *rec = tab + 2 * (size_t)a.tab_len;              // [max_seg][2] tile, cov planes | the first six cov masks
    const uint32_t tab1 = (uint32_t)(uintptr_t)(lds_void *)tab, tab2 = tab1 + a.tab_len * 16;
    for (uint32_t i = threadIdx.x; i < w1 - w0; i += blockDim.x) {
        const WinConst &W = a.wconst[w0 + i];
        wcc[i] = make_uint4((uint32_t)W.eK, 16 * W.alt_total + tab1, tab2, a.wconst[w0 + i + 1].seg_begin);
    }
    {
        const uint4 *src = a.wc_slot + ((size_t)grp * a.n_win + w0) * 16;
        for (uint32_t i = threadIdx.x; i < (w1 - w0) * 16; i += blockDim.x) {
            uint4 v = src[i];
            v.x += tab2;
            v.y += tab2;
            v.z += tab1;
            v.w += tab1;
            wcs[i] = v;
        }
    }
    for (uint32_t i = threadIdx.x; i < 2 * a.tab_len; i += blockDim.x)
        tab[i] = i < a.tab_len ? reinterpret_cast<const uint4 *>(a.pow_1me)[i]
                               : reinterpret_cast<const uint4 *>(a.pow_eps)[i - a.tab_len];
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
    uint32_t m32 = (uint32_t)-32;
    asm volatile("" : "+v"(m32));

    // where the lane's total of reduce16_halves goes: value 8 (n & 1) + 4 (n >> 1 & 1) + 2 (n >> 2 & 1) + (n >> 3 & 1) is
    // s0 / s1 (odd) of register pair value >> 1; lanes 0..15 of each half store
    const uint32_t val = 8 * (n & 1) + 4 * ((n >> 1) & 1) + 2 * ((n >> 2) & 1) + ((n >> 3) & 1);
    // (eight-value path: value 4 (n & 1) + 2 (n >> 1 & 1) + (n >> 2 & 1) = register pair; lanes 0..7 store both sums)
    const uint32_t val8 = 4 * (n & 1) + 2 * ((n >> 1) & 1) + ((n >> 2) & 1), st8_q = slot_of(h, val8);
    const bool st8_ok = n < 8 && st8_q < cnt && st8_q != PSEUDO;
    const size_t st8_row = (((size_t)(grp * TG + st8_q) * a.n_win) * n_half + hc) * 2;
    const bool any_excl = __builtin_amdgcn_ballot_w64(excl != 0) != 0;
    const uint32_t n_iter = 2 * ((cnt + 3) / 4);                   // register pairs that hold comparison individuals
    const uint32_t st_q = slot_of(h, val >> 1);
    const bool st_ok = n < 16 && st_q < cnt && st_q != PSEUDO;
    const size_t st_row = (((size_t)(grp * TG + st_q) * a.n_win) * n_half + hc) * 2 + (val & 1);   // window 0

    uint32_t s = seg0;
    // The operands of a segment (the lane's two tile words, 16 bytes of the target image) are requested PF
    // segments ahead, into a queue of registers: one segment of this kernel is a few hundred cycles of work,
    // a load from L2 or HBM takes a multiple of that.
    constexpr uint32_t PF = IBDG_MFMA_PREFETCH;
    uint2 xq[PF];
    uint4 aq[PF];
#pragma unroll
    for (uint32_t d = 0; d < PF; ++d) {
        xq[d] = make_uint2(0, 0);
        aq[d] = make_uint4(0, 0, 0, 0);
        if (seg0 + d < seg1) {
            const uint32_t tile = __builtin_amdgcn_readfirstlane(rec[2 * d].x) & 0xffffffu;
            xq[d] = reinterpret_cast<const uint2 *>(xt + (size_t)(tile >> 1) * 64)[tile & 1];
            aq[d] = ai[(size_t)(seg0 + d) * 64];
        }
    }
    for (uint32_t w = w0; w < w1; ++w) {
        const uint4 kc = wcc[w - w0];
        const uint32_t se = __builtin_amdgcn_readfirstlane(kc.w);
        v16i acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, acc1 = acc0;
        uint32_t CH = 0;
        for (; s < se; ++s) {
            const uint2 x = xq[0];
            const v4i A = {(int)aq[0].x, (int)aq[0].y, (int)aq[0].z, (int)aq[0].w};
            const uint4 r0 = rec[2 * (s - seg0)];
#pragma unroll
            for (uint32_t d = 0; d + 1 < PF; ++d) {
                xq[d] = xq[d + 1];
                aq[d] = aq[d + 1];
            }
            if (s + PF < seg1) {
                const uint32_t tile = __builtin_amdgcn_readfirstlane(rec[2 * (s + PF - seg0)].x) & 0xffffffu;
                xq[PF - 1] = reinterpret_cast<const uint2 *>(xt + (size_t)(tile >> 1) * 64)[tile & 1];
                aq[PF - 1] = ai[(size_t)(s + PF) * 64];
            }
            // B operand: byte j of dword d of lane half kb = the individual's bit of row 4 kb + d + 8 j -- one shift
            // and one mask per dword, no table
            const uint32_t b0 = x.x >> sh, b1 = x.y >> sh;
            const v4i B0 = {(int)(b0 & 0x01010101u), (int)((b0 >> 1) & 0x01010101u), (int)((b0 >> 2) & 0x01010101u),
                            (int)((b0 >> 3) & 0x01010101u)};
            const v4i B1 = {(int)(b1 & 0x01010101u), (int)((b1 >> 1) & 0x01010101u), (int)((b1 >> 2) & 0x01010101u),
                            (int)((b1 >> 3) & 0x01010101u)};
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B1, acc1, 0, 0, 0);
            const uint32_t hom = x.x & x.y;
            CH += (uint32_t)__popc(hom & r0.y) + ((uint32_t)__popc(hom & r0.z) << 1) + ((uint32_t)__popc(hom & r0.w) << 2);
            const uint32_t ncov = __builtin_amdgcn_readfirstlane(r0.x) >> 24;
            if (ncov > 3) {                      // deep rows (cov >= 8): max_cov < 128, seven planes at most
                const uint4 r1 = rec[2 * (s - seg0) + 1];
                CH += ((uint32_t)__popc(hom & r1.x) << 3) + ((uint32_t)__popc(hom & r1.y) << 4) +
                      ((uint32_t)__popc(hom & r1.z) << 5) + ((uint32_t)__popc(hom & r1.w) << 6);
            }
        }

        // ---- the window's end: every lane finishes its individual against 8 comparison individuals
        const uint32_t C0 = from_upper_half((uint32_t)acc0[14]), a0 = from_upper_half((uint32_t)acc0[15]);
        const uint32_t C1 = from_upper_half((uint32_t)acc1[14]), a1 = from_upper_half((uint32_t)acc1[15]);
        const int eK = (int)kc.x;
        double P2;
        {
            // pDg[x0+x1] (ibdgem.c:715): E3 = C0 + C1 - 2 CH, E2 = AT - a0 - a1 + CH
            uint4 p1, p2;
            lds_read2(p1, p2, lshl_add<4>(CH - (a0 + a1), kc.y), lshl_add<4>(mad24<-2>(CH, C0 + C1), kc.z));
            P2 = ld_value(eK, p1, p2);
        }
        const uint32_t a0m = 0u - 16u * a0, a1m = 0u - 16u * a1, C0s = 16u * C0, C1s = 16u * C1;
        // (the slots' constants through the scalar path instead of LDS, both halves' and a select: 12 % slower)
        const uint32_t slot_addr = (uint32_t)(uintptr_t)(lds_void *)wcs + ((w - w0) * 16 + 2 * h) * 16;
        // ((Q00 + Q01) + Q10) + Q11 of comparison individual slot_of(h, i) (ibdgem.c:716-719, :744-745)
        auto four_products = [&](uint32_t i) -> double {
            uint32_t G00 = (uint32_t)acc0[2 * i], G10 = (uint32_t)acc0[2 * i + 1];
            uint32_t G01 = (uint32_t)acc1[2 * i], G11 = (uint32_t)acc1[2 * i + 1];
            if (i == 7) {                       // the upper half's last pair is the weights' own rows
                G00 = h ? 0u : G00;
                G10 = h ? 0u : G10;
                G01 = h ? 0u : G01;
                G11 = h ? 0u : G11;
            }
            const uint4 kt = lds_read_b128(slot_addr + (4 * (i >> 1) + (i & 1)) * 16);      // slot_of(h, i)
            const uint32_t kc0 = kt.x, kc1 = kt.y, kb0 = kt.z, kb1 = kt.w;
            // pDg[At+hx]: E3 = <t,cov> + Cx - 2 G(x,t), E2 = AT - <t,alt> - ax + G(x,t)
            // (-16 a and 16 C are taken once per window: the sums below are plain additions, which issue at twice
            // the rate of the shift-and-add forms)
            uint32_t ad[8];
            ad[0] = lshl_add<4>(G00, kb0 + a0m);   ad[1] = mad24r(G00, m32, kc0 + C0s);
            ad[2] = lshl_add<4>(G01, kb0 + a1m);   ad[3] = mad24r(G01, m32, kc0 + C1s);
            ad[4] = lshl_add<4>(G10, kb1 + a0m);   ad[5] = mad24r(G10, m32, kc1 + C0s);
            ad[6] = lshl_add<4>(G11, kb1 + a1m);   ad[7] = mad24r(G11, m32, kc1 + C1s);
            uint4 pw[8];
            // (tried: two of the eight through the vector memory path, 30 % slower; plain C++ LDS loads that hipcc
            // may schedule across comparison individuals instead of this statement with its own wait, 7 % slower)
            lds_read_pow8(pw, ad);
            const double Q00 = ld_value(eK, pw[0], pw[1]);
            const double Q01 = ld_value(eK, pw[2], pw[3]);
            const double Q10 = ld_value(eK, pw[4], pw[5]);
            const double Q11 = ld_value(eK, pw[6], pw[7]);
            return ((Q00 + Q01) + Q10) + Q11;
        };
        double s1[8];                                       // the IBD1 addends of the lane's eight comparison individuals
        // (a short group occupies the first register pairs only: slots 4j .. 4j+3 are pairs 2j, 2j+1 of the two halves)
#pragma unroll
        for (uint32_t i = 0; i < 8; ++i)
            s1[i] = i < n_iter ? ((excl >> i) & 1 ? 0.0 : wgt) * four_products(i) : 0.0;          // :744-745
        const double wP2 = wgt * P2;                        // :743
        // The 32 individuals of the half chunk are summed in the tree of wave_sum2 (neighbours first).
        if (!any_excl) {
            // No lane of the wave is one of the group's comparison individuals (all but a few waves): the IBD0
            // addends are the same for the eight of them -- one butterfly for that sum, the transposed reduction
            // for the eight IBD1 sums only.
            double s0 = wP2;
            s0 = s0 + swz_get<1>(s0);
            s0 = s0 + swz_get<2>(s0);
            s0 = swz_add<4>(s0);
            s0 = swz_add<8>(s0);
            s0 = swz_add<16>(s0);
            const double t1 = reduce8_halves(s1, n);
            if (st8_ok)
                *reinterpret_cast<double2 *>(a.partial + st8_row + (size_t)w * n_half * 2) = make_double2(s0, t1);
        } else {
            double sv[16];
#pragma unroll
            for (uint32_t i = 0; i < 8; ++i) {
                sv[2 * i] = (excl >> i) & 1 ? 0.0 : wP2;
                sv[2 * i + 1] = s1[i];
            }
            const double tot = reduce16_halves(sv, n);
            if (st_ok)
                a.partial[st_row + (size_t)w * n_half * 2] = tot;
        }
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
