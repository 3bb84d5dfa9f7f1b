// ibdg_ld_popcount.hip -- the fast --LD kernel: exponent counting on a tile-transposed panel.
//
// What it computes is what src/ibdgem.c:669-722 and :736-753 of the reference compute:
// for every background individual the five window products of P(D|G) factors, then the
// background averages.  How: every factor is one of (src/ibd-math.c:57-70)
//     pDg[0] = C (1-e)^r e^a      pDg[1] = C (1/2)^(r+a)      pDg[2] = C (1-e)^a e^r
// (C = binomial coefficient, r/a = n_ref/n_alt of the row, e = epsilon), so a product over the
// rows of a window is exactly
//     prod = K * (1-e)^E1 * e^E2 * 2^-E3,    K = prod C,
//          = K' * rho^E2 * sigma^E3,  K' = K (1-e)^reads, rho = e/(1-e), sigma = 1/(2(1-e)),
//     E3 = reads on rows where the genotype is 1, E2 = reads that contradict a homozygous
//     genotype (alt reads under 0, ref reads under 2), E1 = all reads - E2 - E3.
// E2 and E3 are integers: sums of small per-row weights over the rows selected by haplotype
// bits -- weighted popcounts.  With the panel transposed into 32-row tiles (one u32 per
// individual, haplotype and tile) a weighted popcount over 32 rows is, per bit-plane k of the
// weights, one v_and_b32 with a wave-uniform mask and one accumulating v_bcnt_u32_b32.
// Per individual and window nine such sums are needed (x0,x1 = its two haplotypes, t0,t1 the
// target's, cov = r+a):
//     A(x0) A(x1)             <x, alt>
//     C(x0) C(x1) C(x0&x1)    <x, cov>
//     G(x,t) for 4 pairs      <x & t, cov>
// and the exponents follow without further per-row work:
//     pDg[x0+x1]:  E3 = C(x0)+C(x1)-2C(x0&x1)        E2 = ALT - A(x0) - A(x1) + C(x0&x1)
//     pDg[t +x ]:  E3 = <t,cov> + C(x) - 2G(x,t)     E2 = ALT - <t,alt> - A(x) + G(x,t)
// The integers are exact, so rows may be visited in any grouping; the floating-point value
//     mK' * ldexp(m1[E2] * m2[E3], eK' + e1[E2] + e2[E3])
// (tables of rho^n and sigma^n as mantissa/exponent pairs, built on the host in extended
// precision; mK' is applied after the sum over the background) carries ~4 roundings, i.e. it is CLOSER to the exact product than the reference's
// 100 sequential multiplications; the two agree to ~1e-14 relative (documented bar: 1e-10).
// Values below the double range come out as 0/subnormal from the final ldexp, like the
// reference's running product.  The host enables this kernel only when the P(D|G) table is
// the unclamped binomial form (no DBL_MIN clamp, exact coefficients); otherwise the strict
// multiplying kernel in ibdg_kernels.hip is used.
//
// Round 4: for ONE comparison individual per workgroup the four sums of a haplotype word that a lane needs --
// <x,cov> <x,alt> <x & t0,cov> <x & t1,cov> -- come from one v_mfma_scale_f32_16x16x128_f8f6f4 with a block-diagonal FP6
// weight matrix and the word's bits as FP4 numbers (k_win_target_mx, lds_fetch_mx, IBDG_SEGMENT_MX; template parameter MX
// of k_ld_popcount; DESIGN.md s4.1, docs/DESIGN_rounds_1-4.md s4.1c); only <x0 & x1,cov> is still three (mask, count) pairs.  The sums are the same
// integers, the results the same bits.  The pairs described above remain the form of option mx_counts 0 and of the kernel
// for groups of four comparison individuals (k_ld_popcount_mt).
//
// Round 5, the IBD1 form (k_ld_popcount<.., IBD1 = true>; PopArgs::ibd1): the product of an individual's OWN
// genotype factors (pDg[x0+x1] above) does not depend on the comparison individual, so ONE launch per site list and background
// keeps it for every individual and window (PopArgs::p2_out) and later launches count the four pDg[t+x] products only: the
// rows' weights become cov (1 - 2 t) and t cov - alt (signed FP6), the instruction's four sums per word are
//     C(x) - 2 G(x,t0)   C(x) - 2 G(x,t1)   G(x,t0) - A(x)   G(x,t1) - A(x)
// -- E3 and E2 above up to the window's constants --, the accumulators start from 1.5 * 2^23 so that their bits ARE the sums,
// and the finalising step takes IBD0 from the kept products in the additions of a launch that counts everything
// (ibd0_from_pass, ibdg_ld_dev.h): the same bits from either form.  A new individual's images are bit selections between
// three fragments made once per site list (k_frag_base, k_win_target_x1).
#include "ibdg_kernels.h"
#include "ibdg_ld_dev.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>


// LDS image of a segment (8 words, 16-byte aligned):
//   flags | cov0 cov1 cov2 | alt0 alt1 | target words t0 t1
// The masks of the rare higher weight bit-planes (cov3.., alt2..) stay in the global Seg array and
// are fetched with scalar loads by the few segments that have them (flags bit 12) -- keeping them
// out of LDS is what lets a fourth workgroup fit on a CU.
// flags = ring slot of the NEXT segment's pair (3) | its tile half (1) | pairs to advance before it (8)
//       | rare planes present (1) | last segment of its window (1) | - | ncov (8) | nalt (8)   (host-built)
// The hot half is read with two BROADCAST ds_read_b128 (every lane the same address), so the
// masks land in VGPRs: on gfx950 a VALU instruction with an SGPR operand issues at half the rate
// of one with VGPR operands only (tools/ubench/issue_rates.hip: v_and_b32 4.1 vs 2.4 cycles),
// and a wave-uniform mask is just as good in a VGPR.
#define IBDG_REC_WORDS 8
// LDS image of a window's constants (8 words), all but eK already table BYTE OFFSETS (16 bytes per
// entry), so that the window end forms its ten table addresses without a shift of their own:
//   eK  16*AT  16*<t0,cov>  16*<t1,cov> | 16*(AT-<t0,alt>)  16*(AT-<t1,alt>)  0  -
// (k_win_target); while a workgroup stages them it adds the LDS address of the table each one indexes
// (rho^n: words 1, 4, 5; sigma^n: words 2, 3, 6), see stage_wc_base.
#define IBDG_WC_WORDS 8
// LDS image of a segment for the counts on the matrix cores (k_ld_popcount<.., MX = true>; 32 words):
//   flags cov0 cov1 cov2 | t0 t1 - - | four A fragments of 24 bytes: the rows' weights in FP6 for the sums
//   <x,cov> <x,alt> <x & t0,cov> <x & t1,cov>  (k_win_target_mx; flags bit 12 there: planes beyond cov 0-2 / alt 0-2)
#define IBDG_RECX_WORDS 32
// cache policy of the tile stream's direct-to-LDS loads (aux of global_load_lds: 0 default, 2 = nt, non-temporal): every tile
// pair is read once by one wave, 2.56 GB per launch = ten times the last-level cache.  nt measured 1.5-2 % faster per step
// on one box, builds alternating (profiles/r05_ab_nt.txt: 0.606 / 0.607 against 0.617 / 0.615 ms with a new individual per
// step, 0.581 / 0.586 against 0.596 / 0.596 with the same one); -DIBDG_TILE_AUX=0 brings the default policy back.
#ifndef IBDG_TILE_AUX
#define IBDG_TILE_AUX 2
#endif

namespace ibdg {

typedef int mx_v8i __attribute__((ext_vector_type(8)));
typedef float mx_v4f __attribute__((ext_vector_type(4)));
typedef uint32_t mx_u4 __attribute__((ext_vector_type(4)));      // (plain vector types: the struct ones cannot be tied asm operands)
typedef uint32_t mx_u2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// Panel transposition (once per upload): site-major rows ->
//     t32[chunk][tile_pair][lane] = uint4 { x0(tile 2q), x1(2q), x0(2q+1), x1(2q+1) }
// x0/x1 = first/second haplotype of individual 64*chunk+lane, bit j = row 32*tile + j.
// A wave reads one tile pair as one fully coalesced 1 KiB global_load_dwordx4, and the four
// pairs of an 8-tile "oct" are 4 KiB contiguous.  Tiles are padded to whole octs (zero bits).
//
// One wave per (tile pair, chunk): lane j loads the chunk's two haplotype words of row 64 q + j (16 bytes;
// the eight waves of a workgroup take eight neighbouring chunks, i.e. whole 128-byte pieces of the rows), and
// the two 64 x 64 bit matrices (rows on lanes, individuals on bits) are transposed in registers by the
// recursive block exchange: at block size s lane l and lane l ^ s swap the off-diagonal s x s blocks --
//     l & s == 0:  w = (w & K) | (t << s & ~K),      l & s != 0:  w = (w & ~K) | (t >> s & K),
// t = the partner's word, K = the bits whose index has bit s clear -- as ONE v_alignbit (a rotation by s or
// 32 - s, whichever the lane needs) and ONE v_bfi per 32-bit word and step; the exchanges are a
// v_permlane32_swap (s = 32), ds_swizzle (16, 4) and DPP moves (8, 2, 1).  ~65 vector instructions per KiB,
// where the first version (every lane picking its bit out of 128 wave-uniform row words) spent ~400 and
// read every row word through the scalar cache: 9.4 ms for the 2.56 GB panel then.
// ---------------------------------------------------------------------------
template <int S>
__device__ __forceinline__ uint32_t partner_word(uint32_t w)
{
    if (S == 16 || S == 4)
        return (uint32_t)__builtin_amdgcn_ds_swizzle((int)w, (S << 10) | 0x1f);
    constexpr int ctrl = S == 8 ? 0x128 /* row_ror:8 */ : (S == 2 ? 0x4E /* quad_perm [2,3,0,1] */ : 0xB1 /* [1,0,3,2] */);
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, ctrl, 0xf, 0xf, true);
}

template <int S>
__device__ __forceinline__ uint32_t block_exchange(uint32_t w, uint32_t lane)
{
    constexpr uint32_t K = S == 16 ? 0x0000ffffu : S == 8 ? 0x00ff00ffu : S == 4 ? 0x0f0f0f0fu : S == 2 ? 0x33333333u : 0x55555555u;
    const bool up = lane & S;
    const uint32_t t = partner_word<S>(w);
    const uint32_t rot = __builtin_amdgcn_alignbit(t, t, up ? S : 32 - S);      // t >> s (up) or t << s, as a rotation
    const uint32_t keep = up ? ~K : K;
    return (w & keep) | (rot & ~keep);                                           // v_bfi_b32
}

// the two 64 x 64 bit matrices of a wave (lane = row; {plane 0 lo, hi, plane 1 lo, hi}) transposed in registers:
// afterwards lane = individual, the return value the uint4 of the layout above
__device__ __forceinline__ uint4 transpose_pair(uint4 w, uint32_t lane)
{
    // s = 32: the high halves of lanes 0..31 and the low halves of lanes 32..63 change places
    {
        auto p0 = __builtin_amdgcn_permlane32_swap(w.x, w.y, false, false);
        w.x = p0[0];
        w.y = p0[1];
        auto p1 = __builtin_amdgcn_permlane32_swap(w.z, w.w, false, false);
        w.z = p1[0];
        w.w = p1[1];
    }
#define IBDG_T_STEP(S)                       \
    w.x = block_exchange<S>(w.x, lane);      \
    w.y = block_exchange<S>(w.y, lane);      \
    w.z = block_exchange<S>(w.z, lane);      \
    w.w = block_exchange<S>(w.w, lane);
    IBDG_T_STEP(16)
    IBDG_T_STEP(8)
    IBDG_T_STEP(4)
    IBDG_T_STEP(2)
    IBDG_T_STEP(1)
#undef IBDG_T_STEP
    // w.x / w.y = rows 0..31 / 32..63 of the first haplotype, w.z / w.w of the second
    return make_uint4(w.x, w.z, w.y, w.w);
}

__global__ __launch_bounds__(512) void k_transpose32(const uint64_t *__restrict__ panel,
                                                     uint32_t stride, size_t n_rows,
                                                     uint32_t n_chunks, uint32_t n_pairs,
                                                     uint4 *__restrict__ t32)
{
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    const unsigned c = blockIdx.y * 8 + wave;
    if (c >= n_chunks)
        return;
    const uint32_t pair = blockIdx.x;
    const size_t r = (size_t)pair * 64 + lane;
    uint4 w = make_uint4(0, 0, 0, 0);                 // {plane 0 lo, hi, plane 1 lo, hi} of row r
    if (r < n_rows)
        w = *reinterpret_cast<const uint4 *>(panel + r * stride + 2 * c);
    t32[((size_t)c * n_pairs + pair) * 64 + lane] = transpose_pair(w, lane);
}

// ---------------------------------------------------------------------------
// The compacted layout of ONE site list (once per ibdg_upload_sites, or once the runs on it have added up): only the
// rows that carry reads, in the order of the site list --
//     virtual row v = (j / W) * R + j % W      for covered row j (W = window, R = virtual rows per window)
// R = W (the default since round 5): the rows back to back, v = j, no padding -- a window of 100 rows spans 3.1 tiles and
// is cut into 4.1 segments where the panel's own tiles (13.5 % rows without reads) make it 3.6 tiles / 4.6 segments;
// R = 32 * TPW, TPW = ceil(W / 32) (round 4, option "compact_align" 32): every window starts on a tile boundary, 4 segments
// per window of 100 but 28 % of the tile words are padding
// -- gathered from the site-major panel through the covered-row list and transposed like above, same uint4
// layout, so the --LD kernels run on it unchanged (their segments are cut from the virtual rows,
// ibdg_prep.hip).  In the reference the rows a window multiplies are the rows that passed the filter chain
// and carry reads (src/ibdgem.c:596-601, :657-663), however far apart they lie in the panel: here a window
// costs TPW tile words whatever the pileup's density, where the in-place tiles cost one word per 32 PANEL
// rows between its first and last row; and no tile is shared by two windows (4 segments per window of 100
// rows instead of 4.6).  The rows of the virtual tiles beyond a window's W (and beyond the last covered row)
// are zero bits.  Access pattern as in k_transpose32: a lane fetches 16 bytes of its row, the eight waves of a
// workgroup eight neighbouring chunks = one 128-byte piece of each of the 64 rows.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_gather_transpose32(const uint64_t *__restrict__ panel, uint32_t stride,
                                                            const uint2 *__restrict__ rec_cov, uint32_t n_cov,
                                                            uint32_t window, uint32_t win_rows /* R */,
                                                            uint32_t n_chunks, uint32_t n_pairs,
                                                            uint4 *__restrict__ t32)
{
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    const unsigned c = blockIdx.y * 8 + wave;
    if (c >= n_chunks)
        return;
    const uint32_t pair = blockIdx.x;
    const uint64_t v = (uint64_t)pair * 64 + lane;
    uint64_t j = v;                      // rows back to back (win_rows == window, the default): virtual row = covered row
    bool in_window = true;
    if (win_rows != window) {            // (wave-uniform: an alignment was asked for)
        const uint64_t win = v / win_rows;
        const uint32_t k = (uint32_t)(v - win * win_rows);
        j = win * window + k;
        in_window = k < window;
    }
    uint4 w = make_uint4(0, 0, 0, 0);
    if (in_window && j < n_cov)
        w = *reinterpret_cast<const uint4 *>(panel + (size_t)rec_cov[j].x * stride + 2 * c);
    t32[((size_t)c * n_pairs + pair) * 64 + lane] = transpose_pair(w, lane);
}

// ---------------------------------------------------------------------------
// Per target, one thread per segment and eight per window: the LDS-ready images the --LD kernel stages
// with plain contiguous copies -- every segment's 8-word record (IBDG_REC_WORDS, layout above) with the
// target's haplotype words of its tile filled in, and the window's 8 constants (IBDG_WC_WORDS) built
// from <t0,cov>, <t1,cov>, <t0,alt>, <t1,alt> summed over the window's rows.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_win_target(PopArgs a, uint32_t *__restrict__ rec_ready,
                                                    uint32_t *__restrict__ wc_ready)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned t = blockIdx.y;
    uint32_t tgt = a.targets[a.t_base + t];
    IBDG_CHECK_TGT(tgt, a.lanes, __func__);
    const uint4 *tt = reinterpret_cast<const uint4 *>(a.t32) + (size_t)(tgt >> 6) * a.n_pairs * 64 + (tgt & 63);
    if (i < a.n_segs) {                      // thread i: the record of segment i
        const Seg S = a.segs[i];
        const uint2 at = tile_words(tt, S.tile);
        uint4 *o = reinterpret_cast<uint4 *>(rec_ready + ((size_t)t * a.n_segs + i) * IBDG_REC_WORDS);
        o[0] = make_uint4(S.flags, S.cov[0], S.cov[1], S.cov[2]);
        o[1] = make_uint4(S.alt[0], S.alt[1], at.x, at.y);
    }
    if ((i >> 3) < a.n_win) {                // threads 8w..8w+7: the constants of window w
        const uint32_t w = i >> 3;
        uint32_t a0cov = 0, a1cov = 0, a0alt = 0, a1alt = 0;
        const uint32_t s1 = a.wconst[w + 1].seg_begin;
        for (uint32_t s = a.wconst[w].seg_begin + (i & 7); s < s1; s += 8) {    // a window has ~4-5 segments
            const Seg &S = a.segs[s];
            const uint2 at = tile_words(tt, S.tile);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a0cov += (uint32_t)__popc(at.x & S.cov[k]) << k;
                a1cov += (uint32_t)__popc(at.y & S.cov[k]) << k;
                a0alt += (uint32_t)__popc(at.x & S.alt[k]) << k;
                a1alt += (uint32_t)__popc(at.y & S.alt[k]) << k;
            }
        }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {    // the 8 lanes of a window sit in one wave (8 | 64, i is 8-aligned)
            a0cov += __shfl_xor(a0cov, m);
            a1cov += __shfl_xor(a1cov, m);
            a0alt += __shfl_xor(a0alt, m);
            a1alt += __shfl_xor(a1alt, m);
        }
        if ((i & 7) == 0) {
            const uint32_t *wcs = reinterpret_cast<const uint32_t *>(a.wconst + w);    // mK(2) eK ct at seg_begin
            uint4 *o = reinterpret_cast<uint4 *>(wc_ready + ((size_t)t * a.n_win + w) * IBDG_WC_WORDS);
            const uint32_t AT = wcs[4];
            o[0] = make_uint4(wcs[2], 16 * AT, 16 * a0cov, 16 * a1cov);
            o[1] = make_uint4(16 * (AT - a0alt), 16 * (AT - a1alt), 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------
// The same for the counts on the matrix cores (round 4; DESIGN.md s4.1, docs/DESIGN_rounds_1-4.md s4.1c).
//
// v_mfma_scale_f32_16x16x128_f8f6f4 multiplies a 16 x 128 matrix A by a 128 x 16 matrix B; lane l holds 32 K-elements
// of row (A) / column (B) l % 16: k = 32 (l / 16) .. + 31.  A is made block diagonal,
//     A[4 kb' + sum][32 kb + r] = weight_sum[r]  if kb == kb'  else 0
// and every lane supplies ITS OWN tile word as "column l % 16, K block l / 16" (its bits as FP4 numbers).  Then
//     D[4 kb + sum][n] = sum_r weight_sum[r] * bit_r(word of lane n + 16 kb)
// and the C/D layout (column = lane % 16, rows 4 (lane / 16) .. + 3 in the lane's four registers) returns to every lane
// the four weighted sums of its own word: <x,cov> <x,alt> <x & t0,cov> <x & t1,cov> -- one instruction for the twelve
// (mask, count) pairs of a haplotype word, no lane movement.  Bits become FP4 (e2m1) without shifts where possible:
//     dword 0 = x & 0x11111111  rows 4j     value 0.5      dword 2 = x & 0x44444444         rows 4j + 2  value 2
//     dword 1 = x & 0x22222222  rows 4j + 1 value 1        dword 3 = (x >> 3) & 0x11111111  rows 4j + 3  value 0.5
// (nibble 1000 is -0: useless) and A carries w, w/2, w/4, w in FP6 e2m3, exact for w = 0..7, with the block scale 2:
// every product is w, the sums are exact integers in f32 (tools/ubench/fp4_count.hip checks the layout with random
// words and weights).  Only the 16 lanes with l % 16 / 4 == l / 16 hold a non-zero A fragment: lanes 20 kb + sum read
// the 24 bytes of `sum` from the segment's record, the others keep zeros.
// Element k = 8 d + j of a lane's 32  <->  row r = 4 j + d of the tile.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fp6_weight_code(uint32_t w, int d)
{
    // e2m3 (bias 1, exponent 0 = subnormal m/8) of w (d = 0, 3), w/2 (d = 1), w/4 (d = 2), w = 0..7, as bytes of two words
    const uint32_t lo = d == 1 ? 0x0c080400u : d == 2 ? 0x06040200u : 0x14100800u;
    const uint32_t hi = d == 1 ? 0x16141210u : d == 2 ? 0x0e0c0a08u : 0x1e1c1a18u;
    return ((w & 4 ? hi : lo) >> (8 * (w & 3))) & 0xffu;
}

__global__ __launch_bounds__(256) void k_win_target_mx(PopArgs a, uint32_t *__restrict__ rec_ready,
                                                       uint32_t *__restrict__ wc_ready)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned t = blockIdx.y;
    uint32_t tgt = a.targets[a.t_base + t];
    IBDG_CHECK_TGT(tgt, a.lanes, __func__);
    const uint4 *tt = reinterpret_cast<const uint4 *>(a.t32) + (size_t)(tgt >> 6) * a.n_pairs * 64 + (tgt & 63);
    if ((i >> 2) < a.n_segs) {               // threads 4s .. 4s+3: the four A fragments of segment s
        const uint32_t sg = i >> 2, sum = i & 3;
        const Seg &S = a.segs[sg];
        IBDG_CHECK_IDX(S.tile, 2 * a.n_pairs, "k_win_target_mx tile");
        const uint2 at = tile_words(tt, S.tile);
        // the rows' weights of this thread's sum as three bit planes of the magnitude and one of the sign (bit 5 of the e2m3 code):
        //   <x,cov> <x,alt> <x & t0,cov> <x & t1,cov>                            -- the form that counts everything
        //   <x, cov (1 - 2 t0)>  <x, cov (1 - 2 t1)>  <x, t0 cov - alt>  <x, t1 cov - alt>   -- ibd1: C(x) - 2 G(x,t) and G(x,t) - A(x),
        //   the table exponents of the IBD1 products up to the window's constants; |weight| <= 7
        uint32_t p0, p1, p2, neg = 0;
        if (!a.ibd1) {
            if (sum == 1) {
                p0 = S.alt[0]; p1 = S.alt[1]; p2 = S.alt[2];
            } else {
                const uint32_t m = sum == 0 ? 0xffffffffu : sum == 2 ? at.x : at.y;
                p0 = S.cov[0] & m; p1 = S.cov[1] & m; p2 = S.cov[2] & m;
            }
        } else {
            const uint32_t tb = (sum & 1) ? at.y : at.x;
            if (sum < 2) {
                p0 = S.cov[0]; p1 = S.cov[1]; p2 = S.cov[2];
                neg = tb;                                       // (-0 where the row has no reads: adds nothing)
            } else {
                // t cov - alt for the 32 rows at once, bit-sliced: a three-bit subtraction, then the magnitude of the negative ones
                const uint32_t x0 = S.cov[0] & tb, x1 = S.cov[1] & tb, x2 = S.cov[2] & tb;
                const uint32_t y0 = S.alt[0], y1 = S.alt[1], y2 = S.alt[2];
                const uint32_t d0 = x0 ^ y0, b0 = ~x0 & y0;
                const uint32_t e1 = x1 ^ y1, d1 = e1 ^ b0, b1 = (~x1 & y1) | (~e1 & b0);
                const uint32_t e2 = x2 ^ y2, d2 = e2 ^ b1;
                neg = (~x2 & y2) | (~e2 & b1);                  // the borrow out of bit 2: the difference is negative
                const uint32_t r1 = ~d1 ^ ~d0, r2 = ~d2 ^ (~d1 & ~d0);      // -d = ~d + 1 (bit 0 stays)
                p0 = d0;
                p1 = (d1 & ~neg) | (r1 & neg);
                p2 = (d2 & ~neg) | (r2 & neg);
            }
        }
        uint32_t f[6] = {0, 0, 0, 0, 0, 0};   // 32 x 6 bits
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int d = k >> 3, r = 4 * (k & 7) + d;
            const uint32_t w = ((p0 >> r) & 1u) | (((p1 >> r) & 1u) << 1) | (((p2 >> r) & 1u) << 2);
            const uint32_t code = fp6_weight_code(w, d) | (((neg >> r) & 1u) << 5);
            const int pos = 6 * k, wd = pos >> 5, sh = pos & 31;
            f[wd] |= code << sh;
            if (sh > 26)
                f[wd + 1] |= code >> (32 - sh);
        }
        uint32_t *o = rec_ready + ((size_t)t * a.n_segs + sg) * IBDG_RECX_WORDS;
        uint2 *fo = reinterpret_cast<uint2 *>(o + 8 + 6 * sum);
        fo[0] = make_uint2(f[0], f[1]);
        fo[1] = make_uint2(f[2], f[3]);
        fo[2] = make_uint2(f[4], f[5]);
        if (sum == 0) {
            // control word of this form: bits 0-13 ring byte offset of the NEXT segment's tile words, 14 planes beyond
            // cov 0-2 / alt 0-2 present, 15 last segment of its window, 16-23 tile pairs to advance before the next segment
            const uint32_t ncov = (S.flags >> 16) & 0xff, nalt = S.flags >> 24;
            const uint32_t fl = ((S.flags & 7) * 1024 + ((S.flags >> 3) & 1) * 8) | ((ncov > 3 || nalt > 3) ? 1u << 14 : 0u) |
                                (((S.flags >> 13) & 1) << 15) | (((S.flags >> 4) & 0xff) << 16);
            uint4 *oh = reinterpret_cast<uint4 *>(o);
            oh[0] = make_uint4(fl, S.cov[0], S.cov[1], S.cov[2]);
            oh[1] = make_uint4(at.x, at.y, ncov | (nalt << 8), 0);
        }
    }
    if ((i >> 3) < a.n_win) {                // threads 8w..8w+7: the constants of window w (as k_win_target)
        const uint32_t w = i >> 3;
        uint32_t a0cov = 0, a1cov = 0, a0alt = 0, a1alt = 0;
        const uint32_t s1 = a.wconst[w + 1].seg_begin;
        for (uint32_t s = a.wconst[w].seg_begin + (i & 7); s < s1; s += 8) {
            const Seg &S = a.segs[s];
            const uint2 at = tile_words(tt, S.tile);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a0cov += (uint32_t)__popc(at.x & S.cov[k]) << k;
                a1cov += (uint32_t)__popc(at.y & S.cov[k]) << k;
                a0alt += (uint32_t)__popc(at.x & S.alt[k]) << k;
                a1alt += (uint32_t)__popc(at.y & S.alt[k]) << k;
            }
        }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            a0cov += __shfl_xor(a0cov, m);
            a1cov += __shfl_xor(a1cov, m);
            a0alt += __shfl_xor(a0alt, m);
            a1alt += __shfl_xor(a1alt, m);
        }
        if ((i & 7) == 0) {
            const uint32_t *wcs = reinterpret_cast<const uint32_t *>(a.wconst + w);
            uint4 *o = reinterpret_cast<uint4 *>(wc_ready + ((size_t)t * a.n_win + w) * IBDG_WC_WORDS);
            const uint32_t AT = wcs[4];
            // byte offsets into tables of 8-byte entries (the matrix-core form's power tables, see its window end)
            const uint32_t sc = a.tab_in_lds ? 8 : 16;       // (16-byte entries where the tables stay in global memory)
            // (ibd1: the sums reach the window end as the bits of 1.5 * 2^23 + sum, whose low 24 bits are 2^22 + sum: the constants
            //  take 8 * 2^22 back, modulo 2^32 like the address arithmetic they enter)
            const uint32_t bias = a.ibd1 ? 1u << 25 : 0u;
            o[0] = make_uint4(wcs[2], sc * AT, sc * a0cov - bias, sc * a1cov - bias);
            o[1] = make_uint4(sc * (AT - a0alt) - bias, sc * (AT - a1alt) - bias, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------
// The IBD1 form's images in two steps (round 5).  What k_win_target_mx builds per comparison individual and segment -- four
// fragments of 32 signed FP6 weights -- depends on the individual only through WHICH of two values a row's weight takes:
//     sums 0 / 1  cov (1 - 2 t):   +cov or -cov   = the code of cov with the sign bit of the rows where t is set
//     sums 2 / 3  t cov - alt:     -alt or cov - alt
// so the three fragments COV, F0 = code(-alt), F1 = code(cov - alt) are made ONCE per site list (k_frag_base, 72 bytes per
// segment), and an individual's images are bit selections between them: its tile word's 32 bits spread into 32 six-bit
// fields (M: 0x3f where the row's t is set), by a 256-entry table a byte at a time --
//     sum 0 / 1 = COV | (M & SIGN)        sum 2 / 3 = (F1 & M) | (F0 & ~M)
// -- ~100 instructions for a segment's two fragments of one target haplotype word instead of ~500 per fragment: the kernel
// that runs beside the previous step's --LD kernel for every NEW individual costs that kernel a third of what it did.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void fp6_fragment(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t neg, uint32_t (&f)[6])
{
#pragma unroll
    for (int i = 0; i < 6; ++i)
        f[i] = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int d = k >> 3, r = 4 * (k & 7) + d;
        const uint32_t w = ((p0 >> r) & 1u) | (((p1 >> r) & 1u) << 1) | (((p2 >> r) & 1u) << 2);
        const uint32_t code = fp6_weight_code(w, d) | (((neg >> r) & 1u) << 5);
        const int pos = 6 * k, wd = pos >> 5, sh = pos & 31;
        f[wd] |= code << sh;
        if (sh > 26)
            f[wd + 1] |= code >> (32 - sh);
    }
}

// t cov - alt for 32 rows at once, bit-sliced (cov planes x, alt planes y): a three-bit subtraction, then the magnitude of
// the negative ones
__device__ __forceinline__ void sliced_diff(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t y0, uint32_t y1, uint32_t y2,
                                            uint32_t &p0, uint32_t &p1, uint32_t &p2, uint32_t &neg)
{
    const uint32_t d0 = x0 ^ y0, b0 = ~x0 & y0;
    const uint32_t e1 = x1 ^ y1, d1 = e1 ^ b0, b1 = (~x1 & y1) | (~e1 & b0);
    const uint32_t e2 = x2 ^ y2, d2 = e2 ^ b1;
    neg = (~x2 & y2) | (~e2 & b1);                  // the borrow out of bit 2: the difference is negative
    const uint32_t r1 = ~d1 ^ ~d0, r2 = ~d2 ^ (~d1 & ~d0);      // -d = ~d + 1 (bit 0 stays)
    p0 = d0;
    p1 = (d1 & ~neg) | (r1 & neg);
    p2 = (d2 & ~neg) | (r2 & neg);
}

// once per site list: [segment][COV, F0, F1][6 words]
__global__ __launch_bounds__(256) void k_frag_base(PopArgs a, uint32_t *__restrict__ frag_base)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t sg = i / 3, which = i - 3 * sg;
    if (sg >= a.n_segs)
        return;
    const Seg &S = a.segs[sg];
    uint32_t p0, p1, p2, neg;
    if (which == 0) {
        p0 = S.cov[0]; p1 = S.cov[1]; p2 = S.cov[2]; neg = 0;
    } else if (which == 1) {
        p0 = S.alt[0]; p1 = S.alt[1]; p2 = S.alt[2]; neg = 0xffffffffu;          // (-0 where the row has no alt read)
    } else {
        sliced_diff(S.cov[0], S.cov[1], S.cov[2], S.alt[0], S.alt[1], S.alt[2], p0, p1, p2, neg);
    }
    uint32_t f[6];
    fp6_fragment(p0, p1, p2, neg, f);
    uint2 *o = reinterpret_cast<uint2 *>(frag_base + ((size_t)sg * 3 + which) * 6);
    o[0] = make_uint2(f[0], f[1]);
    o[1] = make_uint2(f[2], f[3]);
    o[2] = make_uint2(f[4], f[5]);
}

// per comparison individual: two threads per segment (one per haplotype word of the individual), eight per window
__global__ __launch_bounds__(256) void k_win_target_x1(PopArgs a, const uint32_t *__restrict__ frag_base,
                                                       uint32_t *__restrict__ rec_ready, uint32_t *__restrict__ wc_ready)
{
    // eight rows' bits -> eight six-bit fields of ones (48 bits), one entry per thread of the workgroup
    __shared__ uint2 spread8[256];
    {
        const uint32_t b = threadIdx.x;
        uint64_t m = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            m |= (b >> j) & 1u ? (uint64_t)0x3f << (6 * j) : 0;
        spread8[b] = make_uint2((uint32_t)m, (uint32_t)(m >> 32));
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned t = blockIdx.y;
    uint32_t tgt = a.targets[a.t_base + t];
    IBDG_CHECK_TGT(tgt, a.lanes, __func__);
    const uint4 *tt = reinterpret_cast<const uint4 *>(a.t32) + (size_t)(tgt >> 6) * a.n_pairs * 64 + (tgt & 63);
    if ((i >> 1) < a.n_segs) {
        const uint32_t sg = i >> 1, ts = i & 1;
        const Seg &S = a.segs[sg];
        IBDG_CHECK_IDX(S.tile, 2 * a.n_pairs, "k_win_target_x1 tile");
        const uint2 at = tile_words(tt, S.tile);
        const uint32_t tw = ts ? at.y : at.x;
        // M: element k = 8 d + j of the fragment is row 4 j + d of the tile
        uint32_t m[6];
        {
            uint2 e[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint32_t x = (tw >> d) & 0x11111111u;           // rows d, 4 + d, ..., 28 + d at bits 0, 4, ..., 28
                x = (x | (x >> 3)) & 0x03030303u;
                x = (x | (x >> 6)) & 0x000f000fu;
                x = (x | (x >> 12)) & 0xffu;
                e[d] = spread8[x];
            }
            m[0] = e[0].x;
            m[1] = e[0].y | (e[1].x << 16);
            m[2] = (e[1].x >> 16) | (e[1].y << 16);
            m[3] = e[2].x;
            m[4] = e[2].y | (e[3].x << 16);
            m[5] = (e[3].x >> 16) | (e[3].y << 16);
        }
        const uint2 *fb = reinterpret_cast<const uint2 *>(frag_base + (size_t)sg * 18);
        const uint2 c0 = fb[0], c1 = fb[1], c2 = fb[2];         // COV
        const uint2 u0 = fb[3], u1 = fb[4], u2 = fb[5];         // F0
        const uint2 v0 = fb[6], v1 = fb[7], v2 = fb[8];         // F1
        // bit 5 of every six-bit field: the pattern repeats after 96 bits
        constexpr uint32_t SG0 = 0x20820820u, SG1 = 0x08208208u, SG2 = 0x82082082u;
        uint32_t *o = rec_ready + ((size_t)t * a.n_segs + sg) * IBDG_RECX_WORDS;
        uint2 *fa = reinterpret_cast<uint2 *>(o + 8 + 6 * ts), *fd = reinterpret_cast<uint2 *>(o + 8 + 6 * (2 + ts));
        fa[0] = make_uint2(c0.x | (m[0] & SG0), c0.y | (m[1] & SG1));
        fa[1] = make_uint2(c1.x | (m[2] & SG2), c1.y | (m[3] & SG0));
        fa[2] = make_uint2(c2.x | (m[4] & SG1), c2.y | (m[5] & SG2));
        fd[0] = make_uint2((v0.x & m[0]) | (u0.x & ~m[0]), (v0.y & m[1]) | (u0.y & ~m[1]));
        fd[1] = make_uint2((v1.x & m[2]) | (u1.x & ~m[2]), (v1.y & m[3]) | (u1.y & ~m[3]));
        fd[2] = make_uint2((v2.x & m[4]) | (u2.x & ~m[4]), (v2.y & m[5]) | (u2.y & ~m[5]));
        if (ts == 0) {
            const uint32_t ncov = (S.flags >> 16) & 0xff, nalt = S.flags >> 24;
            const uint32_t fl = ((S.flags & 7) * 1024 + ((S.flags >> 3) & 1) * 8) | ((ncov > 3 || nalt > 3) ? 1u << 14 : 0u) |
                                (((S.flags >> 13) & 1) << 15) | (((S.flags >> 4) & 0xff) << 16);
            uint4 *oh = reinterpret_cast<uint4 *>(o);
            oh[0] = make_uint4(fl, S.cov[0], S.cov[1], S.cov[2]);
            oh[1] = make_uint4(at.x, at.y, ncov | (nalt << 8), 0);
        }
    }
    if ((i >> 3) < a.n_win) {                // threads 8w..8w+7: the constants of window w (as k_win_target_mx, ibd1)
        const uint32_t w = i >> 3;
        uint32_t a0cov = 0, a1cov = 0, a0alt = 0, a1alt = 0;
        const uint32_t s1 = a.wconst[w + 1].seg_begin;
        for (uint32_t s = a.wconst[w].seg_begin + (i & 7); s < s1; s += 8) {
            const Seg &S = a.segs[s];
            const uint2 at = tile_words(tt, S.tile);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a0cov += (uint32_t)__popc(at.x & S.cov[k]) << k;
                a1cov += (uint32_t)__popc(at.y & S.cov[k]) << k;
                a0alt += (uint32_t)__popc(at.x & S.alt[k]) << k;
                a1alt += (uint32_t)__popc(at.y & S.alt[k]) << k;
            }
        }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) {
            a0cov += __shfl_xor(a0cov, m);
            a1cov += __shfl_xor(a1cov, m);
            a0alt += __shfl_xor(a0alt, m);
            a1alt += __shfl_xor(a1alt, m);
        }
        if ((i & 7) == 0) {
            const uint32_t *wcs = reinterpret_cast<const uint32_t *>(a.wconst + w);
            uint4 *o = reinterpret_cast<uint4 *>(wc_ready + ((size_t)t * a.n_win + w) * IBDG_WC_WORDS);
            const uint32_t AT = wcs[4];
            const uint32_t bias = 1u << 25;              // (see k_win_target_mx)
            o[0] = make_uint4(wcs[2], 8 * AT, 8 * a0cov - bias, 8 * a1cov - bias);
            o[1] = make_uint4(8 * (AT - a0alt) - bias, 8 * (AT - a1alt) - bias, 0, 0);
        }
    }
}

// the bits of a tile word as 32 FP4 numbers (0 or 0.5 / 1 / 2 / 0.5 by dword, see above)
__device__ __forceinline__ mx_v8i bits_to_fp4(uint32_t x)
{
    mx_v8i b = {0, 0, 0, 0, 0, 0, 0, 0};
    b[0] = (int)(x & 0x11111111u);
    b[1] = (int)(x & 0x22222222u);
    b[2] = (int)(x & 0x44444444u);
    b[3] = (int)((x >> 3) & 0x11111111u);
    return b;
}

// the lanes that hold a non-zero A fragment: 20 kb + sum
#define IBDG_MX_A_LANES 0xF0000F0000F0000Full

// One segment's LDS reads of the matrix-core form: the broadcast header (flags and the cov planes for the x0&x1 counts),
// the lane's two tile words and -- in the 16 lanes that carry one -- the A fragment; the other lanes keep their zeros.
__device__ __forceinline__ void lds_fetch_mx(uint4 &h0, uint2 &x, mx_u4 &a_lo, mx_u2 &a_hi, uint32_t rec_addr, uint32_t x_addr,
                                             uint32_t frag_addr)
{
    // (EXEC is narrowed to the fragment lanes for two reads and put back to what it WAS: a caller inside a divergent
    // branch keeps its dead lanes dead)
    uint64_t exec_was;
    asm volatile("ds_read_b128 %0, %5\n\t"
                 "ds_read_b64 %1, %6\n\t"
                 "s_mov_b64 %4, exec\n\t"
                 "s_and_b64 exec, %4, %8\n\t"
                 "ds_read2_b64 %2, %7 offset1:1\n\t"
                 "ds_read_b64 %3, %7 offset:16\n\t"
                 "s_mov_b64 exec, %4\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(h0), "=&v"(x), "+v"(a_lo), "+v"(a_hi), "=&s"(exec_was)
                 : "v"(rec_addr), "v"(x_addr), "v"(frag_addr), "s"((uint64_t)IBDG_MX_A_LANES)
                 : "memory", "scc");
}

// The same for the IBD1 form, from ONE address register: frag_base = record + 24 * (lane & 3) per lane.  The record's first
// word (its control word) is read by every lane at its own base -- lane 0's is the record itself, and only lane 0's copy is
// used (v_readfirstlane) --, the fragment of lane 20 kb + sum sits 32 bytes further on.
__device__ __forceinline__ void lds_fetch_x1(uint32_t &ctl, uint2 &x, mx_u4 &a_lo, mx_u2 &a_hi, uint32_t x_addr, uint32_t frag_base)
{
    uint64_t exec_was;
    asm volatile("ds_read_b32 %0, %6\n\t"
                 "ds_read_b64 %1, %5\n\t"
                 "s_mov_b64 %4, exec\n\t"
                 "s_and_b64 exec, %4, %7\n\t"
                 "ds_read2_b64 %2, %6 offset0:4 offset1:5\n\t"
                 "ds_read_b64 %3, %6 offset:48\n\t"
                 "s_mov_b64 exec, %4\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(ctl), "=&v"(x), "+v"(a_lo), "+v"(a_hi), "=&s"(exec_was)
                 : "v"(x_addr), "v"(frag_base), "s"((uint64_t)IBDG_MX_A_LANES)
                 : "memory", "scc");
}

// Two wave-wide sums at once through a 1 KiB LDS scratch of the wave: every lane writes its two
// addends into two arrays of 64 doubles; lane 32j+p then reads elements 2p, 2p+1 of sum j (one
// 16-byte read at scratch + 16*lane), adds them, and the 32 lanes of half j finish with five
// exchange-and-add steps.  Every lane of the first half ends up with the total of the first sum,
// every lane of the second half with the second.  Fixed order, no barrier (the scratch is the wave's
// own and a wave's LDS operations execute in order); 6 VALU instructions for the two sums of a window.
__device__ __forceinline__ double wave_sum2(double a, double b, uint32_t scr_w, uint32_t scr_r)
{
    uint4 r;
    asm volatile("ds_write_b64 %1, %2\n\t"
                 "ds_write_b64 %1, %3 offset:512\n\t"
                 "ds_read_b128 %0, %4\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r)
                 : "v"(scr_w), "v"(a), "v"(b), "v"(scr_r)
                 : "memory");
    double v = __hiloint2double((int)r.y, (int)r.x) + __hiloint2double((int)r.w, (int)r.z);
    // a butterfly over the 32 lanes of each half: after every step the lanes of a group hold the same
    // subtotal, so the tree is the one of the DPP sequence this replaced (quad_perm, quad_perm,
    // row_half_mirror, row_mirror, row_bcast:15) and the totals are the same bits
    v = swz_add<1>(v);
    v = swz_add<2>(v);
    v = swz_add<4>(v);
    v = swz_add<8>(v);
    v = swz_add<16>(v);
    return v;
}

// The same with the five exchange steps as DPP moves (two v_mov_dpp and the add per step: 15 vector instructions instead
// of 5, but no trip through the LDS crossbar and nothing to wait for): the additions and their order are those of the
// swizzle form -- after every step the lanes of a group hold the same subtotal, so a mirror within the group's double is the
// exchange with lane ^ X -- and the totals are the same bits.  For the matrix-core form, whose wave time is LDS round trips.
// Lanes 31 and 63 hold the totals of the first and second sum (row_bcast:15 fills rows 1 and 3 only).
__device__ __forceinline__ double wave_sum2_dpp(double a, double b, uint32_t scr_w, uint32_t scr_r)
{
    uint4 r;
    asm volatile("ds_write_b64 %1, %2\n\t"
                 "ds_write_b64 %1, %3 offset:512\n\t"
                 "ds_read_b128 %0, %4\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(r)
                 : "v"(scr_w), "v"(a), "v"(b), "v"(scr_r)
                 : "memory");
    double v = __hiloint2double((int)r.y, (int)r.x) + __hiloint2double((int)r.w, (int)r.z);
    v = dpp_add<0xB1, 0xf>(v);      // quad_perm [1,0,3,2]      lane ^ 1
    v = dpp_add<0x4E, 0xf>(v);      // quad_perm [2,3,0,1]      lane ^ 2
    v = dpp_add<0x141, 0xf>(v);     // row_half_mirror          the other quad of the eight
    v = dpp_add<0x140, 0xf>(v);     // row_mirror               the other eight of the row
    v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1 and 3: the other row of the half
    return v;
}

// Staging of the window constants: the table base each word indexes is added on the way into LDS
// (i = index of the uint4 within the run's constants, two per window).
__device__ __forceinline__ uint4 stage_wc_base(uint4 v, uint32_t i, uint32_t tab1, uint32_t tab2)
{
    if (i & 1) {
        v.x += tab1;
        v.y += tab1;
        v.z = tab2;
    } else {
        v.y += tab1;
        v.z += tab2;
        v.w += tab2;
    }
    return v;
}

// ---------------------------------------------------------------------------
// The --LD loop.
//
// Work split: a workgroup = 8 waves = 8 chunks of 64 background individuals (one individual
// per lane) x one run of consecutive windows (run_begin[]: up to `win_per_group` windows, fewer
// towards the end of the grid -- the host's guided run lengths; workgroups of a run are adjacent
// in blockIdx order).  Each wave streams its chunk's tile pairs exactly once.
//
// Two kernels share this design: k_ld_popcount (one comparison individual per workgroup) and
// k_ld_popcount_mt (four).  Both perform the same operations in the same order per result.
// Their loops are written so that a window's first segment STARTS the counters (IBDG_SEGMENT(=))
// and the others add to them (IBDG_SEGMENT(+=)): nothing is reset between windows.
//
// Data movement (all of it asynchronous to the arithmetic):
//   * once per workgroup the run's segment records (+ the target's haplotype words per
//     segment), the per-window constants and the two power tables are copied to LDS;
//   * the wave's tile pairs go HBM -> LDS by direct-to-LDS loads (global_load_lds_dwordx4,
//     1 KiB per instruction, no VGPRs) into a private ring of NS slots, NS-1 loads in flight
//     while a pair is consumed.  Landing is tracked with counted s_waitcnt vmcnt
//     (vector-memory ops of a wave complete in issue order).
//   * per segment one asm statement reads the hot half of the record with two broadcast
//     ds_read_b128 (masks and the target's words land in VGPRs: VALU instructions with VGPR
//     operands only issue at twice the rate of those with an SGPR operand) and the lane's own
//     two tile words with one ds_read_b64.
//   All LDS reads in the loop are asm: hipcc drains vmcnt(0) before any LDS read that follows
//   a direct-to-LDS load, which would serialise the ring (and so would table look-ups from
//   global memory at the end of every window -- hence the tables in LDS).
// Arithmetic per segment: per weight bit-plane 14 VALU ops for the seven cov-weighted counts
// and 4 for the two alt-weighted ones; three cov planes and two alt planes unconditionally,
// higher planes (rare) under one uniform test.
// At the end of each window the lane turns its counts into the five products, the wave sums
// count[n]*product over its 64 individuals (DPP moves, fixed order) and lane 63 stores the
// per-chunk partial; k_ld_finalize adds the chunks (one wave per window, same fixed order),
// applies the mantissa of K' and divides.
// ---------------------------------------------------------------------------


// Issue and wait in ONE statement: an asm output must be final when the statement ends,
// because hipcc is free to copy it to another register right afterwards (it did, with the
// wait in a later statement: the copy read the register before the LDS data had landed).
// The LDS latency is covered by the other resident waves of the SIMD instead.
__device__ __forceinline__ void lds_fetch(uint4 &h0, uint4 &h1, uint2 &x, uint32_t rec_addr, uint32_t x_addr)
{
    asm volatile("ds_read_b128 %0, %3\n\t"
                 "ds_read_b128 %1, %3 offset:16\n\t"
                 "ds_read_b64 %2, %4\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(h0), "=&v"(h1), "=&v"(x)
                 : "v"(rec_addr), "v"(x_addr)
                 : "memory");
}

// ten power-table entries (16 B each) for the five products of one window
__device__ __forceinline__ void lds_read_pow10(uint4 (&p)[10], const uint32_t (&ad)[10])
{
    asm volatile("ds_read_b128 %0, %10\n\t"
                 "ds_read_b128 %1, %11\n\t"
                 "ds_read_b128 %2, %12\n\t"
                 "ds_read_b128 %3, %13\n\t"
                 "ds_read_b128 %4, %14\n\t"
                 "ds_read_b128 %5, %15\n\t"
                 "ds_read_b128 %6, %16\n\t"
                 "ds_read_b128 %7, %17\n\t"
                 "ds_read_b128 %8, %18\n\t"
                 "ds_read_b128 %9, %19\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]),
                   "=&v"(p[7]), "=&v"(p[8]), "=&v"(p[9])
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]),
                   "v"(ad[8]), "v"(ad[9])
                 : "memory");
}

// ten 8-byte power-table entries (the matrix-core form's tables of plain doubles)
__device__ __forceinline__ void lds_read_pow10_b64(uint2 (&p)[10], const uint32_t (&ad)[10])
{
    asm volatile("ds_read_b64 %0, %10\n\t"
                 "ds_read_b64 %1, %11\n\t"
                 "ds_read_b64 %2, %12\n\t"
                 "ds_read_b64 %3, %13\n\t"
                 "ds_read_b64 %4, %14\n\t"
                 "ds_read_b64 %5, %15\n\t"
                 "ds_read_b64 %6, %16\n\t"
                 "ds_read_b64 %7, %17\n\t"
                 "ds_read_b64 %8, %18\n\t"
                 "ds_read_b64 %9, %19\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]),
                   "=&v"(p[7]), "=&v"(p[8]), "=&v"(p[9])
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7]),
                   "v"(ad[8]), "v"(ad[9])
                 : "memory");
}

// The counting itself is written in assembly, one statement per group of (mask, count) pairs, for the sake of ONE
// scalar instruction inside every pair:
//     v_and_b32 t, x, m ; s_nop 0 ; v_bcnt_u32_b32 c, t, c
// Back to back, a 1:1 stream of v_and_b32 (2.2 cycles per wave alone) and v_bcnt_u32_b32 (4.2) issues at 3.8-4.0
// cycles per instruction; with one scalar instruction per pair -- s_nop 0, any SALU instruction, before or after
// the count -- it issues at 3.2, the average of its parts; two per pair, one per four vector instructions or
// s_nop 1 lose it again (tools/ubench/nop_mix.hip, profiles/r02_nop_mix.txt).  hipcc knows nothing of this and
// moves scalar work of the loop into the stream wherever it fits, so the pairs are fenced by scheduling barriers
// (IBDG_SEGMENT) and everything scalar is computed before them.
// A0..: what the count starts from -- "0" for the first segment of a window (the counters start there, nothing is
// zeroed between windows), the counter itself afterwards.
// One statement per segment (per weight plane in the kernel for several individuals): between two asm statements
// hipcc's hazard recogniser puts an s_nop of its own, a second scalar instruction for that pair.
#define IBDG_PAIR_SET(tmp, a, b, cnt) \
    "v_and_b32 %[" #tmp "], %[" #a "], %[" #b "]\n\ts_nop 0\n\tv_bcnt_u32_b32 %[" #cnt "], %[" #tmp "], 0\n\t"
#define IBDG_PAIR_ADD(tmp, a, b, cnt) \
    "v_and_b32 %[" #tmp "], %[" #a "], %[" #b "]\n\ts_nop 0\n\tv_bcnt_u32_b32 %[" #cnt "], %[" #tmp "], %[" #cnt "]\n\t"
// C(x0) C(x1) C(x0&x1) and the four G(x,t) of weight plane k (u0 = x0 & cov, u1 = x1 & cov are shared)
#define IBDG_COV_PLANE_TEXT(P, k) \
    P(u0, x0, cov##k, c0##k) P(u1, x1, cov##k, c1##k) P(t, hom, cov##k, ch##k) \
    P(t, u0, at0, g00##k) P(t, u1, at0, g01##k) P(t, u0, at1, g10##k) P(t, u1, at1, g11##k)
#define IBDG_ALT_TEXT(P) P(t, x0, alt0, a00) P(t, x1, alt0, a10) P(t, x0, alt1, a01) P(t, x1, alt1, a11)
#define IBDG_SEG_TEXT(P) IBDG_COV_PLANE_TEXT(P, 0) IBDG_COV_PLANE_TEXT(P, 1) IBDG_COV_PLANE_TEXT(P, 2) IBDG_ALT_TEXT(P)
#define IBDG_CNT3(C, name, arr) [name##0] C(arr[0]), [name##1] C(arr[1]), [name##2] C(arr[2])

// the 25 counts of one segment for one comparison individual
template <bool FIRST>
__device__ __forceinline__ void count_segment(uint32_t (&c0)[3], uint32_t (&c1)[3], uint32_t (&ch)[3], uint32_t (&g00)[3],
                                              uint32_t (&g01)[3], uint32_t (&g10)[3], uint32_t (&g11)[3], uint32_t (&A0)[2],
                                              uint32_t (&A1)[2], uint32_t x0, uint32_t x1, uint32_t hom, uint32_t cov0,
                                              uint32_t cov1, uint32_t cov2, uint32_t alt0, uint32_t alt1, uint32_t at0, uint32_t at1)
{
    uint32_t u0, u1, t;
#define IBDG_OUT_SET(v) "=&v"(v)
#define IBDG_OUT_ADD(v) "+v"(v)
#define IBDG_SEG_OPERANDS(C)                                                                                              \
    : IBDG_CNT3(C, c0, c0), IBDG_CNT3(C, c1, c1), IBDG_CNT3(C, ch, ch), IBDG_CNT3(C, g00, g00), IBDG_CNT3(C, g01, g01),   \
      IBDG_CNT3(C, g10, g10), IBDG_CNT3(C, g11, g11), [a00] C(A0[0]), [a01] C(A0[1]), [a10] C(A1[0]), [a11] C(A1[1]),     \
      [u0] "=&v"(u0), [u1] "=&v"(u1), [t] "=&v"(t)                                                                        \
    : [x0] "v"(x0), [x1] "v"(x1), [hom] "v"(hom), [cov0] "v"(cov0), [cov1] "v"(cov1), [cov2] "v"(cov2), [alt0] "v"(alt0), \
      [alt1] "v"(alt1), [at0] "v"(at0), [at1] "v"(at1)
    if (FIRST)
        asm volatile(IBDG_SEG_TEXT(IBDG_PAIR_SET) IBDG_SEG_OPERANDS(IBDG_OUT_SET));
    else
        asm volatile(IBDG_SEG_TEXT(IBDG_PAIR_ADD) IBDG_SEG_OPERANDS(IBDG_OUT_ADD));
#undef IBDG_SEG_OPERANDS
}

// the same for four comparison individuals, one weight plane per statement: the common three counts and 4 x 4 G(x,t)
#define IBDG_G4_TEXT(P, j) P(t, u0, ta##j, g##j##0) P(t, u1, ta##j, g##j##1) P(t, u0, tb##j, g##j##2) P(t, u1, tb##j, g##j##3)
#define IBDG_PLANE4_TEXT(P) P(u0, x0, cov, c0) P(u1, x1, cov, c1) P(t, hom, cov, ch) IBDG_G4_TEXT(P, 0) IBDG_G4_TEXT(P, 1) IBDG_G4_TEXT(P, 2) IBDG_G4_TEXT(P, 3)
#define IBDG_G4_OPS(C, j) [g##j##0] C(gq[j][0][k]), [g##j##1] C(gq[j][1][k]), [g##j##2] C(gq[j][2][k]), [g##j##3] C(gq[j][3][k])
template <bool FIRST>
__device__ __forceinline__ void count_plane_mt(uint32_t &c0, uint32_t &c1, uint32_t &ch, uint32_t (&gq)[4][4][3], int k,
                                               uint32_t x0, uint32_t x1, uint32_t hom, uint32_t cov, const uint32_t (&tw)[4][2])
{
    uint32_t u0, u1, t;
#define IBDG_P4_OPERANDS(C)                                                                                               \
    : [c0] C(c0), [c1] C(c1), [ch] C(ch), IBDG_G4_OPS(C, 0), IBDG_G4_OPS(C, 1), IBDG_G4_OPS(C, 2), IBDG_G4_OPS(C, 3),     \
      [u0] "=&v"(u0), [u1] "=&v"(u1), [t] "=&v"(t)                                                                        \
    : [x0] "v"(x0), [x1] "v"(x1), [hom] "v"(hom), [cov] "v"(cov), [ta0] "v"(tw[0][0]), [tb0] "v"(tw[0][1]),               \
      [ta1] "v"(tw[1][0]), [tb1] "v"(tw[1][1]), [ta2] "v"(tw[2][0]), [tb2] "v"(tw[2][1]), [ta3] "v"(tw[3][0]), [tb3] "v"(tw[3][1])
    if (FIRST)
        asm volatile(IBDG_PLANE4_TEXT(IBDG_PAIR_SET) IBDG_P4_OPERANDS(IBDG_OUT_SET));
    else
        asm volatile(IBDG_PLANE4_TEXT(IBDG_PAIR_ADD) IBDG_P4_OPERANDS(IBDG_OUT_ADD));
#undef IBDG_P4_OPERANDS
}

template <bool FIRST>
__device__ __forceinline__ void count_alt(uint32_t (&A0)[2], uint32_t (&A1)[2], uint32_t x0, uint32_t x1, uint32_t alt0, uint32_t alt1)
{
    uint32_t t;
    if (FIRST)
        asm volatile(IBDG_ALT_TEXT(IBDG_PAIR_SET)
                     : [a00] "=&v"(A0[0]), [a01] "=&v"(A0[1]), [a10] "=&v"(A1[0]), [a11] "=&v"(A1[1]), [t] "=&v"(t)
                     : [x0] "v"(x0), [x1] "v"(x1), [alt0] "v"(alt0), [alt1] "v"(alt1));
    else
        asm volatile(IBDG_ALT_TEXT(IBDG_PAIR_ADD)
                     : [a00] "+v"(A0[0]), [a01] "+v"(A0[1]), [a10] "+v"(A1[0]), [a11] "+v"(A1[1]), [t] "=&v"(t)
                     : [x0] "v"(x0), [x1] "v"(x1), [alt0] "v"(alt0), [alt1] "v"(alt1));
}

// One segment: fetch its record and tile words, advance the ring, count.  FIRST: the first segment of a window
// (the counters start there, so nothing has to be zeroed).
#define IBDG_SEGMENT(FIRST)                                                                                        \
    {                                                                                                           \
        uint4 h0, h1;                                                                                           \
        uint2 x;                                                                                                \
        lds_fetch(h0, h1, x, rec_addr, ring_lane + x_off);                                                      \
        flags = __builtin_amdgcn_readfirstlane(h0.x);                                                           \
        const uint32_t adv = (flags >> 4) & 0xff;                                                               \
        if (adv) {                                                                                              \
            for (uint32_t i = 0; i < adv; ++i, ++q_issue)                                                       \
                if (q_issue <= q_last)                                                                          \
                    __builtin_amdgcn_global_load_lds((const void *)(xt + (size_t)q_issue * 64),                 \
                                                     (lds_void *)(ring + ((q_issue - q0) % NS) * 1024), 16, 0, IBDG_TILE_AUX); \
            if (q_issue - 1 <= q_last)                                                                          \
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS - 1) : "memory");                                   \
            else                                                                                                \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
        }                                                                                                       \
        x_off = (flags & 7) * 1024 + ((flags >> 3) & 1) * 8;                                                    \
        const uint32_t cov0 = h0.y, cov1 = h0.z, cov2 = h0.w, alt0 = h1.x, alt1 = h1.y;                         \
        const uint2 at = make_uint2(h1.z, h1.w);                                                                \
        const uint32_t hom = x.x & x.y;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        count_segment<FIRST>(c0, c1, ch, g00, g01, g10, g11, A0, A1, x.x, x.y, hom, cov0, cov1, cov2, alt0, alt1, at.x, at.y); \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if (flags & (1u << 12)) {                                                                               \
            const uint32_t ncov = (flags >> 16) & 0xff, nalt = flags >> 24;                                     \
            for (uint32_t k = FC; k < ncov; ++k) {                                                              \
                const uint32_t cov = segs[seg0 + s].cov[k];              /* uniform: scalar load */             \
                const uint32_t u0 = x.x & cov, u1 = x.y & cov;                                                  \
                c0[0] += (uint32_t)__popc(u0) << k;                                                             \
                c1[0] += (uint32_t)__popc(u1) << k;                                                             \
                ch[0] += (uint32_t)__popc(hom & cov) << k;                                                      \
                g00[0] += (uint32_t)__popc(u0 & at.x) << k;                                                     \
                g01[0] += (uint32_t)__popc(u1 & at.x) << k;                                                     \
                g10[0] += (uint32_t)__popc(u0 & at.y) << k;                                                     \
                g11[0] += (uint32_t)__popc(u1 & at.y) << k;                                                     \
            }                                                                                                   \
            for (uint32_t k = FA; k < nalt; ++k) {                                                              \
                const uint32_t alt = segs[seg0 + s].alt[k];                                                     \
                A0[0] += (uint32_t)__popc(x.x & alt) << k;                                                      \
                A1[0] += (uint32_t)__popc(x.y & alt) << k;                                                      \
            }                                                                                                   \
        }                                                                                                       \
        rec_addr += IBDG_REC_WORDS * 4;                                                                         \
        ++s;                                                                                                    \
    }

// The same with the counts of the two haplotype words on the matrix cores (acc0 / acc1: <x,cov> <x,alt> <x&t0,cov>
// <x&t1,cov> of x0 / x1 as exact integers in f32; a window's first segment starts them from zero) and the three
// planes of <x0&x1,cov> as (mask, count) pairs.
#define IBDG_SEGMENT_MX(FIRST)                                                                                     \
    {                                                                                                           \
        uint4 h0;                                                                                               \
        uint2 x;                                                                                                \
        lds_fetch_mx(h0, x, af_lo, af_hi, rec_addr, ring_lane + x_off, frag_addr);                              \
        flags = __builtin_amdgcn_readfirstlane(h0.x);                                                           \
        const uint32_t adv = flags >> 16;                                                                       \
        if (adv) {                                                                                              \
            /* every advance requests a pair -- past the run's last one that pair again, into a slot nobody reads any  \
               more -- so the count of loads in flight stays the nominal one and one counted wait serves */     \
            for (uint32_t i = 0; i < adv; ++i, ++q_issue)                                                       \
                __builtin_amdgcn_global_load_lds((const void *)(xt + (size_t)(q_issue < q_last ? q_issue : q_last) * 64), \
                                                 (lds_void *)(ring + ((q_issue - q0) % NS) * 1024), 16, 0, IBDG_TILE_AUX);  \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS - 1) : "memory");                                       \
        }                                                                                                       \
        x_off = flags & 0x3fff;                                                                                 \
        const uint32_t hom = x.x & x.y;                                                                         \
        {                                                                                                       \
            const mx_v8i av = {(int)af_lo.x, (int)af_lo.y, (int)af_lo.z, (int)af_lo.w, (int)af_hi.x, (int)af_hi.y, 0, 0}; \
            const mx_v4f zero = {0.f, 0.f, 0.f, 0.f};                                                           \
            acc0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bits_to_fp4(x.x), FIRST ? zero : acc0, 2, 4, 0, \
                                                                    0x7f80, 1, 0x7f80);                         \
            acc1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bits_to_fp4(x.y), FIRST ? zero : acc1, 2, 4, 0, \
                                                                    0x7f80, 1, 0x7f80);                         \
        }                                                                                                       \
        if (FIRST) {                                                                                            \
            ch[0] = __popc(hom & h0.y); ch[1] = __popc(hom & h0.z); ch[2] = __popc(hom & h0.w);                 \
        } else {                                                                                                \
            ch[0] += __popc(hom & h0.y); ch[1] += __popc(hom & h0.z); ch[2] += __popc(hom & h0.w);              \
        }                                                                                                       \
        if (flags & (1u << 14)) {                                                                               \
            const uint4 h1 = lds_read_b128(rec_addr + 16);                                                      \
            const uint2 at = make_uint2(h1.x, h1.y);                                                            \
            const uint32_t nn = __builtin_amdgcn_readfirstlane(h1.z), ncov = nn & 0xff, nalt = nn >> 8;         \
            for (uint32_t k = 3; k < ncov; ++k) {                                                               \
                const uint32_t cov = segs[seg0 + s].cov[k];              /* uniform: scalar load */             \
                const uint32_t u0 = x.x & cov, u1 = x.y & cov;                                                  \
                acc0[0] += (float)((uint32_t)__popc(u0) << k);                                                  \
                acc1[0] += (float)((uint32_t)__popc(u1) << k);                                                  \
                ch[0] += (uint32_t)__popc(hom & cov) << k;                                                      \
                acc0[2] += (float)((uint32_t)__popc(u0 & at.x) << k);                                           \
                acc1[2] += (float)((uint32_t)__popc(u1 & at.x) << k);                                           \
                acc0[3] += (float)((uint32_t)__popc(u0 & at.y) << k);                                           \
                acc1[3] += (float)((uint32_t)__popc(u1 & at.y) << k);                                           \
            }                                                                                                   \
            for (uint32_t k = 3; k < nalt; ++k) {                                                               \
                const uint32_t alt = segs[seg0 + s].alt[k];                                                     \
                acc0[1] += (float)((uint32_t)__popc(x.x & alt) << k);                                           \
                acc1[1] += (float)((uint32_t)__popc(x.y & alt) << k);                                           \
            }                                                                                                   \
        }                                                                                                       \
        rec_addr += IBDG_RECX_WORDS * 4;                                                                        \
        frag_addr += IBDG_RECX_WORDS * 4;                                                                       \
        ++s;                                                                                                    \
    }

// The IBD1 form (PopArgs::ibd1): no counts for the individual's own genotype factors -- their products come from the one pass
// over the site list --, and the four sums of a word are the table exponents themselves (k_win_target_mx); a window's first
// segment starts the accumulators from 1.5 * 2^23, so that their BITS hold the (signed) sums.
#define IBDG_SEGMENT_X1(FIRST)                                                                                     \
    {                                                                                                           \
        uint32_t ctl;                                                                                           \
        uint2 x;                                                                                                \
        lds_fetch_x1(ctl, x, af_lo, af_hi, ring_lane + x_off, frag_base);                                       \
        flags = __builtin_amdgcn_readfirstlane(ctl);                                                            \
        const uint32_t adv = flags >> 16;                                                                       \
        if (adv) {                                                                                              \
            for (uint32_t i = 0; i < adv; ++i, ++q_issue)                                                       \
                __builtin_amdgcn_global_load_lds((const void *)(xt + (size_t)(q_issue < q_last ? q_issue : q_last) * 64), \
                                                 (lds_void *)(ring + ((q_issue - q0) % NS) * 1024), 16, 0, IBDG_TILE_AUX);  \
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS - 1) : "memory");                                       \
        }                                                                                                       \
        x_off = flags & 0x3fff;                                                                                 \
        {                                                                                                       \
            const mx_v8i av = {(int)af_lo.x, (int)af_lo.y, (int)af_lo.z, (int)af_lo.w, (int)af_hi.x, (int)af_hi.y, 0, 0}; \
            acc0 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bits_to_fp4(x.x), FIRST ? bias4 : acc0, 2, 4, 0, \
                                                                    0x7f80, 1, 0x7f80);                         \
            acc1 = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bits_to_fp4(x.y), FIRST ? bias4 : acc1, 2, 4, 0, \
                                                                    0x7f80, 1, 0x7f80);                         \
        }                                                                                                       \
        if (flags & (1u << 14)) {                                                                               \
            const uint4 h1 = lds_read_b128((uint32_t)__builtin_amdgcn_readfirstlane((int)frag_base) + 16);      \
            const uint2 at = make_uint2(h1.x, h1.y);                                                            \
            const uint32_t nn = __builtin_amdgcn_readfirstlane(h1.z), ncov = nn & 0xff, nalt = nn >> 8;         \
            for (uint32_t k = 3; k < ncov; ++k) {                                                               \
                const uint32_t cov = segs[seg0 + s].cov[k];              /* uniform: scalar load */             \
                const uint32_t u0 = x.x & cov, u1 = x.y & cov;                                                  \
                const int c0 = __popc(u0), c1 = __popc(u1), m = 1 << k;                                         \
                const int g00 = __popc(u0 & at.x), g01 = __popc(u1 & at.x);                                     \
                const int g10 = __popc(u0 & at.y), g11 = __popc(u1 & at.y);                                     \
                acc0[0] += (float)((c0 - 2 * g00) * m);                                                         \
                acc0[1] += (float)((c0 - 2 * g10) * m);                                                         \
                acc0[2] += (float)(g00 * m);                                                                    \
                acc0[3] += (float)(g10 * m);                                                                    \
                acc1[0] += (float)((c1 - 2 * g01) * m);                                                         \
                acc1[1] += (float)((c1 - 2 * g11) * m);                                                         \
                acc1[2] += (float)(g01 * m);                                                                    \
                acc1[3] += (float)(g11 * m);                                                                    \
            }                                                                                                   \
            for (uint32_t k = 3; k < nalt; ++k) {                                                               \
                const uint32_t alt = segs[seg0 + s].alt[k];                                                     \
                const float a0 = (float)((uint32_t)__popc(x.x & alt) << k), a1 = (float)((uint32_t)__popc(x.y & alt) << k); \
                acc0[2] -= a0;                                                                                  \
                acc0[3] -= a0;                                                                                  \
                acc1[2] -= a1;                                                                                  \
                acc1[3] -= a1;                                                                                  \
            }                                                                                                   \
        }                                                                                                       \
        frag_base += IBDG_RECX_WORDS * 4;                                                                       \
        ++s;                                                                                                    \
    }

// eight 8-byte power-table entries (the four IBD1 products of the IBD1 form)
__device__ __forceinline__ void lds_read_pow8_b64(uint2 (&p)[8], const uint32_t (&ad)[8])
{
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
                 : "v"(ad[0]), "v"(ad[1]), "v"(ad[2]), "v"(ad[3]), "v"(ad[4]), "v"(ad[5]), "v"(ad[6]), "v"(ad[7])
                 : "memory");
}

template <int NS, bool TAB_LDS, bool MX, bool IBD1 = false>
__global__ __launch_bounds__(512) void k_ld_popcount(const uint4 *__restrict__ t32,
                                                     const Seg *__restrict__ segs,
                                                     const uint32_t *__restrict__ rec_ready,
                                                     const WinConst *__restrict__ wconst,
                                                     const uint32_t *__restrict__ wc_ready,
                                                     const uint4 *__restrict__ pow_1me,
                                                     const uint4 *__restrict__ pow_eps,
                                                     const uint32_t *__restrict__ run_begin,
                                                     PopArgs a)
{
    constexpr int FC = 3, FA = 2;      // weight bit-planes with counters of their own (cov, alt)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    const unsigned t = blockIdx.z;
    // workgroups of one run are neighbours in blockIdx order (and so in dispatch order): the
    // runs at the end of the grid are the short ones (host: guided run lengths)
    // The first workgroups of the grid (a.fin_prev != null: ceil(windows / waves of a workgroup) of them) do the finalising step of the PREVIOUS
    // run of the same shape -- k_ld_finalize's arithmetic, a wave per window -- whose partial sums that launch left in the
    // other half of their buffer: complete and visible, a kernel boundary lies between.  The workgroups of the runs follow.
    uint32_t bx = blockIdx.x;
    if (a.fin_prev) {
        const uint32_t n_fin = (a.n_win + a.waves_per_group - 1) / a.waves_per_group;
        if (bx < n_fin) {
            const uint32_t w = bx * a.waves_per_group + wave;
            if (w < a.n_win) {
                const unsigned tt = a.t_base + t;
                const double2 *p = reinterpret_cast<const double2 *>(a.fin_prev) + ((size_t)tt * a.n_win + w) * a.n_chunks;
                double t0 = 0.0, t1 = 0.0;
                for (uint32_t cc = lane; cc < a.n_chunks; cc += 64) {
                    const double2 v = p[cc];
                    t0 += v.x;
                    t1 += v.y;
                }
                uint32_t tgt_own = a.fin_p2c ? a.fin_targets[tt] : 0u;
                IBDG_CHECK_TGT(tgt_own, a.fin_p2c ? a.lanes : 1u, "fused finalize");
                t0 = a.fin_p2c ? ibd0_from_pass(a.fin_p2c, a.fin_p2w, a.lanes, a.n_chunks, w, tgt_own, lane)
                               : wave_sum_to_lane63(t0);
                t1 = wave_sum_to_lane63(t1);
                if (lane == 63) {
                    const int nref = a.n_refpanel[tt];
                    const double mK = wconst[w].mK;             // mantissa of K' (its exponent went into every term)
                    double *o = a.win_ll + ((size_t)tt * a.n_win + w) * 3;
                    o[0] = (t0 * mK) / (double)nref;
                    o[1] = (t1 * mK) / (double)(nref * 4);
                }
            }
            return;
        }
        bx -= n_fin;
    }
    const uint32_t run = bx / a.n_cgroups, cgroup = bx - run * a.n_cgroups;
    const uint32_t w0 = run_begin[run], w1 = run_begin[run + 1];
    const uint32_t seg0 = wconst[w0].seg_begin, seg1 = wconst[w1].seg_begin;
    const uint32_t nseg = seg1 - seg0;
    if (nseg == 0)
        return;

    // ---- LDS carve-up (see ld_popcount_lds_bytes)
    constexpr uint32_t RECW = MX ? IBDG_RECX_WORDS : IBDG_REC_WORDS;
    uint32_t *rec_lds = reinterpret_cast<uint32_t *>(smem);                       // [max_seg][RECW]
    uint32_t *wc_lds = rec_lds + (size_t)a.max_seg * RECW;                         // [win_per_group][8]
    uint4 *tab_lds = reinterpret_cast<uint4 *>(
        smem + ((((size_t)a.max_seg * RECW + (size_t)a.win_per_group * IBDG_WC_WORDS) * 4 + 15) & ~(size_t)15));
    // (matrix-core form with the tables in LDS: plain doubles, 8 bytes per entry -- see the window end)
    constexpr bool PLAIN = MX && TAB_LDS;
    const size_t tab_bytes = TAB_LDS ? (size_t)a.tab_len * (PLAIN ? 16 : 32) : 0;
    char *ring0 = smem + ((((size_t)a.max_seg * RECW + (size_t)a.win_per_group * IBDG_WC_WORDS) * 4 + 15 + tab_bytes + 1023) & ~(size_t)1023);

    // ---- prime the ring FIRST: pairs q0 .. q0+NS-1 (not past the run's last pair).  The
    // direct-to-LDS loads fly while the workgroup stages its records and tables below, so the
    // two start-up latencies of a workgroup overlap instead of adding up.
    const unsigned c = cgroup * a.waves_per_group + wave;
    const bool has_chunk = c < a.n_chunks;
    char *ring = ring0 + (size_t)wave * NS * 1024;
    const uint4 *xt = t32 + (size_t)c * a.n_pairs * 64 + lane;      // + pair*64
    const uint32_t tile0 = segs[seg0].tile;
    const uint32_t q0 = tile0 >> 1, q_last = segs[seg1 - 1].tile >> 1;
    IBDG_CHECK_IDX(q_last, a.n_pairs, "k_ld_popcount last pair");
    IBDG_CHECK_IDX(q0, q_last + 1, "k_ld_popcount first pair");
    uint32_t q_issue = q0;                           // next pair to request (nominal: runs past q_last)
    if (has_chunk) {
#pragma unroll
        for (int i = 0; i < NS; ++i, ++q_issue)
            if (MX || q_issue <= q_last)              // (matrix-core form: always, see its segment)
                __builtin_amdgcn_global_load_lds((const void *)(xt + (size_t)(q_issue < q_last ? q_issue : q_last) * 64),
                                                 (lds_void *)(ring + ((q_issue - q0) % NS) * 1024), 16, 0, IBDG_TILE_AUX);
    }

    // ---- stage the run's records, window constants and tables (whole workgroup)
    {
        // plain contiguous copies (k_win_target prepared the LDS images): every load of a thread is
        // independent of the others, so the whole staging costs about one memory latency
        const uint4 *rsrc = reinterpret_cast<const uint4 *>(rec_ready) + ((size_t)t * a.n_segs + seg0) * (RECW / 4);
        uint4 *rdst = reinterpret_cast<uint4 *>(rec_lds);
        for (uint32_t i = threadIdx.x; i < nseg * (RECW / 4); i += blockDim.x)
            rdst[i] = rsrc[i];
        // with the tables in LDS the constants become LDS addresses (table base + 16 * exponent), otherwise
        // they stay byte offsets into the global tables
        const uint32_t stab1 = TAB_LDS ? (uint32_t)(uintptr_t)(lds_void *)tab_lds : 0u;
        const uint32_t stab2 = TAB_LDS ? stab1 + a.tab_len * (PLAIN ? 8 : 16) : 0u;
        const uint4 *wsrc = reinterpret_cast<const uint4 *>(wc_ready) + ((size_t)t * a.n_win + w0) * (IBDG_WC_WORDS / 4);
        uint4 *wdst = reinterpret_cast<uint4 *>(wc_lds);
        for (uint32_t i = threadIdx.x; i < (w1 - w0) * (IBDG_WC_WORDS / 4); i += blockDim.x)
            wdst[i] = stage_wc_base(wsrc[i], i, stab1, stab2);
        if (PLAIN) {
            // rho^n * 2^(s n) and sigma^n as plain doubles: the mantissas of the {mantissa, exponent} tables, exactly (a power
            // of two moves no bit); s = a.rho_shift keeps rho^n inside the double range for every n of the table
            // (the host checks), sigma = 1 / (2 (1 - eps)) > 1/2 needs none
            double *td = reinterpret_cast<double *>(tab_lds);
            const PowEntry *p1 = reinterpret_cast<const PowEntry *>(pow_1me), *p2 = reinterpret_cast<const PowEntry *>(pow_eps);
            for (uint32_t i = threadIdx.x; i < 2 * a.tab_len; i += blockDim.x) {
                const PowEntry e = i < a.tab_len ? p1[i] : p2[i - a.tab_len];
                td[i] = __builtin_ldexp(e.m, e.e + (i < a.tab_len ? (int)(a.rho_shift * i) : 0));
            }
        } else if (TAB_LDS) {
            for (uint32_t i = threadIdx.x; i < 2 * a.tab_len; i += blockDim.x)
                tab_lds[i] = i < a.tab_len ? pow_1me[i] : pow_eps[i - a.tab_len];
        }
    }
    __syncthreads();

    if (!has_chunk)
        return;
    const uint32_t ring_lane = (uint32_t)(uintptr_t)(lds_void *)ring + lane * 16;
    // the wave's 1 KiB scratch for wave_sum2, behind the rings
    const uint32_t scr = (uint32_t)(uintptr_t)(lds_void *)(ring0 + (size_t)a.waves_per_group * NS * 1024 + wave * 1024);
    const uint32_t scr_w = scr + lane * 8, scr_r = scr + lane * 16;

    const double wgt = a.weight[(size_t)(a.t_base + t) * a.lanes + c * 64 + lane];

    // Counters per weight bit-plane: three planes for the cov-weighted sums, two for the
    // alt-weighted ones are kept apart (one v_bcnt_u32_b32 accumulates into them directly);
    // the rare higher planes are shifted into plane 0 as they are counted.
    uint32_t c0[FC], c1[FC], ch[FC], g00[FC], g01[FC], g10[FC], g11[FC], A0[FA], A1[FA];

    // the first pair must have landed (it was requested before the staging loads, so it has)
    if (MX || q_issue - 1 <= q_last)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS - 1) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // Where a segment's tile words sit in the ring and how far the ring must advance before the
    // NEXT segment are precomputed by the host into each record's flag word (ring slot relative to
    // the run's first pair), so the loop carries no tile/pair arithmetic:
    //   flags = next slot (3) | next half (1) | pairs to advance (8) | rare planes (1) | last (1) | .. | ncov (8) | nalt (8)
    uint32_t m32 = (uint32_t)-32;                    // multiplier of the 2 G(x,t) term in a table address, kept in a VGPR
    asm volatile("" : "+v"(m32));
    uint32_t x_off = (tile0 & 1) * 8;                // ring byte offset of the current segment's words (slot 0)
    uint32_t rec_addr = (uint32_t)(uintptr_t)(lds_void *)rec_lds;     // same value in every lane (VGPR)
    const uint32_t wc_base = (uint32_t)(uintptr_t)(lds_void *)wc_lds;
    uint32_t s = 0;
    // (matrix-core form) the two accumulators, the lane's A fragment -- zero except in the lanes 20 kb + sum, which read
    // the 24 bytes of `sum` behind the record's header
    mx_v4f acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    mx_u4 af_lo = {0, 0, 0, 0};
    mx_u2 af_hi = {0, 0};
    uint32_t frag_addr = rec_addr + 32 + 24 * (lane & 3);
    if constexpr (IBD1) {
        static_assert(MX && TAB_LDS, "the IBD1 form exists with the counts on the matrix cores and the tables in LDS");
        // 1.5 * 2^23 in four registers that stay: the start value of a window's accumulators (an inline constant it is not)
        mx_v4f bias4 = {12582912.f, 12582912.f, 12582912.f, 12582912.f};
        asm volatile("" : "+v"(bias4));
        const uint32_t tab1 = (uint32_t)(uintptr_t)(lds_void *)tab_lds;
        uint32_t frag_base = rec_addr + 24 * (lane & 3);
        // one window: its segments, then the four IBD1 products of the lane's individual (:716-719, :744-745) times its multiplicity
        auto window_sum = [&](uint32_t w) __attribute__((always_inline)) -> double {
            uint32_t flags;
            IBDG_SEGMENT_X1(true)
            while (!(flags & (1u << 15)))
                IBDG_SEGMENT_X1(false)
            uint4 k0, k1;
            lds_read2(k0, k1, wc_base + (w - w0) * (IBDG_WC_WORDS * 4), wc_base + (w - w0) * (IBDG_WC_WORDS * 4) + 16);
            const int eK = (int)k0.x;
            const uint32_t kc0 = k0.z, kc1 = k0.w, kb0 = k1.x, kb1 = k1.y;     // (less 8 * 2^22 each, k_win_target_mx)
            // acc[0] / [1] = C(x) - 2 G(x,t0 / t1), acc[2] / [3] = G(x,t0 / t1) - A(x); v_mad_i32_i24 reads the low 24 bits of the
            // accumulator's own bits.  Products in the order of the other forms: (t0,x0) (t0,x1) (t1,x0) (t1,x1).
            uint32_t ad[8];
            ad[0] = mad24<8>(__float_as_uint(acc0[2]), kb0);   ad[1] = mad24<8>(__float_as_uint(acc0[0]), kc0);
            ad[2] = mad24<8>(__float_as_uint(acc1[2]), kb0);   ad[3] = mad24<8>(__float_as_uint(acc1[0]), kc0);
            ad[4] = mad24<8>(__float_as_uint(acc0[3]), kb1);   ad[5] = mad24<8>(__float_as_uint(acc0[1]), kc1);
            ad[6] = mad24<8>(__float_as_uint(acc1[3]), kb1);   ad[7] = mad24<8>(__float_as_uint(acc1[1]), kc1);
            uint2 pq[8];
            lds_read_pow8_b64(pq, ad);
            double val[4];
            if (a.rho_shift == 8) {
                const uint32_t e1 = (uint32_t)eK + tab1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double m1 = __hiloint2double((int)pq[2 * i].y, (int)pq[2 * i].x);
                    const double m2 = __hiloint2double((int)pq[2 * i + 1].y, (int)pq[2 * i + 1].x);
                    val[i] = __builtin_ldexp(m1 * m2, (int)(e1 - ad[2 * i]));
                }
            } else {
                const uint32_t e8 = mad24r(tab1, a.rho_shift, (uint32_t)eK << 3);
                const uint32_t ms = 0u - a.rho_shift;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const double m1 = __hiloint2double((int)pq[2 * i].y, (int)pq[2 * i].x);
                    const double m2 = __hiloint2double((int)pq[2 * i + 1].y, (int)pq[2 * i + 1].x);
                    val[i] = __builtin_ldexp(m1 * m2, (int)mad24r(ad[2 * i], ms, e8) >> 3);
                }
            }
            return wgt * (((val[0] + val[1]) + val[2]) + val[3]);
        };
        // the sum over the wave's 64 individuals in the tree of wave_sum2 (the lane number's bits in turn), by DPP moves alone:
        // this form has no scratch in LDS (a third ring slot per wave at the same four workgroups a CU measured 3 % slower)
        uint32_t w = w0;
        while (s < nseg) {                                    // (window_sum advances s)
            const double tot = wave_sum_lane63_only(window_sum(w));
            if (lane == 63)
                a.partial[(((size_t)(a.t_base + t) * a.n_win + w) * a.n_chunks + c) * 2 + 1] = tot;
            ++w;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    for (uint32_t w = w0; s < nseg; ++w) {               // one window per turn (a run's windows are consecutive)
        uint32_t flags;
        if constexpr (MX) {
            IBDG_SEGMENT_MX(true)
            while (!(flags & (1u << 15)))                // (a run ends with the last segment of a window)
                IBDG_SEGMENT_MX(false)
        } else {
            IBDG_SEGMENT(true)                           // its first segment starts the counters
            while (!(flags & (1u << 13)) && s < nseg)    // the others add to them
                IBDG_SEGMENT(false)
        }
        {
        {
            uint4 k0, k1;                           // the window's constants, broadcast into VGPRs
            lds_read2(k0, k1, wc_base + (w - w0) * (IBDG_WC_WORDS * 4), wc_base + (w - w0) * (IBDG_WC_WORDS * 4) + 16);
            const int eK = (int)k0.x;
            // table addresses of 16*AT, 16*<t0,cov>, 16*<t1,cov>, 16*(AT-<t0,alt>), 16*(AT-<t1,alt>), 0
            const uint32_t kAT = k0.y, kc0 = k0.z, kc1 = k0.w, kb0 = k1.x, kb1 = k1.y, ktab2 = k1.z;
            uint32_t C0, C1, G00, G01, G10, G11, a0, a1;
            const uint32_t CH = planes_sum<FC>(ch);
            if constexpr (MX) {
                C0 = (uint32_t)acc0[0]; a0 = (uint32_t)acc0[1]; G00 = (uint32_t)acc0[2]; G10 = (uint32_t)acc0[3];
                C1 = (uint32_t)acc1[0]; a1 = (uint32_t)acc1[1]; G01 = (uint32_t)acc1[2]; G11 = (uint32_t)acc1[3];
            } else {
                C0 = planes_sum<FC>(c0); C1 = planes_sum<FC>(c1);
                G00 = planes_sum<FC>(g00); G01 = planes_sum<FC>(g01);
                G10 = planes_sum<FC>(g10); G11 = planes_sum<FC>(g11);
                a0 = planes_sum<FA>(A0); a1 = planes_sum<FA>(A1);
            }
            // ad[2i] / ad[2i+1]: where rho^E2 / sigma^E3 of product i sit (table base + 16 * exponent), with
            //   pDg[x0+x1] (ibdgem.c:715): E3 = C0 + C1 - 2 CH          E2 = AT - a0 - a1 + CH
            //   pDg[At+hx] (:716-719):     E3 = <t,cov> + Cx - 2 G(x,t)  E2 = AT - <t,alt> - ax + G(x,t)
            uint32_t ad[10];
            if constexpr (PLAIN) {
                // 8-byte entries; rho^E2 sits in its table as rho^E2 * 2^(s E2): the product's exponent is made up for it from
                // the table address itself, eK - s E2 = (8 eK + s tab1 - s ad) >> 3
                ad[0] = lshl_add<3>(CH - (a0 + a1), kAT);
                ad[1] = lshl_add<3>(mad24<-2>(CH, C0 + C1), ktab2);
                ad[2] = lshl_add<3>(G00, mad24<-8>(a0, kb0));   ad[3] = mad24<-16>(G00, lshl_add<3>(C0, kc0));
                ad[4] = lshl_add<3>(G01, mad24<-8>(a1, kb0));   ad[5] = mad24<-16>(G01, lshl_add<3>(C1, kc0));
                ad[6] = lshl_add<3>(G10, mad24<-8>(a0, kb1));   ad[7] = mad24<-16>(G10, lshl_add<3>(C0, kc1));
                ad[8] = lshl_add<3>(G11, mad24<-8>(a1, kb1));   ad[9] = mad24<-16>(G11, lshl_add<3>(C1, kc1));
                uint2 pq[10];
                lds_read_pow10_b64(pq, ad);
                const uint32_t tab1 = (uint32_t)(uintptr_t)(lds_void *)tab_lds;
                double val[5];
                if (a.rho_shift == 8) {                  // (the host's choice where the table allows it): eK - 8 E2 = eK + tab1 - ad
                    const uint32_t e1 = (uint32_t)eK + tab1;
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        const double m1 = __hiloint2double((int)pq[2 * i].y, (int)pq[2 * i].x);
                        const double m2 = __hiloint2double((int)pq[2 * i + 1].y, (int)pq[2 * i + 1].x);
                        val[i] = __builtin_ldexp(m1 * m2, (int)(e1 - ad[2 * i]));
                    }
                } else {
                    const uint32_t e8 = mad24r(tab1, a.rho_shift, (uint32_t)eK << 3);        // 8 eK + s tab1 (wave-uniform)
                    const uint32_t ms = 0u - a.rho_shift;
#pragma unroll
                    for (int i = 0; i < 5; ++i) {
                        const double m1 = __hiloint2double((int)pq[2 * i].y, (int)pq[2 * i].x);
                        const double m2 = __hiloint2double((int)pq[2 * i + 1].y, (int)pq[2 * i + 1].x);
                        val[i] = __builtin_ldexp(m1 * m2, (int)mad24r(ad[2 * i], ms, e8) >> 3);
                    }
                }
                double s0 = wgt * val[0];                                   // :743
                if (a.p2_out)
                    a.p2_out[(size_t)w * a.lanes + c * 64 + lane] = s0;
                double s1 = wgt * (((val[1] + val[2]) + val[3]) + val[4]);  // :744-745
                const double tot = a.sum_dpp ? wave_sum2_dpp(s0, s1, scr_w, scr_r) : wave_sum2(s0, s1, scr_w, scr_r);
                if ((lane & 31) == 31)
                    a.partial[(((size_t)(a.t_base + t) * a.n_win + w) * a.n_chunks + c) * 2 + (lane >> 5)] = tot;
                continue;
            }
            ad[0] = lshl_add<4>(CH - (a0 + a1), kAT);
            ad[1] = lshl_add<4>(mad24<-2>(CH, C0 + C1), ktab2);
            ad[2] = lshl_add<4>(G00, mad24<-16>(a0, kb0));   ad[3] = mad24r(G00, m32, lshl_add<4>(C0, kc0));   // A0, h0
            ad[4] = lshl_add<4>(G01, mad24<-16>(a1, kb0));   ad[5] = mad24r(G01, m32, lshl_add<4>(C1, kc0));   // A0, h1
            ad[6] = lshl_add<4>(G10, mad24<-16>(a0, kb1));   ad[7] = mad24r(G10, m32, lshl_add<4>(C0, kc1));   // A1, h0
            ad[8] = lshl_add<4>(G11, mad24<-16>(a1, kb1));   ad[9] = mad24r(G11, m32, lshl_add<4>(C1, kc1));   // A1, h1
            uint4 pw[10];                           // (not the matrix-core form with its tables in LDS: that one has left above)
            if (TAB_LDS) {
                lds_read_pow10(pw, ad);
            } else {
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    pw[2 * i] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(pow_1me) + ad[2 * i]);
                    pw[2 * i + 1] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(pow_eps) + ad[2 * i + 1]);
                }
            }
            const double P2 = ld_value(eK, pw[0], pw[1]);
            const double Q00 = ld_value(eK, pw[2], pw[3]);
            const double Q01 = ld_value(eK, pw[4], pw[5]);
            const double Q10 = ld_value(eK, pw[6], pw[7]);
            const double Q11 = ld_value(eK, pw[8], pw[9]);
            double s0 = wgt * P2;                                   // :743
            double s1 = wgt * (((Q00 + Q01) + Q10) + Q11);          // :744-745
            if (a.p2_out)
                a.p2_out[(size_t)w * a.lanes + c * 64 + lane] = s0;
            const double tot = wave_sum2(s0, s1, scr_w, scr_r);      // first half: sum of s0, second half: of s1
            if ((lane & 31) == 31)
                a.partial[(((size_t)(a.t_base + t) * a.n_win + w) * a.n_chunks + c) * 2 + (lane >> 5)] = tot;
        }
        }
    }
    // leave no direct-to-LDS load in flight when the wave ends
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#undef IBDG_SEGMENT
#undef IBDG_SEGMENT_MX
#undef IBDG_SEGMENT_X1

// ---------------------------------------------------------------------------
// Several comparison individuals per workgroup (BASELINE.json configs[4]: hundreds of them against
// one panel).  Of the nine sums per background individual and window only the four G(x,t) depend
// on the comparison individual; A(x0) A(x1) C(x0) C(x1) C(x0&x1), the tile words, the masks and
// the product of the individual's own genotype factors (ibdgem.c:715) do not.  A workgroup of
// k_ld_popcount_mt therefore serves TB comparison individuals at once: per segment the common
// 13 counts are taken once and 12 more per individual; per window the common exponents and P2
// once, then per individual the four IBD1 products, its weight (the individual itself is
// excluded from its own background, ibdgem.c:714) and its two wave sums -- in exactly the
// operations and order of k_ld_popcount, so the results are the same bits.
//
// LDS images (written by k_win_target_mt):
//   segment, 8 + 2 TB words:        flags cov0 cov1 cov2 | alt0 alt1 - - | TB x {t0 t1}   (IBDG_RECM_WORDS)
//   window, 8 + 4 TB words:         mK(2) eK CT | 16*AT 0 - - | TB x {16*<t0,cov> 16*<t1,cov> 16*(AT-<t0,alt>) 16*(AT-<t1,alt>)}
//   (table byte offsets like IBDG_WC_WORDS; the staging adds the table bases, stage_wcm_base)
// ---------------------------------------------------------------------------
#ifndef IBDG_MT
#define IBDG_MT 4
#endif
#define IBDG_RECM_WORDS (8 + 4 * ((IBDG_MT + 1) / 2))      /* target words padded to whole uint4 */
#define IBDG_WCM_WORDS (8 + 4 * IBDG_MT)

__global__ __launch_bounds__(256) void k_win_target_mt(PopArgs a, uint32_t *__restrict__ rec_ready,
                                                       uint32_t *__restrict__ wc_ready)
{
    constexpr int TB = IBDG_MT;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned g = blockIdx.y;                     // group of TB comparison individuals
    const uint4 *tt[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
        const uint32_t tgt = a.targets[a.t_base + g * TB + j];
        tt[j] = reinterpret_cast<const uint4 *>(a.t32) + (size_t)(tgt >> 6) * a.n_pairs * 64 + (tgt & 63);
    }
    if (i < a.n_segs) {
        const Seg S = a.segs[i];
        uint4 *o = reinterpret_cast<uint4 *>(rec_ready + ((size_t)g * a.n_segs + i) * IBDG_RECM_WORDS);
        o[0] = make_uint4(S.flags, S.cov[0], S.cov[1], S.cov[2]);
        o[1] = make_uint4(S.alt[0], S.alt[1], 0, 0);
#pragma unroll
        for (int j = 0; j < TB; j += 2) {
            const uint2 ta = tile_words(tt[j], S.tile);
            const uint2 tb = j + 1 < TB ? tile_words(tt[j + 1 < TB ? j + 1 : j], S.tile) : make_uint2(0, 0);
            o[2 + j / 2] = make_uint4(ta.x, ta.y, tb.x, tb.y);
        }
    }
    if ((i >> 3) < a.n_win) {
        const uint32_t w = i >> 3;
        uint32_t acc[TB][4];
#pragma unroll
        for (int j = 0; j < TB; ++j)
            acc[j][0] = acc[j][1] = acc[j][2] = acc[j][3] = 0;
        const uint32_t s1 = a.wconst[w + 1].seg_begin;
        for (uint32_t s = a.wconst[w].seg_begin + (i & 7); s < s1; s += 8) {
            const Seg &S = a.segs[s];
#pragma unroll
            for (int j = 0; j < TB; ++j) {
                const uint2 at = tile_words(tt[j], S.tile);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    acc[j][0] += (uint32_t)__popc(at.x & S.cov[k]) << k;
                    acc[j][1] += (uint32_t)__popc(at.y & S.cov[k]) << k;
                    acc[j][2] += (uint32_t)__popc(at.x & S.alt[k]) << k;
                    acc[j][3] += (uint32_t)__popc(at.y & S.alt[k]) << k;
                }
            }
        }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1)
#pragma unroll
            for (int j = 0; j < TB; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    acc[j][q] += __shfl_xor(acc[j][q], m);
        if ((i & 7) == 0) {
            const uint32_t *wcs = reinterpret_cast<const uint32_t *>(a.wconst + w);
            uint4 *o = reinterpret_cast<uint4 *>(wc_ready + ((size_t)g * a.n_win + w) * IBDG_WCM_WORDS);
            const uint32_t AT = wcs[4];
            o[0] = make_uint4(wcs[0], wcs[1], wcs[2], wcs[3]);
            o[1] = make_uint4(16 * AT, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TB; ++j)
                o[2 + j] = make_uint4(16 * acc[j][0], 16 * acc[j][1], 16 * (AT - acc[j][2]), 16 * (AT - acc[j][3]));
        }
    }
}

// Staging of the multi-individual window constants (j = index of the uint4 within its window's record)
__device__ __forceinline__ uint4 stage_wcm_base(uint4 v, uint32_t j, uint32_t tab1, uint32_t tab2)
{
    if (j == 1) {
        v.x += tab1;
        v.y = tab2;
    } else if (j >= 2) {
        v.x += tab2;
        v.y += tab2;
        v.z += tab1;
        v.w += tab1;
    }
    return v;
}

__device__ __forceinline__ void lds_fetch_mt(uint4 &h0, uint4 &h1, uint4 &h2, uint4 &h3, uint2 &x, uint32_t rec_addr,
                                             uint32_t x_addr)
{
    asm volatile("ds_read_b128 %0, %5\n\t"
                 "ds_read_b128 %1, %5 offset:16\n\t"
                 "ds_read_b128 %2, %5 offset:32\n\t"
                 "ds_read_b128 %3, %5 offset:48\n\t"
                 "ds_read_b64 %4, %6\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3), "=&v"(x)
                 : "v"(rec_addr), "v"(x_addr)
                 : "memory");
}

// One segment for TB comparison individuals (FIRST as in IBDG_SEGMENT of k_ld_popcount)
#define IBDG_SEGMENT_MT(FIRST)                                                                                     \
    {                                                                                                           \
        uint4 h0, h1, h2, h3;                                                                                   \
        uint2 x;                                                                                                \
        lds_fetch_mt(h0, h1, h2, h3, x, rec_addr, ring_lane + x_off);                                           \
        flags = __builtin_amdgcn_readfirstlane(h0.x);                                                           \
        const uint32_t adv = (flags >> 4) & 0xff;                                                               \
        if (adv) {                                                                                              \
            for (uint32_t i = 0; i < adv; ++i, ++q_issue)                                                       \
                if (q_issue <= q_last)                                                                          \
                    __builtin_amdgcn_global_load_lds((const void *)(xt + (size_t)q_issue * 64),                 \
                                                     (lds_void *)(ring + ((q_issue - q0) % NS) * 1024), 16, 0, IBDG_TILE_AUX); \
            if (q_issue - 1 <= q_last)                                                                          \
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS - 1) : "memory");                                   \
            else                                                                                                \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                \
        }                                                                                                       \
        x_off = (flags & 7) * 1024 + ((flags >> 3) & 1) * 8;                                                    \
        const uint32_t cov0 = h0.y, cov1 = h0.z, cov2 = h0.w, alt0 = h1.x, alt1 = h1.y;                         \
        const uint32_t tw4[4][2] = {{h2.x, h2.y}, {h2.z, h2.w}, {h3.x, h3.y}, {h3.z, h3.w}};                    \
        const uint32_t (&tw)[4][2] = tw4;                                                                       \
        const uint32_t hom = x.x & x.y;                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        count_plane_mt<FIRST>(c0[0], c1[0], ch[0], gq, 0, x.x, x.y, hom, cov0, tw);                             \
        count_plane_mt<FIRST>(c0[1], c1[1], ch[1], gq, 1, x.x, x.y, hom, cov1, tw);                             \
        count_plane_mt<FIRST>(c0[2], c1[2], ch[2], gq, 2, x.x, x.y, hom, cov2, tw);                             \
        count_alt<FIRST>(A0, A1, x.x, x.y, alt0, alt1);                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        if (flags & (1u << 12)) {                                                                               \
            const uint32_t ncov = (flags >> 16) & 0xff, nalt = flags >> 24;                                     \
            for (uint32_t k = FC; k < ncov; ++k) {                                                              \
                const uint32_t cov = segs[seg0 + s].cov[k];              /* uniform: scalar load */             \
                const uint32_t u0 = x.x & cov, u1 = x.y & cov;                                                  \
                c0[0] += (uint32_t)__popc(u0) << k;                                                             \
                c1[0] += (uint32_t)__popc(u1) << k;                                                             \
                ch[0] += (uint32_t)__popc(hom & cov) << k;                                                      \
                _Pragma("unroll") for (int j = 0; j < TB; ++j)                                                  \
                {                                                                                               \
                    gq[j][0][0] += (uint32_t)__popc(u0 & tw[j][0]) << k;                                        \
                    gq[j][1][0] += (uint32_t)__popc(u1 & tw[j][0]) << k;                                        \
                    gq[j][2][0] += (uint32_t)__popc(u0 & tw[j][1]) << k;                                        \
                    gq[j][3][0] += (uint32_t)__popc(u1 & tw[j][1]) << k;                                        \
                }                                                                                               \
            }                                                                                                   \
            for (uint32_t k = FA; k < nalt; ++k) {                                                              \
                const uint32_t alt = segs[seg0 + s].alt[k];                                                     \
                A0[0] += (uint32_t)__popc(x.x & alt) << k;                                                      \
                A1[0] += (uint32_t)__popc(x.y & alt) << k;                                                      \
            }                                                                                                   \
        }                                                                                                       \
        rec_addr += IBDG_RECM_WORDS * 4;                                                                        \
        ++s;                                                                                                    \
    }

template <int NS, bool TAB_LDS>
__global__ __launch_bounds__(512) void k_ld_popcount_mt(const uint4 *__restrict__ t32,
                                                        const Seg *__restrict__ segs,
                                                        const uint32_t *__restrict__ rec_ready,
                                                        const WinConst *__restrict__ wconst,
                                                        const uint32_t *__restrict__ wc_ready,
                                                        const uint4 *__restrict__ pow_1me,
                                                        const uint4 *__restrict__ pow_eps,
                                                        const uint32_t *__restrict__ run_begin,
                                                        PopArgs a)
{
    constexpr int FC = 3, FA = 2, TB = IBDG_MT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    const unsigned g = blockIdx.z;                     // group of TB comparison individuals
    const uint32_t run = blockIdx.x / a.n_cgroups, cgroup = blockIdx.x - run * a.n_cgroups;
    const uint32_t w0 = run_begin[run], w1 = run_begin[run + 1];
    const uint32_t seg0 = wconst[w0].seg_begin, seg1 = wconst[w1].seg_begin;
    const uint32_t nseg = seg1 - seg0;
    if (nseg == 0)
        return;

    // ---- LDS carve-up (ld_popcount_lds_bytes with the record / constant sizes of this kernel)
    uint32_t *rec_lds = reinterpret_cast<uint32_t *>(smem);
    uint32_t *wc_lds = rec_lds + (size_t)a.max_seg * IBDG_RECM_WORDS;
    const size_t head = ((size_t)a.max_seg * IBDG_RECM_WORDS + (size_t)a.win_per_group * IBDG_WCM_WORDS) * 4 + 15;
    uint4 *tab_lds = reinterpret_cast<uint4 *>(smem + (head & ~(size_t)15));
    const size_t tab_bytes = TAB_LDS ? (size_t)a.tab_len * 32 : 0;
    char *ring0 = smem + ((head + tab_bytes + 1023) & ~(size_t)1023);

    // ---- prime the ring, then stage (see k_ld_popcount)
    const unsigned c = cgroup * a.waves_per_group + wave;
    const bool has_chunk = c < a.n_chunks;
    char *ring = ring0 + (size_t)wave * NS * 1024;
    const uint4 *xt = t32 + (size_t)c * a.n_pairs * 64 + lane;
    const uint32_t tile0 = segs[seg0].tile;
    const uint32_t q0 = tile0 >> 1, q_last = segs[seg1 - 1].tile >> 1;
    uint32_t q_issue = q0;
    if (has_chunk) {
#pragma unroll
        for (int i = 0; i < NS; ++i, ++q_issue)
            if (q_issue <= q_last)
                __builtin_amdgcn_global_load_lds((const void *)(xt + (size_t)q_issue * 64),
                                                 (lds_void *)(ring + ((q_issue - q0) % NS) * 1024), 16, 0, IBDG_TILE_AUX);
    }
    {
        const uint4 *rsrc = reinterpret_cast<const uint4 *>(rec_ready) + ((size_t)g * a.n_segs + seg0) * (IBDG_RECM_WORDS / 4);
        uint4 *rdst = reinterpret_cast<uint4 *>(rec_lds);
        for (uint32_t i = threadIdx.x; i < nseg * (IBDG_RECM_WORDS / 4); i += blockDim.x)
            rdst[i] = rsrc[i];
        const uint32_t stab1 = TAB_LDS ? (uint32_t)(uintptr_t)(lds_void *)tab_lds : 0u;
        const uint32_t stab2 = TAB_LDS ? stab1 + a.tab_len * 16 : 0u;
        const uint4 *wsrc = reinterpret_cast<const uint4 *>(wc_ready) + ((size_t)g * a.n_win + w0) * (IBDG_WCM_WORDS / 4);
        uint4 *wdst = reinterpret_cast<uint4 *>(wc_lds);
        for (uint32_t i = threadIdx.x; i < (w1 - w0) * (IBDG_WCM_WORDS / 4); i += blockDim.x)
            wdst[i] = stage_wcm_base(wsrc[i], i % (IBDG_WCM_WORDS / 4), stab1, stab2);
        if (TAB_LDS)
            for (uint32_t i = threadIdx.x; i < 2 * a.tab_len; i += blockDim.x)
                tab_lds[i] = i < a.tab_len ? pow_1me[i] : pow_eps[i - a.tab_len];
    }
    __syncthreads();
    if (!has_chunk)
        return;
    const uint32_t ring_lane = (uint32_t)(uintptr_t)(lds_void *)ring + lane * 16;
    // the wave's 1 KiB scratch for wave_sum2, behind the rings
    const uint32_t scr = (uint32_t)(uintptr_t)(lds_void *)(ring0 + (size_t)a.waves_per_group * NS * 1024 + wave * 1024);
    const uint32_t scr_w = scr + lane * 8, scr_r = scr + lane * 16;

    double wgt[TB];
#pragma unroll
    for (int j = 0; j < TB; ++j)
        wgt[j] = a.weight[(size_t)(a.t_base + g * TB + j) * a.lanes + c * 64 + lane];

    uint32_t c0[FC], c1[FC], ch[FC], A0[FA], A1[FA];
    uint32_t gq[TB][4][FC];          // G(x0,t0) G(x1,t0) G(x0,t1) G(x1,t1) per individual and plane
    if (q_issue - 1 <= q_last)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS - 1) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    uint32_t m32 = (uint32_t)-32;
    asm volatile("" : "+v"(m32));
    uint32_t x_off = (tile0 & 1) * 8;
    uint32_t rec_addr = (uint32_t)(uintptr_t)(lds_void *)rec_lds;
    const uint32_t wc_base = (uint32_t)(uintptr_t)(lds_void *)wc_lds;
    uint32_t s = 0;
    for (uint32_t w = w0; s < nseg; ++w) {               // one window per turn (a run's windows are consecutive)
        uint32_t flags;
        IBDG_SEGMENT_MT(true)                            // its first segment starts the counters
        while (!(flags & (1u << 13)) && s < nseg)        // the others add to them
            IBDG_SEGMENT_MT(false)
        {
        const uint32_t wc_addr = wc_base + (w - w0) * (IBDG_WCM_WORDS * 4);
        static_assert(TB == 4, "the six-read statement below fetches 8 + 4*4 words");
        uint4 k0, k1, kt4[4];                     // the window's constants of all TB individuals in one round trip
        asm volatile("ds_read_b128 %0, %6\n\t"
                     "ds_read_b128 %1, %6 offset:16\n\t"
                     "ds_read_b128 %2, %6 offset:32\n\t"
                     "ds_read_b128 %3, %6 offset:48\n\t"
                     "ds_read_b128 %4, %6 offset:64\n\t"
                     "ds_read_b128 %5, %6 offset:80\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(k0), "=&v"(k1), "=&v"(kt4[0]), "=&v"(kt4[1]), "=&v"(kt4[2]), "=&v"(kt4[3])
                     : "v"(wc_addr)
                     : "memory");
        const int eK = (int)k0.z;
        const uint32_t kAT = k1.x, ktab2 = k1.y;
        const uint32_t C0 = planes_sum<FC>(c0), C1 = planes_sum<FC>(c1), CH = planes_sum<FC>(ch);
        const uint32_t a0 = planes_sum<FA>(A0), a1 = planes_sum<FA>(A1);
        double P2;
        {
            // pDg[x0+x1] (ibdgem.c:715): E3 = C0 + C1 - 2 CH, E2 = AT - a0 - a1 + CH
            const uint32_t ad2 = lshl_add<4>(CH - (a0 + a1), kAT), ad3 = lshl_add<4>(mad24<-2>(CH, C0 + C1), ktab2);
            uint4 p1, p2;
            if (TAB_LDS) {
                lds_read2(p1, p2, ad2, ad3);
            } else {
                p1 = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(pow_1me) + ad2);
                p2 = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(pow_eps) + ad3);
            }
            P2 = ld_value(eK, p1, p2);
        }
#pragma unroll
        for (int j = 0; j < TB; ++j) {
            const uint4 kt = kt4[j];
            const uint32_t kc0 = kt.x, kc1 = kt.y, kb0 = kt.z, kb1 = kt.w;
            const uint32_t G00 = planes_sum<FC>(gq[j][0]), G01 = planes_sum<FC>(gq[j][1]);
            const uint32_t G10 = planes_sum<FC>(gq[j][2]), G11 = planes_sum<FC>(gq[j][3]);
            // pDg[At+hx] (:716-719): E3 = <t,cov> + Cx - 2 G(x,t), E2 = AT - <t,alt> - ax + G(x,t)
            uint32_t ad[8];
            ad[0] = lshl_add<4>(G00, mad24<-16>(a0, kb0));   ad[1] = mad24r(G00, m32, lshl_add<4>(C0, kc0));
            ad[2] = lshl_add<4>(G01, mad24<-16>(a1, kb0));   ad[3] = mad24r(G01, m32, lshl_add<4>(C1, kc0));
            ad[4] = lshl_add<4>(G10, mad24<-16>(a0, kb1));   ad[5] = mad24r(G10, m32, lshl_add<4>(C0, kc1));
            ad[6] = lshl_add<4>(G11, mad24<-16>(a1, kb1));   ad[7] = mad24r(G11, m32, lshl_add<4>(C1, kc1));
            uint4 pw[8];
            if (TAB_LDS) {
                lds_read_pow8(pw, ad);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    pw[2 * i] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(pow_1me) + ad[2 * i]);
                    pw[2 * i + 1] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const char *>(pow_eps) + ad[2 * i + 1]);
                }
            }
            const double Q00 = ld_value(eK, pw[0], pw[1]);
            const double Q01 = ld_value(eK, pw[2], pw[3]);
            const double Q10 = ld_value(eK, pw[4], pw[5]);
            const double Q11 = ld_value(eK, pw[6], pw[7]);
            double s0 = wgt[j] * P2;                                   // :743
            double s1 = wgt[j] * (((Q00 + Q01) + Q10) + Q11);          // :744-745
            const double tot = wave_sum2(s0, s1, scr_w, scr_r);      // lane 31: sum of s0, lane 63: of s1
            if ((lane & 31) == 31)
                a.partial[(((size_t)(a.t_base + g * TB + j) * a.n_win + w) * a.n_chunks + c) * 2 + (lane >> 5)] = tot;
        }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

#undef IBDG_SEGMENT_MT

// Sum the per-chunk partials of a window and take the background average (src/ibdgem.c:751-752).
// One wave per window: lane c adds chunks c, c+64, .. (coalesced 16-byte loads), then the wave
// sums its lanes in the fixed order of wave_sum_to_lane63 -- the same order whatever the launch.
__global__ __launch_bounds__(256) void k_ld_finalize(PopFinalArgs a)
{
    const uint32_t w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (w >= a.n_win)
        return;
    const unsigned lane = threadIdx.x & 63;
    const unsigned t = blockIdx.y + a.t_base;
    double t0 = 0.0, t1 = 0.0;
    if (a.halves) {
        // k_ld_mfma sums 32 individuals per wave: a chunk's sum is (individuals 0..31) + (32..63), the last
        // addition of the 64-lane tree of wave_sum2
        const double2 *p = reinterpret_cast<const double2 *>(a.partial) + ((size_t)(t - a.p_t0) * a.n_win + w) * a.n_chunks * 2;
        for (uint32_t c = lane; c < a.n_chunks; c += 64) {
            const double2 v = p[2 * c], u = p[2 * c + 1];
            t0 += v.x + u.x;
            t1 += v.y + u.y;
        }
    } else {
        const double2 *p = reinterpret_cast<const double2 *>(a.partial) + ((size_t)(t - a.p_t0) * a.n_win + w) * a.n_chunks;
        for (uint32_t c = lane; c < a.n_chunks; c += 64) {
            const double2 v = p[c];
            t0 += v.x;
            t1 += v.y;
        }
    }
    uint32_t tgt_own = a.p2c ? a.targets[t] : 0u;
    IBDG_CHECK_TGT(tgt_own, a.p2c ? a.lanes : 1u, "k_ld_finalize");
    t0 = a.p2c ? ibd0_from_pass(a.p2c, a.p2w, a.lanes, a.n_chunks, w, tgt_own, lane) : wave_sum_to_lane63(t0);
    t1 = wave_sum_to_lane63(t1);
    if (lane == 63) {
        const int nref = a.n_refpanel[t];
        const double mK = a.wconst[w].mK;             // mantissa of K' (its exponent went into every term)
        double *o = a.win_ll + ((size_t)t * a.n_win + w) * 3;
        o[0] = (t0 * mK) / (double)nref;
        o[1] = (t1 * mK) / (double)(nref * 4);
    }
}

// ---------------------------------------------------------------------------
void launch_transpose32(const uint64_t *panel, uint32_t stride, size_t n_rows, uint32_t n_chunks,
                        uint32_t n_pairs, uint32_t *t32, hipStream_t st)
{
    if (n_pairs == 0)
        return;
    hipLaunchKernelGGL(k_transpose32, dim3(n_pairs, (n_chunks + 7) / 8), dim3(512), 0, st, panel, stride,
                       n_rows, n_chunks, n_pairs, reinterpret_cast<uint4 *>(t32));
}

void launch_gather_transpose32(const uint64_t *panel, uint32_t stride, const uint2 *rec_cov, uint32_t n_cov,
                               uint32_t window, uint32_t win_rows, uint32_t n_chunks, uint32_t n_pairs, uint32_t *t32,
                               hipStream_t st)
{
    if (n_pairs == 0)
        return;
    hipLaunchKernelGGL(k_gather_transpose32, dim3(n_pairs, (n_chunks + 7) / 8), dim3(512), 0, st, panel, stride, rec_cov,
                       n_cov, window, win_rows, n_chunks, n_pairs, reinterpret_cast<uint4 *>(t32));
}

// ev.start / ev.stop (may be null): events the dispatch itself updates with the kernel's start and
// stop time (hipExtLaunchKernel) -- no event-record packet on the stream.
void launch_win_target(const PopArgs &a, unsigned n_targets, hipStream_t st, KernelEvents ev)
{
    if (a.n_win == 0)
        return;
    if (a.mx_counts && a.ibd1 && a.frag_base && a.tab_in_lds) {
        const uint32_t n = a.n_segs * 2 > a.n_win * 8 ? a.n_segs * 2 : a.n_win * 8;
        hipExtLaunchKernelGGL(k_win_target_x1, dim3((n + 255) / 256, n_targets), dim3(256), 0, st, ev.start, ev.stop, 0, a, a.frag_base,
                              const_cast<uint32_t *>(a.rec_ready), const_cast<uint32_t *>(a.wc_ready));
        return;
    }
    if (a.mx_counts) {
        const uint32_t n = a.n_segs * 4 > a.n_win * 8 ? a.n_segs * 4 : a.n_win * 8;
        hipExtLaunchKernelGGL(k_win_target_mx, dim3((n + 255) / 256, n_targets), dim3(256), 0, st, ev.start, ev.stop, 0, a,
                              const_cast<uint32_t *>(a.rec_ready), const_cast<uint32_t *>(a.wc_ready));
        return;
    }
    const uint32_t n = a.n_segs > a.n_win * 8 ? a.n_segs : a.n_win * 8;
    hipExtLaunchKernelGGL(k_win_target, dim3((n + 255) / 256, n_targets), dim3(256), 0, st, ev.start, ev.stop, 0, a,
                          const_cast<uint32_t *>(a.rec_ready), const_cast<uint32_t *>(a.wc_ready));
}

// the IBD1 form's three fragments per segment that do not depend on the comparison individual (72 bytes per segment)
void launch_frag_base(const PopArgs &a, uint32_t *frag_base, hipStream_t st)
{
    if (a.n_segs == 0)
        return;
    hipLaunchKernelGGL(k_frag_base, dim3((a.n_segs * 3 + 255) / 256), dim3(256), 0, st, a, frag_base);
}

size_t ld_popcount_rec_bytes(int mx_counts) { return (mx_counts ? IBDG_RECX_WORDS : IBDG_REC_WORDS) * 4; }

// LDS of one workgroup: records + window constants (+ power tables) rounded to 1 KiB, then 8 rings.
size_t ld_popcount_lds_bytes(uint32_t max_seg, uint32_t win_per_group, uint32_t tab_len, int tab_in_lds,
                             int ring_slots, int multi_target)
{
    if (multi_target == 3)       // the IBD1 form of 2: no scratch
        return ld_popcount_lds_bytes(max_seg, win_per_group, tab_len, tab_in_lds, ring_slots, 2) - 8 * 1024;
    // multi_target: 0 = one comparison individual (vector-ALU counts), 1 = groups of IBDG_MT, 2 = one, counts on the matrix cores
    const size_t rec_words = multi_target == 1 ? IBDG_RECM_WORDS : multi_target == 2 ? IBDG_RECX_WORDS : IBDG_REC_WORDS;
    const size_t wc_words = multi_target == 1 ? IBDG_WCM_WORDS : IBDG_WC_WORDS;
    size_t head = ((size_t)max_seg * rec_words + (size_t)win_per_group * wc_words) * 4 + 15;
    if (tab_in_lds)
        head += (size_t)tab_len * (multi_target == 2 ? 16 : 32);      // (matrix-core form: plain doubles)
    return ((head + 1023) & ~(size_t)1023) + 8 * (size_t)ring_slots * 1024 + 8 * 1024;    // rings + wave_sum2 scratch
}

template <int NS, bool TAB, bool MX, bool IBD1 = false>
static int launch_pop(const PopArgs &a, dim3 grid, hipStream_t st, KernelEvents ev)
{
    const size_t lds = ld_popcount_lds_bytes(a.max_seg, a.win_per_group, a.tab_len, TAB, NS, IBD1 ? 3 : MX ? 2 : 0);
    auto kern = k_ld_popcount<NS, TAB, MX, IBD1>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return 1;
    hipExtLaunchKernelGGL(kern, grid, dim3(64 * a.waves_per_group), (uint32_t)lds, st, ev.start, ev.stop, 0,
                          (const uint4 *)a.t32, a.segs, a.rec_ready, a.wconst, a.wc_ready, (const uint4 *)a.pow_1me,
                          (const uint4 *)a.pow_eps, a.run_begin, a);
    return 0;
}

int launch_ld_popcount(const PopArgs &a, unsigned n_targets, int planes, hipStream_t st, KernelEvents ev)
{
    if (a.n_win == 0)
        return 0;
    if (planes < 1 || planes > 8)
        return 1;
    dim3 grid(a.n_runs * a.n_cgroups + (a.fin_prev ? (a.n_win + a.waves_per_group - 1) / a.waves_per_group : 0), 1, n_targets);
    if (a.ibd1) {
        if (!a.mx_counts || !a.tab_in_lds || a.p2_out)
            return 1;
        if (a.ring_slots == 2)
            return launch_pop<2, true, true, true>(a, grid, st, ev);
        if (a.ring_slots == 3)
            return launch_pop<3, true, true, true>(a, grid, st, ev);
        if (a.ring_slots == 4)
            return launch_pop<4, true, true, true>(a, grid, st, ev);
        return launch_pop<8, true, true, true>(a, grid, st, ev);
    }
    if (a.mx_counts) {
        if (a.ring_slots == 2)
            return a.tab_in_lds ? launch_pop<2, true, true>(a, grid, st, ev) : launch_pop<2, false, true>(a, grid, st, ev);
        if (a.ring_slots == 3)
            return a.tab_in_lds ? launch_pop<3, true, true>(a, grid, st, ev) : launch_pop<3, false, true>(a, grid, st, ev);
        if (a.ring_slots == 4)
            return a.tab_in_lds ? launch_pop<4, true, true>(a, grid, st, ev) : launch_pop<4, false, true>(a, grid, st, ev);
        return a.tab_in_lds ? launch_pop<8, true, true>(a, grid, st, ev) : launch_pop<8, false, true>(a, grid, st, ev);
    }
    if (a.ring_slots == 2)
        return a.tab_in_lds ? launch_pop<2, true, false>(a, grid, st, ev) : launch_pop<2, false, false>(a, grid, st, ev);
    if (a.ring_slots == 3)
        return a.tab_in_lds ? launch_pop<3, true, false>(a, grid, st, ev) : launch_pop<3, false, false>(a, grid, st, ev);
    if (a.ring_slots == 4)
        return a.tab_in_lds ? launch_pop<4, true, false>(a, grid, st, ev) : launch_pop<4, false, false>(a, grid, st, ev);
    return a.tab_in_lds ? launch_pop<8, true, false>(a, grid, st, ev) : launch_pop<8, false, false>(a, grid, st, ev);
}

// The same for groups of IBDG_MT comparison individuals (a.t_base = first of them, n_groups groups)
int ld_popcount_mt_width(void) { return IBDG_MT; }
size_t ld_popcount_mt_rec_bytes(void) { return IBDG_RECM_WORDS * 4; }
size_t ld_popcount_mt_wc_bytes(void) { return IBDG_WCM_WORDS * 4; }

void launch_win_target_mt(const PopArgs &a, unsigned n_groups, hipStream_t st, KernelEvents ev)
{
    if (a.n_win == 0 || n_groups == 0)
        return;
    const uint32_t n = a.n_segs > a.n_win * 8 ? a.n_segs : a.n_win * 8;
    hipExtLaunchKernelGGL(k_win_target_mt, dim3((n + 255) / 256, n_groups), dim3(256), 0, st, ev.start, ev.stop, 0, a,
                          const_cast<uint32_t *>(a.rec_ready), const_cast<uint32_t *>(a.wc_ready));
}

template <int NS, bool TAB>
static int launch_pop_mt(const PopArgs &a, dim3 grid, hipStream_t st, KernelEvents ev)
{
    const size_t lds = ld_popcount_lds_bytes(a.max_seg, a.win_per_group, a.tab_len, TAB, NS, 1);
    auto kern = k_ld_popcount_mt<NS, TAB>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess)
        return 1;
    hipExtLaunchKernelGGL(kern, grid, dim3(64 * a.waves_per_group), (uint32_t)lds, st, ev.start, ev.stop, 0,
                          (const uint4 *)a.t32, a.segs, a.rec_ready, a.wconst, a.wc_ready, (const uint4 *)a.pow_1me,
                          (const uint4 *)a.pow_eps, a.run_begin, a);
    return 0;
}

int launch_ld_popcount_mt(const PopArgs &a, unsigned n_groups, hipStream_t st, KernelEvents ev)
{
    if (a.n_win == 0 || n_groups == 0)
        return 0;
    dim3 grid(a.n_runs * a.n_cgroups, 1, n_groups);
    if (a.ring_slots == 2)
        return a.tab_in_lds ? launch_pop_mt<2, true>(a, grid, st, ev) : launch_pop_mt<2, false>(a, grid, st, ev);
    if (a.ring_slots == 3)
        return a.tab_in_lds ? launch_pop_mt<3, true>(a, grid, st, ev) : launch_pop_mt<3, false>(a, grid, st, ev);
    if (a.ring_slots == 4)
        return a.tab_in_lds ? launch_pop_mt<4, true>(a, grid, st, ev) : launch_pop_mt<4, false>(a, grid, st, ev);
    return a.tab_in_lds ? launch_pop_mt<8, true>(a, grid, st, ev) : launch_pop_mt<8, false>(a, grid, st, ev);
}

void launch_ld_finalize(const PopFinalArgs &a, unsigned n_targets, hipStream_t st, KernelEvents ev)
{
    if (a.n_win == 0)
        return;
    hipExtLaunchKernelGGL(k_ld_finalize, dim3((a.n_win + 3) / 4, n_targets), dim3(256), 0, st, ev.start, ev.stop, 0, a);
}

}  // namespace ibdg
