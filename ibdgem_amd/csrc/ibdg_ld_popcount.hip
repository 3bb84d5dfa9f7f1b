// ibdg_ld_popcount.hip -- the fast --LD kernel: exponent counting on a tile-transposed panel.
//
// What it computes is what src/ibdgem.c:669-722 and :736-753 of the reference compute:
// for every background individual the five window products of P(D|G) factors, then the
// background averages.  How: every factor is one of (src/ibd-math.c:57-70)
//     pDg[0] = C (1-e)^r e^a      pDg[1] = C (1/2)^(r+a)      pDg[2] = C (1-e)^a e^r
// (C = binomial coefficient, r/a = n_ref/n_alt of the row, e = epsilon), so a product over the
// rows of a window is exactly
//     prod = K * (1-e)^E1 * e^E2 * 2^-E3,    K = prod C,
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
//     ldexp(mK * m1[E1] * m2[E2], eK + e1[E1] + e2[E2] - E3)
// (tables of (1-e)^n and e^n as mantissa/exponent pairs, built on the host in extended
// precision) carries ~4 roundings, i.e. it is CLOSER to the exact product than the reference's
// 100 sequential multiplications; the two agree to ~1e-14 relative (documented bar: 1e-10).
// Values below the double range come out as 0/subnormal from the final ldexp, like the
// reference's running product.  The host enables this kernel only when the P(D|G) table is
// the unclamped binomial form (no DBL_MIN clamp, exact coefficients); otherwise the strict
// multiplying kernel in ibdg_kernels.hip is used.
#include "ibdg_kernels.h"

#include <hip/hip_runtime.h>

namespace ibdg {

// ---------------------------------------------------------------------------
// Panel transposition (once per upload): site-major rows -> t32[chunk][tile][lane][plane],
// bit j of a word = row 32*tile + j.  One wave per (tile, chunk); the row words are
// wave-uniform, each lane extracts its own individual's bit.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_transpose32(const uint64_t *__restrict__ panel,
                                                     uint32_t stride, size_t n_rows,
                                                     uint32_t n_chunks, uint32_t n_tiles,
                                                     uint32_t *__restrict__ t32)
{
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    const unsigned c = blockIdx.y * 4 + wave;
    if (c >= n_chunks)
        return;
    const uint32_t tile = blockIdx.x;
    uint32_t x0 = 0, x1 = 0;
    const size_t r0 = (size_t)tile * 32;
#pragma unroll 8
    for (unsigned j = 0; j < 32; ++j) {
        const size_t r = r0 + j;
        if (r < n_rows) {
            const uint64_t w0 = panel[r * stride + 2 * c], w1 = panel[r * stride + 2 * c + 1];
            x0 |= (uint32_t)((w0 >> lane) & 1u) << j;
            x1 |= (uint32_t)((w1 >> lane) & 1u) << j;
        }
    }
    uint2 *dst = reinterpret_cast<uint2 *>(t32) + ((size_t)c * n_tiles + tile) * 64 + lane;
    *dst = make_uint2(x0, x1);
}

// ---------------------------------------------------------------------------
// Per (window, target): <t0,cov>, <t1,cov>, <t0,alt>, <t1,alt> over the window's rows.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_win_target(PopArgs a, WinTarget *__restrict__ out)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.n_win)
        return;
    const unsigned t = blockIdx.y;
    const uint32_t tgt = a.targets[t];
    const uint2 *tt = reinterpret_cast<const uint2 *>(a.t32) + (size_t)(tgt >> 6) * a.n_tiles * 64 + (tgt & 63);
    WinTarget r = {0, 0, 0, 0};
    const uint32_t s1 = a.wconst[w + 1].seg_begin;
    for (uint32_t s = a.wconst[w].seg_begin; s < s1; ++s) {
        const Seg &S = a.segs[s];
        const uint2 at = tt[(size_t)S.tile * 64];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            r.a0cov += (uint32_t)__popc(at.x & S.cov[k]) << k;
            r.a1cov += (uint32_t)__popc(at.y & S.cov[k]) << k;
            r.a0alt += (uint32_t)__popc(at.x & S.alt[k]) << k;
            r.a1alt += (uint32_t)__popc(at.y & S.alt[k]) << k;
        }
    }
    out[(size_t)t * a.n_win + w] = r;
}

// K * (1-e)^E1 * e^E2 * 2^-E3
__device__ __forceinline__ double ld_value(const PopArgs &a, double mK, int eK, uint32_t E1,
                                           uint32_t E2, uint32_t E3)
{
    const PowEntry p1 = a.pow_1me[E1];
    const PowEntry p2 = a.pow_eps[E2];
    const double m = (mK * p1.m) * p2.m;
    return __builtin_ldexp(m, eK + p1.e + p2.e - (int)E3);
}

// A segment record held in (scalar) registers: five 16-byte loads.
struct SegRegs {
    uint4 h, c0, c1, a0, a1;
    __device__ __forceinline__ void load(const Seg *p)
    {
        const uint4 *q = reinterpret_cast<const uint4 *>(p);
        h = q[0]; c0 = q[1]; c1 = q[2]; a0 = q[3]; a1 = q[4];
    }
    __device__ __forceinline__ uint32_t tile() const { return h.x; }
    __device__ __forceinline__ uint32_t win() const { return h.y; }
    __device__ __forceinline__ uint32_t last() const { return h.z; }
    __device__ __forceinline__ uint32_t cov(int k) const
    {
        return k == 0 ? c0.x : k == 1 ? c0.y : k == 2 ? c0.z : k == 3 ? c0.w
             : k == 4 ? c1.x : k == 5 ? c1.y : k == 6 ? c1.z : c1.w;
    }
    __device__ __forceinline__ uint32_t alt(int k) const
    {
        return k == 0 ? a0.x : k == 1 ? a0.y : k == 2 ? a0.z : k == 3 ? a0.w
             : k == 4 ? a1.x : k == 5 ? a1.y : k == 6 ? a1.z : a1.w;
    }
};

template <int KP>
__device__ __forceinline__ uint32_t planes_sum(const uint32_t (&v)[KP])
{
    uint32_t s = v[0];
#pragma unroll
    for (int k = 1; k < KP; ++k)
        s += v[k] << k;
    return s;
}

// ---------------------------------------------------------------------------
// The --LD loop.  A wave owns one chunk of 64 background individuals (one per lane) and a
// run of consecutive windows; it streams that chunk's tiles (contiguous in memory, 8 bytes
// per lane and tile) exactly once.  Segment records (wave-uniform) arrive through the scalar
// path; all-zero bit-planes are skipped by uniform branches.
// At the end of each window the lane turns its counts into the five products, the wave sums
// count[n]*product over its 64 individuals (fixed shuffle order) and lane 0 stores the
// per-chunk partial; k_ld_finalize adds the chunks in ascending order and divides.
// ---------------------------------------------------------------------------
// t32/segs/wconst are passed as __restrict__ kernel arguments as well as inside `a`: only
// then does hipcc know they cannot alias the stores to a.partial and keep the wave-uniform
// loads (segment records, the target's haplotype words) on the scalar path.
template <int KP>
__global__ __launch_bounds__(512) void k_ld_popcount(const uint2 *__restrict__ t32,
                                                     const Seg *__restrict__ segs,
                                                     const WinConst *__restrict__ wconst,
                                                     const WinTarget *__restrict__ wtarget,
                                                     PopArgs a)
{
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lane = threadIdx.x & 63;
    const unsigned c = blockIdx.y * 8 + wave;
    if (c >= a.n_chunks)
        return;
    const unsigned t = blockIdx.z;
    const uint32_t w0 = blockIdx.x * a.win_per_group;
    const uint32_t w1 = min(w0 + a.win_per_group, a.n_win);
    const uint32_t seg0 = wconst[w0].seg_begin, seg1 = wconst[w1].seg_begin;
    if (seg0 >= seg1)
        return;
    const uint32_t tgt = a.targets[t];
    const uint2 *xt = t32 + (size_t)c * a.n_tiles * 64 + lane;
    const uint2 *tt = t32 + (size_t)(tgt >> 6) * a.n_tiles * 64 + (tgt & 63);
    const double wgt = a.weight[(size_t)t * a.lanes + c * 64 + lane];

    uint32_t c0[KP], c1[KP], ch[KP], g00[KP], g01[KP], g10[KP], g11[KP], A0[KP], A1[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k)
        c0[k] = c1[k] = ch[k] = g00[k] = g01[k] = g10[k] = g11[k] = A0[k] = A1[k] = 0;

    // Software pipeline: segment records two ahead (scalar), tile words and the target's
    // words one ahead, so no load is waited for in the iteration that issued it.
    SegRegs r1, r2;
    r1.load(segs + seg0);
    r2.load(segs + min(seg0 + 1, seg1 - 1));
    uint2 xn = xt[(size_t)r1.tile() * 64];
    uint2 atn = tt[(size_t)r1.tile() * 64];
    for (uint32_t s = seg0; s < seg1; ++s) {
        const SegRegs S = r1;
        const uint2 x = xn;
        const uint2 at = atn;                              // the target's two haplotypes (uniform)
        r1 = r2;
        r2.load(segs + min(s + 2, seg1 - 1));
        xn = xt[(size_t)r1.tile() * 64];
        atn = tt[(size_t)r1.tile() * 64];
        const uint32_t hom = x.x & x.y;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const uint32_t cov = S.cov(k);
            if (cov) {
                const uint32_t m0 = at.x & cov, m1 = at.y & cov;
                c0[k] += __popc(x.x & cov);
                c1[k] += __popc(x.y & cov);
                ch[k] += __popc(hom & cov);
                g00[k] += __popc(x.x & m0);
                g01[k] += __popc(x.y & m0);
                g10[k] += __popc(x.x & m1);
                g11[k] += __popc(x.y & m1);
            }
        }
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const uint32_t alt = S.alt(k);
            if (alt) {
                A0[k] += __popc(x.x & alt);
                A1[k] += __popc(x.y & alt);
            }
        }
        if (S.last()) {
            const uint32_t w = S.win();
            const WinConst wc = wconst[w];
            const WinTarget wt = wtarget[(size_t)t * a.n_win + w];
            const uint32_t C0 = planes_sum<KP>(c0), C1 = planes_sum<KP>(c1), CH = planes_sum<KP>(ch);
            const uint32_t G00 = planes_sum<KP>(g00), G01 = planes_sum<KP>(g01);
            const uint32_t G10 = planes_sum<KP>(g10), G11 = planes_sum<KP>(g11);
            const uint32_t a0 = planes_sum<KP>(A0), a1 = planes_sum<KP>(A1);
            const uint32_t CT = wc.cov_total, AT = wc.alt_total;
            // pDg[x0+x1]   (sum_ibd2_ref, src/ibdgem.c:715)
            uint32_t E3 = C0 + C1 - 2 * CH;
            uint32_t E2 = AT - a0 - a1 + CH;
            const double P2 = ld_value(a, wc.mK, wc.eK, CT - E2 - E3, E2, E3);
            // pDg[A0+h0], pDg[A0+h1], pDg[A1+h0], pDg[A1+h1]   (sum_ibd1_ref, :716-719)
            E3 = wt.a0cov + C0 - 2 * G00; E2 = AT - wt.a0alt - a0 + G00;
            const double Q00 = ld_value(a, wc.mK, wc.eK, CT - E2 - E3, E2, E3);
            E3 = wt.a0cov + C1 - 2 * G01; E2 = AT - wt.a0alt - a1 + G01;
            const double Q01 = ld_value(a, wc.mK, wc.eK, CT - E2 - E3, E2, E3);
            E3 = wt.a1cov + C0 - 2 * G10; E2 = AT - wt.a1alt - a0 + G10;
            const double Q10 = ld_value(a, wc.mK, wc.eK, CT - E2 - E3, E2, E3);
            E3 = wt.a1cov + C1 - 2 * G11; E2 = AT - wt.a1alt - a1 + G11;
            const double Q11 = ld_value(a, wc.mK, wc.eK, CT - E2 - E3, E2, E3);

            double s0 = wgt * P2;                                   // :743
            double s1 = wgt * (((Q00 + Q01) + Q10) + Q11);          // :744-745
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                s0 += __shfl_xor(s0, off);
                s1 += __shfl_xor(s1, off);
            }
            if (lane == 0) {
                double *o = a.partial + (((size_t)t * a.n_win + w) * a.n_chunks + c) * 2;
                o[0] = s0;
                o[1] = s1;
            }
#pragma unroll
            for (int k = 0; k < KP; ++k)
                c0[k] = c1[k] = ch[k] = g00[k] = g01[k] = g10[k] = g11[k] = A0[k] = A1[k] = 0;
        }
    }
}

// Sum the per-chunk partials in ascending chunk order and take the background average
// (src/ibdgem.c:751-752).
__global__ __launch_bounds__(256) void k_ld_finalize(PopFinalArgs a)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.n_win)
        return;
    const unsigned t = blockIdx.y;
    const double *p = a.partial + ((size_t)t * a.n_win + w) * a.n_chunks * 2;
    double t0 = 0.0, t1 = 0.0;
    for (uint32_t c = 0; c < a.n_chunks; ++c) {
        t0 += p[2 * c];
        t1 += p[2 * c + 1];
    }
    const int nref = a.n_refpanel[t];
    double *o = a.win_ll + ((size_t)t * a.n_win + w) * 3;
    o[0] = t0 / (double)nref;
    o[1] = t1 / (double)(nref * 4);
}

// ---------------------------------------------------------------------------
void launch_transpose32(const uint64_t *panel, uint32_t stride, size_t n_rows, uint32_t n_chunks,
                        uint32_t n_tiles, uint32_t *t32, hipStream_t st)
{
    if (n_tiles == 0)
        return;
    hipLaunchKernelGGL(k_transpose32, dim3(n_tiles, (n_chunks + 3) / 4), dim3(256), 0, st, panel, stride,
                       n_rows, n_chunks, n_tiles, t32);
}

void launch_win_target(const PopArgs &a, unsigned n_targets, hipStream_t st)
{
    if (a.n_win == 0)
        return;
    hipLaunchKernelGGL(k_win_target, dim3((a.n_win + 255) / 256, n_targets), dim3(256), 0, st, a,
                       const_cast<WinTarget *>(a.wtarget));
}

int launch_ld_popcount(const PopArgs &a, unsigned n_targets, int planes, hipStream_t st)
{
    if (a.n_win == 0)
        return 0;
    dim3 grid((a.n_win + a.win_per_group - 1) / a.win_per_group, (a.n_chunks + 7) / 8, n_targets);
    dim3 block(512);
    switch (planes) {
    case 1: case 2: case 3:
        hipLaunchKernelGGL(k_ld_popcount<3>, grid, block, 0, st, (const uint2 *)a.t32, a.segs, a.wconst, a.wtarget, a); break;
    case 4: case 5:
        hipLaunchKernelGGL(k_ld_popcount<5>, grid, block, 0, st, (const uint2 *)a.t32, a.segs, a.wconst, a.wtarget, a); break;
    case 6: case 7:
        hipLaunchKernelGGL(k_ld_popcount<7>, grid, block, 0, st, (const uint2 *)a.t32, a.segs, a.wconst, a.wtarget, a); break;
    default: return 1;
    }
    return 0;
}

void launch_ld_finalize(const PopFinalArgs &a, unsigned n_targets, hipStream_t st)
{
    if (a.n_win == 0)
        return;
    hipLaunchKernelGGL(k_ld_finalize, dim3((a.n_win + 255) / 256, n_targets), dim3(256), 0, st, a);
}

}  // namespace ibdg
