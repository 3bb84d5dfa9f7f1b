// ibdg_kernels.hip -- gfx950 (MI355X) kernels of the IBD-likelihood engine.
//
// Every kernel cites the reference lines whose result it reproduces
// (paths relative to /root/reference).  Arithmetic is IEEE fp64 mul/add in the
// reference's association order; the file is compiled with -ffp-contract=off
// because the reference (x86-64, -O0) never fuses.  No pow() on the device: all
// powers come from host-built tables (glibc pow, like the reference).
//
// Data layout:
//   panel   [n_rows][stride] u64, word 2c = first haplotypes of individuals
//           64c..64c+63, word 2c+1 = their second haplotypes (bit = n%64)
//   lut     [(M+1)^2][3] f64 = P(D|G) for G=00,01,11, index n_ref*(M+1)+n_alt
//   rec     {row_index u32, lut byte offset u32}; offset 0 <=> zero coverage
#include "ibdg_kernels.h"
#include "ibdg_ld_dev.h"

#include <hip/hip_runtime.h>

namespace ibdg {

// ---------------------------------------------------------------------------
// K0: alt-allele count of every panel row = popcount of the packed row.
// Replaces find_f_impute / find_f_vcf (src/ibd-parse.c:91-110): the count over
// ALL 2*n_ids alleles of the row; the division happens where f is used.
//
// The panel is read as one flat stream of 16-byte units, 64 consecutive units (1 KiB) per wave
// load, whatever the row length: with `pairs` units per row, R = 64/gcd(pairs,64) rows make a whole
// number of loads (2504 individuals: 40 units per row, 8 rows = 5 loads), so no lane ever idles
// (a wave per row left 24 of 64 lanes without work on 640-byte rows: 3.96 TB/s).  Every lane
// drops the popcount of its unit into the wave's LDS strip; then gcd lanes per row add the row's
// units up (strided reads, xor-shuffle among the gcd neighbours).
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned popc16(const uint4 v)
{
    return __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
}

// LOADS > 0: the number of 64-unit loads per group, known at compile time (all of them in flight at
// once); LOADS == 0: any number (one at a time, hidden by the other waves of the CU).
template <int LOADS>
__global__ __launch_bounds__(64) void k_alt_count(const uint4 *__restrict__ panel16, uint32_t pairs, uint32_t g,
                                                  uint32_t loads, size_t n_rows, size_t n_groups,
                                                  uint32_t *__restrict__ alt_count)
{
    extern __shared__ uint32_t cnt[];                 // loads * 64 unit counts
    const unsigned lane = threadIdx.x;
    const uint32_t R = 64 / g;                        // rows per group
    const size_t unit_end = n_rows * pairs;
    const unsigned r = lane / g, part = lane % g;
    for (size_t grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        const size_t unit0 = grp * (size_t)loads * 64;
        if (LOADS > 0 && unit0 + (size_t)LOADS * 64 <= unit_end) {     // a whole group: plain loads, nothing masked
            uint4 v[LOADS > 0 ? LOADS : 1];
#pragma unroll
            for (int i = 0; i < LOADS; ++i)
                v[i] = panel16[unit0 + (size_t)i * 64 + lane];
#pragma unroll
            for (int i = 0; i < LOADS; ++i)
                cnt[i * 64 + lane] = popc16(v[i]);
        } else {
            for (uint32_t i = 0; i < loads; ++i) {
                const size_t u = unit0 + (size_t)i * 64 + lane;
                const uint4 v = panel16[u < unit_end ? u : unit_end - 1];      // clamped, never masked
                cnt[i * 64 + lane] = u < unit_end ? popc16(v) : 0u;
            }
        }
        __syncthreads();                              // one wave per block: orders its LDS writes and reads
        unsigned c = 0;
        for (uint32_t k = part; k < pairs; k += g)
            c += cnt[r * pairs + k];
        for (uint32_t m = 1; m < g; m <<= 1)
            c += __shfl_xor(c, m);
        const size_t row = grp * R + r;
        if (part == 0 && row < n_rows)
            alt_count[row] = c;
        __syncthreads();
    }
}

// Rows too long for the LDS strip (pairs * 64 / gcd counts): a wave per row, 16 B per lane and turn.
__global__ __launch_bounds__(256) void k_alt_count_long(const uint64_t *__restrict__ panel,
                                                        uint32_t stride, size_t n_rows,
                                                        uint32_t *__restrict__ alt_count)
{
    const unsigned lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * (blockDim.x >> 6);
    const uint32_t pairs = stride >> 1;          // stride is even: 16-byte units per row
    for (size_t r = wave; r < n_rows; r += n_waves) {
        const uint4 *row = reinterpret_cast<const uint4 *>(panel + r * stride);
        unsigned c = 0;
        for (uint32_t i = lane; i < pairs; i += 64)
            c += popc16(row[i]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
            c += __shfl_xor(c, off);
        if (lane == 0)
            alt_count[r] = c;
    }
}

// ---------------------------------------------------------------------------
// K1 + K3 in one launch: the per-row LIBD0/LIBD1/LIBD2 (tab columns) and the window products.
//   f          src/ibd-parse.c:98 (count / (2*n_ids)) or the -A override
//   ibd0       find_pDgf      src/ibd-math.c:84-101
//   ibd1       find_pDgIBD1   src/ibd-math.c:104-142
//   ibd2       src/ibdgem.c:643-651
//   S0,S1,S2   src/ibdgem.c:562, :665-667, :755: products over the covered rows of a window in row
//              order; in --LD mode only LIBD2 = S2 is written (:752), LIBD0/LIBD1 come from the LD kernels.
// pow(1-f,2.0) and pow(f,2.0) come from pow_tab (indexed by alt count) or, with an -A override, from
// the per-site fo array {f, pow(1-f,2), pow(f,2)}.
//
// A wave per (two consecutive windows, comparison individual): for each window its lanes take the rows from the window's first covered
// row up to the next window's (rows without reads in between are printed but join no window,
// src/ibdgem.c:657-663; the rows before the first window go with window 0, those behind the last one
// with it), 128 rows per turn with all their gathers in flight, compute the row's three values, store
// them when per-site results are wanted, and leave the factors -- 1.0 for a row without reads: x * 1.0
// is x -- in the window's LDS strip of the wave, where three lanes per window multiply them up in row order (six side by side).  The per-site triple
// is never read back (a separate product kernel re-read 24 B per row), and a run that wants window
// results only (FULL = false in --LD mode: the host program's --summary-only) computes just the IBD2
// pick of every row: one table look-up, nothing stored.
// ---------------------------------------------------------------------------
// One turn of a window: rows [base, base + 128) below e, two per lane, all their gathers in flight; the rows' values go to
// site_ll (when kept) and their factors for the window products -- 1.0 for a row without reads or beyond e -- into `buf`.
#ifndef IBDG_SITE_NT
#define IBDG_SITE_NT 1
#endif
template <bool FULL>
__device__ __forceinline__ void rows_turn(const RowsArgs &a, unsigned t, uint32_t tgt, size_t base, size_t e, unsigned lane,
                                          double *__restrict__ buf)
{
    const double qnan = __longlong_as_double(0x7ff8000000000000ll);
    uint2 rc[2];
    bool live[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const size_t s = base + 64 * u + lane;
        live[u] = s < e;
        rc[u] = live[u] ? a.rec_all[s] : make_uint2(0, 0);
    }
    uint32_t k[2];
    uint2 tw[2];
    double p00[2], p01[2], p11[2], fo[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const double *L = reinterpret_cast<const double *>(reinterpret_cast<const char *>(a.lut) + rc[u].y);
        p00[u] = L[0];
        p01[u] = L[1];
        p11[u] = L[2];
        k[u] = FULL && live[u] ? a.alt_count[rc[u].x] : 0u;
        if (a.t32) {
            // the target's alleles from the tile-transposed copy: one 8-byte word pair serves 32
            // consecutive rows (the site-major row costs two 64-byte sectors per row for two bits)
            const uint2 *p = reinterpret_cast<const uint2 *>(
                a.t32 + ((size_t)(tgt >> 6) * a.n_pairs + (rc[u].x >> 6)) * 64 + (tgt & 63));
            tw[u] = live[u] ? p[(rc[u].x >> 5) & 1] : make_uint2(0, 0);
        } else {
            const uint64_t *row = a.panel + (size_t)rc[u].x * a.stride;
            const uint64_t r0 = live[u] ? row[2 * (tgt >> 6)] : 0, r1 = live[u] ? row[2 * (tgt >> 6) + 1] : 0;
            tw[u] = make_uint2((uint32_t)((r0 >> (tgt & 63)) & 1u) << (rc[u].x & 31),
                               (uint32_t)((r1 >> (tgt & 63)) & 1u) << (rc[u].x & 31));
        }
        fo[u] = FULL && a.fo && live[u] ? a.fo[3 * (base + 64 * u + lane)] : qnan;
    }
    double f[2], pw1[2], pw2[2];
    if (FULL) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            f[u] = (double)k[u] / (double)(int)(2u * a.n_ids);       // src/ibd-parse.c:98
            pw1[u] = a.pow_tab[2 * k[u]];
            pw2[u] = a.pow_tab[2 * k[u] + 1];
            if (fo[u] == fo[u]) {          // not NaN: -A override (src/ibdgem.c:609-614)
                const size_t s = base + 64 * u + lane;
                f[u] = fo[u];
                pw1[u] = a.fo[3 * s + 1];
                pw2[u] = a.fo[3 * s + 2];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const size_t s = base + 64 * u + lane;
        const unsigned A0 = (tw[u].x >> (rc[u].x & 31)) & 1u, A1 = (tw[u].y >> (rc[u].x & 31)) & 1u;
        const unsigned g = A0 + A1;
        const double ibd2 = g == 0 ? p00[u] : (g == 1 ? p01[u] : p11[u]);
        const bool covered = rc[u].y != 0;            // table offset 0 <=> no reads
        double *o = buf + (64 * u + lane);            // value v of the row at o[128 v]: a chain's factors are neighbours
        if (FULL) {
            const double omf = 1 - f[u];
            double ibd0 = 1.0;
            if (!(p00[u] == 1 || p01[u] == 1 || p11[u] == 1)) {
                const double t1 = pw1[u] * p00[u];
                const double t2 = ((2 * omf) * f[u]) * p01[u];
                const double t3 = pw2[u] * p11[u];
                ibd0 = (t1 + t2) + t3;
                if (ibd0 == 0.0)
                    ibd0 = 2.2250738585072014e-308;      // DBL_MIN
            }
            double ibd1;
            if (g == 0)
                ibd1 = (f[u] * p01[u]) + (omf * p00[u]);
            else if (g == 1)
                ibd1 = ((0.5 * p01[u]) + ((0.5 * omf) * p00[u])) + ((0.5 * f[u]) * p11[u]);
            else
                ibd1 = (omf * p01[u]) + (f[u] * p11[u]);
            if (ibd1 == 0.0)
                ibd1 = 2.2250738585072014e-308;
            if (live[u]) {
                if (a.site_ll) {
                    double *d = a.site_ll + ((size_t)t * a.n_sites + s) * 3;
#if IBDG_SITE_NT
                    // (written once, read by nobody on the device: past the caches, which the --LD kernel's tables and the
                    //  targets' tile words live in)
                    __builtin_nontemporal_store(ibd0, d);
                    __builtin_nontemporal_store(ibd1, d + 1);
                    __builtin_nontemporal_store(ibd2, d + 2);
#else
                    d[0] = ibd0;
                    d[1] = ibd1;
                    d[2] = ibd2;
#endif
                }
            }
            if (!a.ld_mode) {                          // (--LD: only the IBD2 products are taken from here, :752)
                o[0] = covered ? ibd0 : 1.0;
                o[128] = covered ? ibd1 : 1.0;
            }
            o[256] = covered ? ibd2 : 1.0;
        } else {
            o[0] = covered ? ibd2 : 1.0;
        }
    }
}

// WPW consecutive windows per wave: their turns fill WPW strips one after the other, then lanes 0 .. NV*WPW-1 multiply the
// strips up side by side -- an LDS read instruction costs the LDS pipeline the same 8 cycles whether 3 lanes or 64 take part,
// and with one window per wave those reads kept it busy 60 % of the kernel's time (PMC LdsUtil), the busiest unit of all.
#ifndef IBDG_ROWS_WPW
#define IBDG_ROWS_WPW 2
#endif
template <bool FULL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void k_rows_windows(RowsArgs a)
{
    constexpr int NV = FULL ? 3 : 1;                 // values per row kept for the products
    constexpr int WPW = IBDG_ROWS_WPW;
    __shared__ __attribute__((aligned(16))) double strip[4][WPW][128 * NV];
    // (the wave's number as a scalar: the windows' bounds then come through scalar loads and live in SGPRs)
    const unsigned wave = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const unsigned t = blockIdx.y;
    uint32_t tgt = a.targets[t];
    IBDG_CHECK_TGT(tgt, a.n_ids, __func__);
    const uint32_t n_w = a.n_win ? a.n_win : 1;      // no covered row at all: the rows still get their values
    const uint32_t n_groups = (n_w + WPW - 1) / WPW;
    const unsigned cq = lane / NV, cc = lane % NV;   // the chain a lane multiplies up: window cq of the group, value cc
    for (uint32_t g = blockIdx.x * 4 + wave; g < n_groups; g += gridDim.x * 4) {
        const uint32_t w0 = g * WPW;
        size_t bd[WPW + 1];                          // window w0 + q takes the rows [bd[q], bd[q + 1])
        bd[0] = w0 == 0 ? 0 : a.cov_site[(size_t)w0 * a.window];
        size_t span = 0;
#pragma unroll
        for (int q = 0; q < WPW; ++q) {
            const uint32_t w = w0 + q;
            bd[q + 1] = w >= n_w ? bd[q] : (w + 1 >= a.n_win ? a.n_sites : a.cov_site[(size_t)(w + 1) * a.window]);
            span = bd[q + 1] - bd[q] > span ? bd[q + 1] - bd[q] : span;
        }
        double acc = 1.0;
        for (size_t off = 0; off < span; off += 128) {
            // (the strips' previous turn has been read: a wave's LDS operations execute in order)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < WPW; ++q)            // (a window that has run out of rows gets a strip of 1.0s)
                rows_turn<FULL>(a, t, tgt, bd[q] + off, bd[q + 1], lane, strip[wave][q]);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint32_t n = (uint32_t)(span - off < 128 ? span - off : 128);
            if (lane < NV * WPW && (!FULL || !a.ld_mode || cc == 2)) {
                // two factors per LDS instruction (the pipeline's cost is per instruction): the strip holds a value's
                // factors side by side, and behind an odd count sits a 1.0 or the next row's factor of a longer window
                // of the group -- which n, the longest window's count, covers or this lane's rows_turn filled with 1.0
                const double2 *src = reinterpret_cast<const double2 *>(&strip[wave][cq][cc * 128]);
#pragma unroll 4
                for (uint32_t j = 0; j < (n + 1) / 2; ++j) {
                    const double2 v = src[j];
                    acc *= v.x;
                    acc *= v.y;
                }
            }
        }
        if (lane < NV * WPW && w0 + cq < a.n_win) {
            double *o = a.win_ll + ((size_t)t * a.n_win + w0 + cq) * 3;
            if (!FULL)
                o[2] = acc;
            else if (!a.ld_mode || cc == 2)
                o[cc] = acc;
        }
    }
}

// ---------------------------------------------------------------------------
// K2: the --LD background-panel loop (src/ibdgem.c:669-722) and the window
// average (src/ibdgem.c:736-753).
//
// One workgroup per (window, target).  A wave owns CPW chunks of 64 background
// individuals at a time (one individual per lane) and keeps the reference's
// five running products per individual in registers:
//     P2  = prod pDg[h0+h1]          (sum_ibd2_ref[n])
//     Q00 = prod pDg[A0+h0], Q01 = prod pDg[A0+h1],
//     Q10 = prod pDg[A1+h0], Q11 = prod pDg[A1+h1]   (sum_ibd1_ref[4n..4n+3])
// multiplied site by site in row order starting from 1.0, exactly like the
// reference, so each individual's products are bit-identical to it.
//
// The panel words of a chunk ARE the lane masks: word 2c bit l = first
// haplotype of the lane-l individual.  They arrive through the scalar path
// (s_load) and drive v_cndmask directly (inverse ballot), so picking
// pDg[x+y] costs two v_cndmask_b32 per double and no per-lane bit twiddling.
//
// After the last site each lane adds  count[n] * product  (count = times the
// individual is listed as background, 0 when it is the target or the -N
// sample: src/ibdgem.c:714, :742-750), lanes and waves are reduced in a fixed
// order, and thread 0 divides by n_refpanel resp. 4*n_refpanel.
// The sum order differs from the reference's n-ascending loop (tree instead of
// serial): within ~1e-15 relative, the documented tolerance is 1e-10.
// ---------------------------------------------------------------------------
// One site for CPW chunks, branch-free.  The target's alleles A0,A1 at the site are
// wave-uniform, so the allele-dependent choices are made once per site on the scalar
// side and the per-lane work is twelve v_cndmask_b32 and five v_mul_f64 per chunk:
//   (u0,v0) = pDg[A0+0], pDg[A0+1];  (u1,v1) likewise for A1
//   f00 = h0 ? v0 : u0   f01 = h1 ? v0 : u0   f10 = h0 ? v1 : u1   f11 = h1 ? v1 : u1
//   pDg[h0+h1] = (h1 xor A0) ? Y : f00  with  Y = h0 ? vy : uy  and (uy,vy) the pair of
//   the allele 1-A0 (for a heterozygous target that is (u1,v1), i.e. Y = f10).
// (A four-way switch on (A0,A1) needs fewer selects, but hipcc then copies every
// accumulator twice per site across the arms.)
template <int CPW>
__device__ __forceinline__ void ld_site(const uint64_t *__restrict__ m, unsigned A0, unsigned A1,
                                        double p00, double p01, double p11, double (&P2)[CPW],
                                        double (&Q00)[CPW], double (&Q01)[CPW],
                                        double (&Q10)[CPW], double (&Q11)[CPW])
{
    const double u0 = A0 ? p01 : p00, v0 = A0 ? p11 : p01;
    const double u1 = A1 ? p01 : p00, v1 = A1 ? p11 : p01;
    const double uy = A0 ? p00 : p01, vy = A0 ? p01 : p11;
    const uint64_t flip = 0ull - (uint64_t)A0;   // all ones when A0 == 1 (scalar)
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const uint64_t m0 = m[2 * c], m1 = m[2 * c + 1];
        const bool b0 = __builtin_amdgcn_inverse_ballot_w64(m0);
        const bool b1 = __builtin_amdgcn_inverse_ballot_w64(m1);
        const bool b1f = __builtin_amdgcn_inverse_ballot_w64(m1 ^ flip);
        const double f00 = b0 ? v0 : u0;
        const double f01 = b1 ? v0 : u0;
        const double f10 = b0 ? v1 : u1;
        const double f11 = b1 ? v1 : u1;
        const double y = b0 ? vy : uy;
        P2[c] *= b1f ? y : f00;           // pDg[h0+h1]
        Q00[c] *= f00;
        Q01[c] *= f01;
        Q10[c] *= f10;
        Q11[c] *= f11;
    }
}

template <int CPW>
__global__ __launch_bounds__(512) void k_ld_window(LdArgs a)
{
    __shared__ double red[2][8];
    const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned n_waves = blockDim.x >> 6;
    const unsigned lane = threadIdx.x & 63;
    const unsigned w = blockIdx.x, t = blockIdx.y + a.t_base;

    const uint32_t begin = w * a.window;
    const uint32_t end = min(begin + a.window, a.n_cov);
    uint32_t tgt = a.targets[t];
    IBDG_CHECK_TGT(tgt, (64u * a.n_groups * 8u), __func__);
    const uint32_t tw = 2 * (tgt >> 6), tb = tgt & 63;
    const double *wt = a.weight + (size_t)t * a.n_groups * CPW * 64;

    double s0 = 0.0, s1 = 0.0;
    for (unsigned g = wave; g < a.n_groups; g += n_waves) {
        double P2[CPW], Q00[CPW], Q01[CPW], Q10[CPW], Q11[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c)
            P2[c] = Q00[c] = Q01[c] = Q10[c] = Q11[c] = 1.0;

        for (uint32_t j = begin; j < end; ++j) {
            const uint2 rc = a.rec_cov[j];
            const uint64_t *row = a.panel + (size_t)rc.x * a.stride;
            const double *L = reinterpret_cast<const double *>(
                reinterpret_cast<const char *>(a.lut) + rc.y);
            const double p00 = L[0], p01 = L[1], p11 = L[2];
            const unsigned A0 = __builtin_amdgcn_readfirstlane((unsigned)(row[tw] >> tb) & 1u);
            const unsigned A1 = __builtin_amdgcn_readfirstlane((unsigned)(row[tw + 1] >> tb) & 1u);
            const uint64_t *m = row + 2 * CPW * g;
            ld_site<CPW>(m, A0, A1, p00, p01, p11, P2, Q00, Q01, Q10, Q11);
        }
        if (a.vals) {          // reference-order mode: hand the per-individual values to k_ld_ordered_sum
#pragma unroll
            for (int c = 0; c < CPW; ++c)
                a.vals[(size_t)w * a.n_groups * CPW * 64 + (g * CPW + c) * 64 + lane] =
                    make_double2(P2[c], ((Q00[c] + Q01[c]) + Q10[c]) + Q11[c]);      // src/ibdgem.c:744-745
            continue;
        }
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const double cnt = wt[(g * CPW + c) * 64 + lane];
            s0 += cnt * P2[c];
            s1 += cnt * (((Q00[c] + Q01[c]) + Q10[c]) + Q11[c]);
        }
    }
    if (a.vals)
        return;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        s0 += __shfl_xor(s0, off);
        s1 += __shfl_xor(s1, off);
    }
    if (lane == 0) {
        red[0][wave] = s0;
        red[1][wave] = s1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t0 = 0.0, t1 = 0.0;
        for (unsigned i = 0; i < n_waves; ++i) {
            t0 += red[0][i];
            t1 += red[1][i];
        }
        const int nref = a.n_refpanel[t];
        double *o = a.win_ll + ((size_t)t * a.n_win + w) * 3;
        o[0] = t0 / (double)nref;               // src/ibdgem.c:752
        o[1] = t1 / (double)(nref * 4);
    }
}

// Reference-order sums (src/ibdgem.c:741-750): one thread per window walks the background list in
// the reference's order with two serial double accumulators, skipping the comparison individual
// and the pileup's own sample and counting what is left -- the same additions in the same order
// as the reference, hence the same bits.
__global__ __launch_bounds__(64) void k_ld_ordered_sum(OrdArgs a)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= a.n_win)
        return;
    const double2 *v = a.vals + (size_t)w * a.lanes;
    double s0 = 0.0, s1 = 0.0;
    int n_refpanel = (int)a.n_order;
    for (uint32_t k = 0; k < a.n_order; ++k) {
        const uint32_t n = a.order[k];
        if ((int)n != a.pu_id && n != a.target) {
            const double2 x = v[n];
            s0 += x.x;
            s1 += x.y;
        } else {
            n_refpanel--;
        }
    }
    a.win_ll[(size_t)w * 3] = s0 / n_refpanel;                 // :752
    a.win_ll[(size_t)w * 3 + 1] = s1 / (n_refpanel * 4);
}

void launch_ld_ordered_sum(const OrdArgs &a, hipStream_t st)
{
    if (a.n_win == 0)
        return;
    hipLaunchKernelGGL(k_ld_ordered_sum, dim3((a.n_win + 63) / 64), dim3(64), 0, st, a);
}

// ---------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------
void launch_alt_count(const uint64_t *panel, uint32_t stride, size_t n_rows, uint32_t *alt_count,
                      hipStream_t st, unsigned max_blocks)
{
    if (n_rows == 0)
        return;
    const uint32_t pairs = stride >> 1;
    uint32_t g = 64;                                  // gcd(pairs, 64): the lowest set bit of pairs, at most 64
    while (g > 1 && pairs % g)
        g >>= 1;
    const uint32_t loads = pairs / g;
    const size_t lds = (size_t)loads * 64 * 4;
    if (lds <= 16 * 1024) {
        const size_t R = 64 / g, n_groups = (n_rows + R - 1) / R;
        size_t blocks = n_groups < 256 * 32 * 4 ? n_groups : 256 * 32 * 4;            // a few rounds of 32 waves per CU
        if (max_blocks && blocks > max_blocks)
            blocks = max_blocks;
        auto kern = k_alt_count<0>;
        switch (loads) {
        case 1: kern = k_alt_count<1>; break;
        case 2: kern = k_alt_count<2>; break;
        case 3: kern = k_alt_count<3>; break;
        case 4: kern = k_alt_count<4>; break;
        case 5: kern = k_alt_count<5>; break;
        case 6: kern = k_alt_count<6>; break;
        case 7: kern = k_alt_count<7>; break;
        case 8: kern = k_alt_count<8>; break;
        default: break;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64), (uint32_t)lds, st,
                           reinterpret_cast<const uint4 *>(panel), pairs, g, loads, n_rows, n_groups, alt_count);
        return;
    }
    size_t blocks = (n_rows + 3) / 4;
    if (blocks > 256 * 32)
        blocks = 256 * 32;
    hipLaunchKernelGGL(k_alt_count_long, dim3((unsigned)blocks), dim3(256), 0, st, panel, stride, n_rows,
                       alt_count);
}

// The AF column (tab column 6): the alt-allele fraction of a row's panel row, src/ibd-parse.c:98, or the -A override
// (src/ibdgem.c:609-614).  It does not depend on the comparison individual and the host has it from the alt counts
// anyway, so it is produced when somebody asks (ibdg_get_site_af), not stored by every run.
__global__ __launch_bounds__(256) void k_site_af(RowsArgs a)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.n_sites)
        return;
    double f = (double)a.alt_count[a.rec_all[s].x] / (double)(int)(2u * a.n_ids);
    if (a.fo) {
        const double o = a.fo[3 * s];
        if (o == o)
            f = o;
    }
    a.af[s] = f;
}

// ---------------------------------------------------------------------------
// Several comparison individuals over one site list (--LD runs): a row's three values depend on the comparison
// individual through its genotype at the row only -- LIBD0 not at all (find_pDgf, src/ibd-math.c:84-101), LIBD1 through
// the three branches of find_pDgIBD1 (:104-142), LIBD2 = P(D|G) of the genotype (src/ibdgem.c:643-651).  So a run over T
// individuals keeps ONE table per row, {LIBD0, LIBD1 under genotype 0, 1, 2} (k_row_table, the arithmetic of rows_turn
// operation for operation), and the per-site table of individual t is put together when somebody asks for it
// (k_site_expand, ibdg_get_site_ll): T x rows x 24 bytes of stores and of device memory become rows x 32 bytes --
// 5.8 GB per 60 individuals at 4M rows, which cost the matrix-core kernel beside it a sixth of its time.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_row_table(RowsArgs a)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.n_sites)
        return;
    const uint2 rc = a.rec_all[s];
    const double *L = reinterpret_cast<const double *>(reinterpret_cast<const char *>(a.lut) + rc.y);
    const double p00 = L[0], p01 = L[1], p11 = L[2];
    const uint32_t k = a.alt_count[rc.x];
    double f = (double)k / (double)(int)(2u * a.n_ids);       // src/ibd-parse.c:98
    double pw1 = a.pow_tab[2 * k], pw2 = a.pow_tab[2 * k + 1];
    if (a.fo) {
        const double o = a.fo[3 * s];
        if (o == o) {                      // not NaN: -A override (src/ibdgem.c:609-614)
            f = o;
            pw1 = a.fo[3 * s + 1];
            pw2 = a.fo[3 * s + 2];
        }
    }
    const double omf = 1 - f;
    double ibd0 = 1.0;
    if (!(p00 == 1 || p01 == 1 || p11 == 1)) {
        const double t1 = pw1 * p00;
        const double t2 = ((2 * omf) * f) * p01;
        const double t3 = pw2 * p11;
        ibd0 = (t1 + t2) + t3;
        if (ibd0 == 0.0)
            ibd0 = 2.2250738585072014e-308;      // DBL_MIN
    }
    double g0 = (f * p01) + (omf * p00);
    double g1 = ((0.5 * p01) + ((0.5 * omf) * p00)) + ((0.5 * f) * p11);
    double g2 = (omf * p01) + (f * p11);
    if (g0 == 0.0) g0 = 2.2250738585072014e-308;
    if (g1 == 0.0) g1 = 2.2250738585072014e-308;
    if (g2 == 0.0) g2 = 2.2250738585072014e-308;
    double4 *o = reinterpret_cast<double4 *>(a.row_tab) + s;
    *o = make_double4(ibd0, g0, g1, g2);
}

// the per-site table of comparison individual `tgt` from the row table: out[s] = {LIBD0, LIBD1[g], P(D|G = g)}
__global__ __launch_bounds__(256) void k_site_expand(RowsArgs a, uint32_t tgt, double *__restrict__ out)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.n_sites)
        return;
    const uint2 rc = a.rec_all[s];
    unsigned A0, A1;
    if (a.t32) {
        const uint2 *p = reinterpret_cast<const uint2 *>(a.t32 + ((size_t)(tgt >> 6) * a.n_pairs + (rc.x >> 6)) * 64 + (tgt & 63));
        const uint2 tw = p[(rc.x >> 5) & 1];
        A0 = (tw.x >> (rc.x & 31)) & 1u;
        A1 = (tw.y >> (rc.x & 31)) & 1u;
    } else {
        const uint64_t *row = a.panel + (size_t)rc.x * a.stride;
        A0 = (unsigned)((row[2 * (tgt >> 6)] >> (tgt & 63)) & 1u);
        A1 = (unsigned)((row[2 * (tgt >> 6) + 1] >> (tgt & 63)) & 1u);
    }
    const unsigned g = A0 + A1;
    const double4 v = reinterpret_cast<const double4 *>(a.row_tab)[s];
    const double *L = reinterpret_cast<const double *>(reinterpret_cast<const char *>(a.lut) + rc.y);
    double *o = out + 3 * s;
    o[0] = v.x;
    o[1] = g == 0 ? v.y : (g == 1 ? v.z : v.w);
    o[2] = L[g];
}

void launch_row_table(const RowsArgs &a, hipStream_t st)
{
    if (a.n_sites == 0)
        return;
    hipLaunchKernelGGL(k_row_table, dim3((unsigned)((a.n_sites + 255) / 256)), dim3(256), 0, st, a);
}

void launch_site_expand(const RowsArgs &a, uint32_t tgt, double *out, hipStream_t st)
{
    if (a.n_sites == 0)
        return;
    hipLaunchKernelGGL(k_site_expand, dim3((unsigned)((a.n_sites + 255) / 256)), dim3(256), 0, st, a, tgt, out);
}

void launch_site_af(const RowsArgs &a, hipStream_t st)
{
    if (a.n_sites == 0)
        return;
    hipLaunchKernelGGL(k_site_af, dim3((unsigned)((a.n_sites + 255) / 256)), dim3(256), 0, st, a);
}

void launch_rows_windows(const RowsArgs &a, unsigned n_targets, hipStream_t st, unsigned max_blocks)
{
    if (a.n_sites == 0 || n_targets == 0)
        return;
    const uint32_t n_w = a.n_win ? a.n_win : 1;
    size_t blocks = (((size_t)n_w + IBDG_ROWS_WPW - 1) / IBDG_ROWS_WPW + 3) / 4;      // a wave per group of windows
    if (max_blocks && blocks > max_blocks)
        blocks = max_blocks;
    dim3 grid((unsigned)blocks, n_targets);
    // everything but LIBD2 of the windows unwanted (--LD, no per-site results): the one-value form
    if (a.ld_mode && !a.site_ll)
        hipLaunchKernelGGL(k_rows_windows<false>, grid, dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL(k_rows_windows<true>, grid, dim3(256), 0, st, a);
}

int launch_ld(const LdArgs &a, unsigned n_targets, int cpw, unsigned waves, hipStream_t st)
{
    if (a.n_win == 0 || n_targets == 0)
        return 0;
    if (waves < 1) waves = 1;
    if (waves > 8) waves = 8;
    if (waves > a.n_groups) waves = a.n_groups;
    dim3 grid(a.n_win, n_targets), block(64 * waves);
    switch (cpw) {
    case 1: hipLaunchKernelGGL(k_ld_window<1>, grid, block, 0, st, a); break;
    case 2: hipLaunchKernelGGL(k_ld_window<2>, grid, block, 0, st, a); break;
    case 3: hipLaunchKernelGGL(k_ld_window<3>, grid, block, 0, st, a); break;
    case 4: hipLaunchKernelGGL(k_ld_window<4>, grid, block, 0, st, a); break;
    case 5: hipLaunchKernelGGL(k_ld_window<5>, grid, block, 0, st, a); break;
    default: return 1;
    }
    return 0;
}

}  // namespace ibdg
