// ibdg_prep.hip -- per-comparison site preparation on the device.
//
// What the reference does row by row inside its window loop -- telling covered rows from
// zero-coverage rows (src/ibdgem.c:657-663), cutting windows of `window` covered rows
// (:562-570, :723-730) and looking up the binomial coefficient of every row (src/ibd-math.c:55)
// -- is done here once per ibdg_upload_sites for all rows at once:
//   stage A  covered-row flags -> exclusive scan -> site records, covered-row list
//   stage B  (window, 32-row tile) segment starts -> scan -> segment masks, per-window constants,
//            control words of the exponent-counting --LD kernel (ibdg_ld_popcount.hip)
// Both scans are the plain three-kernel kind (per-block totals, one block scans the totals, the
// blocks redo their flags and scatter): 6-12 bytes per row, a few microseconds per kernel.  (A
// single-pass scan -- tiles chained by decoupled look-back -- was built and measured: 124 us where the three
// kernels of stage A take 27.  The workgroups of a launch sit on eight XCDs, so the chain's words must
// be read with agent-scope loads that go past the L2s, and any counter all workgroups share -- a ticket,
// a "who is last" count -- costs ~12 ns per atomic on its one address: 3907 tiles make that 47 us.)
// What the host needs between the stages (PrepInfo: a few words) is written to a host-mapped mirror by a
// one-wave kernel at the end of each stage; the host polls the mirror's sequence number instead of
// queueing a copy and draining the stream (a round trip of 10-15 us each).
//
// K = prod C(cov, n_ref) over a window is taken in the x87 extended format the host used to
// take it in (64-bit mantissa, round to nearest even after every factor, in row order), done in
// integer arithmetic: the bits of K' = K (1-eps)^reads are the same as a `long double` loop on the
// host would give.
#include "ibdg_kernels.h"

#include <hip/hip_runtime.h>

namespace ibdg {

namespace {

constexpr int PREP_THREADS = 256;
constexpr int PREP_ITEMS = 4;                        // rows of 256 consecutive elements per block: few, so that many
                                                     // waves are in flight -- these kernels wait on memory, and a
                                                     // wave's rows are processed one after the other
constexpr int PREP_BLOCK = PREP_THREADS * PREP_ITEMS;

// exclusive prefix over the 64 lanes of a wave; *total = the wave's sum (all lanes)
__device__ __forceinline__ uint32_t wave_excl_scan(uint32_t v, uint32_t *total)
{
    const unsigned lane = threadIdx.x & 63;
    uint32_t x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off);
        if (lane >= (unsigned)off)
            x += y;
    }
    *total = __shfl(x, 63);
    return x - v;
}

// Set bits of a wave-wide mask below this lane (v_mbcnt_lo / v_mbcnt_hi).  The kernels of this file take a lane's rank
// and its own bit of a ballot this way, and never shift a 64-bit value by the lane number: on gfx950 a 64-bit vector
// shift whose AMOUNT sits in the last VGPR of the wave's allocation reads it wrongly every so often (it takes v0's low
// bits instead; tools/ubench/shift64_top_vgpr.hip: 7 % of executions, never with the amount one register lower), hipcc
// does not work around it for this target, and where the register allocator puts a lane number is not ours to decide.
// That is what made a variant of k_prep_site_scatter with 24 instead of 22 VGPRs scatter some waves' rows to wrong
// ranks in round 3 (docs/DESIGN_rounds_1-4.md s4.4); tools/audit_shift64.py checks every kernel's ISA for the pattern.
__device__ __forceinline__ uint32_t bits_below_lane(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// ---- the block-level part both stages share ---------------------------------------------------
// A block owns PREP_BLOCK consecutive elements; thread t looks at elements base + i*256 + t, so
// every access is coalesced.  flag_of(e) says whether element e is kept.
// count: number of kept elements of the block.
template <class F>
__device__ __forceinline__ uint32_t block_count(size_t base, size_t n, F flag_of)
{
    __shared__ uint32_t wave_cnt[PREP_THREADS / 64];
    uint32_t cnt = 0;
#pragma unroll
    for (int i = 0; i < PREP_ITEMS; ++i) {
        const size_t e = base + (size_t)i * PREP_THREADS + threadIdx.x;
        const bool f = e < n && flag_of(e);
        cnt += (uint32_t)__popcll(__ballot(f));           // the same number in every lane of the wave
    }
    if ((threadIdx.x & 63) == 0)
        wave_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < PREP_THREADS / 64; ++w)
        tot += wave_cnt[w];
    return tot;
}

// scatter: calls emit(e, k) for every kept element e of the block, k = its rank among all kept
// elements (block_off = kept elements before this block), and other(e) for every element.
template <class F, class E>
__device__ __forceinline__ void block_scatter(size_t base, size_t n, uint32_t block_off, F flag_of, E emit)
{
    constexpr int NW = PREP_THREADS / 64;
    __shared__ uint32_t pre[PREP_ITEMS * NW];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t ballots[PREP_ITEMS];
    uint32_t mine = 0;                                 // bit i: this thread's element of row i is kept
#pragma unroll
    for (int i = 0; i < PREP_ITEMS; ++i) {
        const size_t e = base + (size_t)i * PREP_THREADS + threadIdx.x;
        const bool f = e < n && flag_of(e);
        mine |= (uint32_t)f << i;
        ballots[i] = __ballot(f);
        if (lane == 0)
            pre[i * NW + wave] = (uint32_t)__popcll(ballots[i]);
    }
    __syncthreads();
    if (wave == 0) {                                   // at most 64 entries: one wave scans them
        static_assert(PREP_ITEMS * NW <= 64, "one wave scans the per-(row, wave) totals");
        uint32_t tot;
        const uint32_t v = lane < PREP_ITEMS * NW ? pre[lane] : 0u;
        const uint32_t x = wave_excl_scan(v, &tot);
        if (lane < PREP_ITEMS * NW)
            pre[lane] = x;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PREP_ITEMS; ++i) {
        const size_t e = base + (size_t)i * PREP_THREADS + threadIdx.x;
        if ((mine >> i) & 1)
            emit(e, block_off + pre[i * NW + wave] + bits_below_lane(ballots[i]));
    }
}

// ---- stage A: rows -> records ----------------------------------------------------------------
struct SiteIn {
    const uint32_t *row_index;      // NULL: row s of the panel is site s
    const uint8_t *n_ref, *n_alt;
    size_t n_sites;
    size_t n_rows;
    uint32_t max_cov;
};

__global__ __launch_bounds__(PREP_THREADS) void k_prep_site_count(SiteIn in, uint32_t *__restrict__ block_cnt,
                                                                  PrepInfo *__restrict__ info)
{
    const size_t base = (size_t)blockIdx.x * PREP_BLOCK;
    const uint32_t tot = block_count(base, in.n_sites, [&](size_t s) {
        const unsigned r = in.n_ref[s], a = in.n_alt[s];
        if (r + a > in.max_cov)
            atomicMin(&info->err_cov_site, (uint32_t)s);
        if (in.row_index && in.row_index[s] >= in.n_rows)
            atomicMin(&info->err_row_site, (uint32_t)s);
        // the panel rows the sites span (in file order: what the covered rows span at most) -- here, not in the scatter
        // kernel, so that the host has everything it waits for when the scan behind this kernel is done
        if (s == 0)
            info->first_row = in.row_index ? in.row_index[s] : (uint32_t)s;
        if (s + 1 == in.n_sites)
            info->last_row = in.row_index ? in.row_index[s] : (uint32_t)s;
        return r + a >= 1;
    });
    if (threadIdx.x == 0)
        block_cnt[blockIdx.x] = tot;
}

// One block: exclusive scan of cnt[0..n) in place, the grand total to *total.
// stage A (info and mirror given): the host's copy of PrepInfo is written here, by the thread that knows the total --
// everything the host waits for after stage A (the covered rows' number, the first offending site, the rows spanned) is
// known once the counting kernel and this scan are done, so it is told now and decides the layout, builds the run table
// and queues stage B while the scatter kernel still runs (a hand-over kernel of its own behind the scatter kernel cost
// its launch and the host's turn-around on top: 25 us of an upload of 0.16 ms).
__global__ __launch_bounds__(1024) void k_prep_scan(uint32_t *__restrict__ cnt, uint32_t n, uint32_t *__restrict__ total,
                                                   WinConst *__restrict__ behind_last, PrepInfo *__restrict__ info,
                                                   PrepInfo *__restrict__ mirror, uint32_t seq)
{
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? cnt[i] : 0;
        uint32_t wt;
        const uint32_t x = wave_excl_scan(v, &wt);
        if (lane == 0)
            wave_tot[wave] = wt;
        __syncthreads();
        uint32_t off = carry_s;
        for (unsigned w = 0; w < wave; ++w)
            off += wave_tot[w];
        if (i < n)
            cnt[i] = off + x;
        __syncthreads();
        if (threadIdx.x == 1023)
            carry_s = off + x + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *total = carry_s;
        if (behind_last) {             // the entry behind the last window carries seg_begin = n_segs
            behind_last->mK = 0.0;
            behind_last->eK = 0;
            behind_last->cov_total = behind_last->alt_total = 0;
            behind_last->seg_begin = carry_s;
        }
        if (mirror) {                  // the hand-over of stage A (what k_prep_mirror does for stage B)
            const uint32_t *src = reinterpret_cast<const uint32_t *>(info);
            volatile uint32_t *dst = reinterpret_cast<volatile uint32_t *>(mirror);
            constexpr int N = (int)(offsetof(PrepInfo, seq) / 4);
            for (int i = 0; i < N; ++i)
                dst[i] = i == 0 ? carry_s : src[i];           // (word 0 = n_cov = *total, written a moment ago)
            __threadfence_system();
            mirror->seq = seq;
            __threadfence_system();
            info->err_row_site = info->err_cov_site = 0xffffffffu;
            info->first_row = 0xffffffffu;
            info->last_row = 0;
            info->out_of_order = info->n_segs = info->ct_max = info->max_seg = info->adv_overflow = 0;
        }
    }
}

// The host's copy of PrepInfo: all words, then the sequence number (host-mapped, fine-grained memory).  One wave,
// queued behind the last kernel of a stage; stage 0 (after the site records): what the host has now is cleared for
// the next upload, and so is what stage B accumulates into; stage 1 (after the control words): what a rerun with
// another run structure accumulates again.
__global__ __launch_bounds__(64) void k_prep_mirror(PrepInfo *__restrict__ info, PrepInfo *__restrict__ mirror, uint32_t seq,
                                                    int stage)
{
    if (threadIdx.x != 0)
        return;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(info);
    volatile uint32_t *dst = reinterpret_cast<volatile uint32_t *>(mirror);
    constexpr int N = (int)(offsetof(PrepInfo, seq) / 4);
    for (int i = 0; i < N; ++i)
        dst[i] = src[i];
    __threadfence_system();
    mirror->seq = seq;
    __threadfence_system();
    if (stage == 0) {
        info->err_row_site = info->err_cov_site = 0xffffffffu;
        info->first_row = 0xffffffffu;
        info->last_row = 0;
        info->out_of_order = info->n_segs = info->ct_max = info->max_seg = info->adv_overflow = 0;
    } else {
        info->max_seg = info->adv_overflow = 0;
    }
}

struct SiteOut {
    uint2 *rec_all;         // [n_sites] {row, lut byte offset}
    uint2 *rec_cov;         // [n_cov] the same for covered rows
    uint32_t *cov_site;     // [n_cov] site index of every covered row
};

__global__ __launch_bounds__(PREP_THREADS) void k_prep_site_scatter(SiteIn in, const uint32_t *__restrict__ block_off,
                                                                    SiteOut out, PrepInfo *__restrict__ info)
{
    const size_t base = (size_t)blockIdx.x * PREP_BLOCK;
    const uint32_t d = in.max_cov + 1;
    block_scatter(
        base, in.n_sites, block_off[blockIdx.x],
        [&](size_t s) {
            const unsigned r = in.n_ref[s], a = in.n_alt[s];
            // every row gets its record (an out-of-range pair is reported by the host and never used)
            uint2 rc;
            rc.x = in.row_index ? in.row_index[s] : (uint32_t)s;
            rc.y = r + a <= in.max_cov ? (r * d + a) * 24u : 0u;
            out.rec_all[s] = rc;
            return r + a >= 1;
        },
        [&](size_t s, uint32_t j) {
            const unsigned r = in.n_ref[s], a = in.n_alt[s];
            uint2 rc;
            rc.x = in.row_index ? in.row_index[s] : (uint32_t)s;
            rc.y = r + a <= in.max_cov ? (r * d + a) * 24u : 0u;
            out.rec_cov[j] = rc;
            out.cov_site[j] = (uint32_t)s;
        });
}

// ---- stage B: covered rows -> segments --------------------------------------------------------
// Covered row j starts a segment when it starts a window (j % window == 0) or lies in another
// 32-row tile than row j-1.
// With the compacted layout (k_gather_transpose32) the row of covered row j is the VIRTUAL row
// (j / window) * win_rows + j % window: rows without reads do not exist, and the order of the rows in the panel plays no
// part.  win_rows = window rounded up to the layout's alignment: window itself = the rows back to back (dense: v = j,
// windows straddle tiles like on the panel's own rows), 32 * ceil(window / 32) = every window on a tile boundary.
struct SegIn {
    const uint2 *rec_cov;
    uint32_t n_cov;
    uint32_t window;
    uint32_t d;             // max_cov + 1
    uint32_t win_rows;      // 0: the panel's own rows (in-place tiles); otherwise virtual rows per window
};

__device__ __forceinline__ uint32_t seg_row(const SegIn &in, size_t j)
{
    if (in.win_rows == 0)
        return in.rec_cov[j].x;
    const size_t w = j / in.window;
    return (uint32_t)(w * in.win_rows + (j - w * in.window));
}

__device__ __forceinline__ bool seg_start(const SegIn &in, size_t j)
{
    const size_t k = j % in.window;
    if (k == 0)
        return true;
    if (in.win_rows)
        return (seg_row(in, j) & 31) == 0;       // (consecutive virtual rows within a window: a new tile starts at a multiple of 32)
    return (in.rec_cov[j].x >> 5) != (in.rec_cov[j - 1].x >> 5);
}

__global__ __launch_bounds__(PREP_THREADS) void k_prep_seg_count(SegIn in, uint32_t *__restrict__ block_cnt,
                                                                 PrepInfo *__restrict__ info)
{
    const size_t base = (size_t)blockIdx.x * PREP_BLOCK;
    const uint32_t tot = block_count(base, in.n_cov, [&](size_t j) {
        if (in.win_rows == 0 && j > 0 && in.rec_cov[j].x <= in.rec_cov[j - 1].x)
            info->out_of_order = 1;                    // not in file order: only the strict kernel applies
        return seg_start(in, j);
    });
    if (threadIdx.x == 0)
        block_cnt[blockIdx.x] = tot;
}

// Segments in two steps.  (1) k_prep_seg_scatter: every covered row that starts a segment writes its index to
// seg_first[its segment's rank] (the scatter of stage A again, on the start flags); the first row of a window also leaves
// the window's first segment in wconst.  (2) k_prep_seg_walk: a thread per segment walks the segment's rows -- at most
// 32, consecutive in rec_cov, eight loads in flight at a time -- ORs their bits into the sixteen mask words in registers
// and stores the whole 80-byte record.  Nothing is zeroed beforehand and no atomic is involved.  (Round 2/3 built the
// masks row-parallel with LDS and global atomicOr into a zeroed array: 57 us at 4M rows, plus the 13 MB of zeros; a
// thread per segment had been 400 us then with one dependent load per row.)
__global__ __launch_bounds__(PREP_THREADS) void k_prep_seg_scatter(SegIn in, const uint32_t *__restrict__ block_off,
                                                                   uint32_t *__restrict__ seg_first, uint32_t seg_cap,
                                                                   WinConst *__restrict__ wconst)
{
    const size_t base = (size_t)blockIdx.x * PREP_BLOCK;
    block_scatter(
        base, in.n_cov, block_off[blockIdx.x], [&](size_t j) { return seg_start(in, j); },
        [&](size_t j, uint32_t seg) {
            if (seg < seg_cap)
                seg_first[seg] = (uint32_t)j;
            if (j % in.window == 0)
                wconst[j / in.window].seg_begin = seg;
        });
}

// The control word of the --LD loop (ibdg::Seg::flags) for the run structure the host has sent ahead is made here as
// well: the neighbour's tile comes from its first row, the run's bounds from the windows' first segments (written by the
// scatter kernel before this one); thread i < n_runs also reports its run's segment count.  (k_prep_seg_flags does the
// same from the finished records when the host has to try shorter runs.)
__global__ __launch_bounds__(256) void k_prep_seg_walk(SegIn in, const uint32_t *__restrict__ seg_first, Seg *__restrict__ segs,
                                                       uint32_t seg_cap, const WinConst *__restrict__ wconst,
                                                       const uint32_t *__restrict__ run_begin, uint32_t n_runs, uint32_t ring,
                                                       PrepInfo *__restrict__ info)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_segs = info->n_segs;
    if (s < n_runs)
        atomicMax(&info->max_seg, wconst[run_begin[s + 1]].seg_begin - wconst[run_begin[s]].seg_begin);
    if (s >= n_segs || n_segs > seg_cap)       // more segments than room: rows out of order, nothing to build
        return;
    const uint32_t j0 = seg_first[s], j1 = s + 1 < n_segs ? seg_first[s + 1] : in.n_cov;
    uint32_t cov[8] = {0, 0, 0, 0, 0, 0, 0, 0}, alt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t j = j0; j < j1; j += 8) {
        uint2 rc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            rc[u] = j + u < j1 ? in.rec_cov[j + u] : make_uint2(0, 0);          // offset 0 = no reads: no bit anywhere
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t idx = rc[u].y / 24u, r = idx / in.d, al = idx - r * in.d, cv = r + al;
            const uint32_t row = in.win_rows ? seg_row(in, j + u) : rc[u].x;
            const uint32_t bit = 1u << (row & 31);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                cov[k] |= (0u - ((cv >> k) & 1u)) & bit;
                alt[k] |= (0u - ((al >> k) & 1u)) & bit;
            }
        }
    }
    auto tile_of = [&](uint32_t seg) {
        const uint32_t j = seg_first[seg];
        return (in.win_rows ? seg_row(in, j) : in.rec_cov[j].x) >> 5;
    };
    const uint32_t row0 = in.win_rows ? seg_row(in, j0) : in.rec_cov[j0].x;
    const uint32_t tile = row0 >> 5, win = (uint32_t)(j0 / in.window);
    const uint32_t last = (j1 >= in.n_cov || j1 % in.window == 0) ? 1u : 0u;
    uint32_t flags = 0;
    {
        // the run of the segment's window: last r with run_begin[r] <= win
        uint32_t lo = 0, hi = n_runs;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (run_begin[mid] <= win)
                lo = mid;
            else
                hi = mid;
        }
        const uint32_t s0 = wconst[run_begin[lo]].seg_begin, s1 = wconst[run_begin[lo + 1]].seg_begin;
        uint32_t nc = 0, na = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (cov[k]) nc = k + 1;
            if (alt[k]) na = k + 1;
        }
        uint32_t nslot = 0, nhalf = 0, adv = 0;
        if (s + 1 < s1) {
            const uint32_t tn = tile_of(s + 1), qn = tn >> 1;
            adv = qn - (tile >> 1);
            if (adv > 255) {
                info->adv_overflow = 1;    // rows too far apart for the record format (or out of order)
                adv = 255;
            }
            nslot = (qn - (tile_of(s0) >> 1)) % ring;
            nhalf = tn & 1;
        }
        flags = nslot | (nhalf << 3) | (adv << 4) | ((nc > 3 || na > 2) ? 1u << 12 : 0u) | (last ? 1u << 13 : 0u) | (nc << 16) |
                (na << 24);
    }
    uint4 *o = reinterpret_cast<uint4 *>(&segs[s]);
    o[0] = make_uint4(tile, win, last, flags);
    o[1] = make_uint4(cov[0], cov[1], cov[2], cov[3]);
    o[2] = make_uint4(cov[4], cov[5], cov[6], cov[7]);
    o[3] = make_uint4(alt[0], alt[1], alt[2], alt[3]);
    o[4] = make_uint4(alt[4], alt[5], alt[6], alt[7]);
}

// x = m / 2^64 * 2^e with m in [2^63, 2^64): a normalised x87 extended number
struct X87 {
    uint64_t m;
    int32_t e;
};

// a * b rounded to nearest even at 64 bits, renormalised (what fmul + frexpl give on the host)
__device__ __forceinline__ X87 x87_mul(X87 a, uint64_t bm, int32_t be)
{
    uint64_t hi = __umul64hi(a.m, bm), lo = a.m * bm;
    int32_t e = a.e + be;
    if (!(hi >> 63)) {                 // product in [2^126, 2^127): one bit up
        hi = (hi << 1) | (lo >> 63);
        lo <<= 1;
        e -= 1;
    }
    const uint64_t half = 1ull << 63;
    if (lo > half || (lo == half && (hi & 1))) {
        hi += 1;
        if (hi == 0) {                 // carried out of the mantissa
            hi = half;
            e += 1;
        }
    }
    X87 r;
    r.m = hi;
    r.e = e;
    return r;
}

// One thread per window: reads, alt reads and K = prod C(cov, n_ref) of its rows in row order -- a chain of roundings
// (src/ibd-math.c:55 factors of every P(D|G) of the window).  What the thread waits for is memory, not the chain: per
// eight rows one round trip for their records and, behind it, one for their coefficients (58 us at 35 000 windows in
// round 3's form, as long as the three segment kernels beside it).  Now the coefficient table -- (max_cov + 1)^2 entries,
// 7 KB at -M 20 -- is staged into LDS once per workgroup (LDS = true; larger tables stay in memory), and the records of
// the next eight rows are requested before the products of these eight are taken.  Runs on the second stream beside
// the segment kernels, which need none of it.
template <bool LDS>
__global__ __launch_bounds__(128) void k_prep_win_const(SegIn in, uint32_t n_win, const WinRaw *__restrict__ nck,
                                                       WinConst *__restrict__ wconst, WinRaw *__restrict__ raw,
                                                       PrepInfo *__restrict__ info)
{
    extern __shared__ __attribute__((aligned(16))) char smem_wc[];
    const WinRaw *tab = nck;
    if (LDS) {
        uint4 *dst = reinterpret_cast<uint4 *>(smem_wc);
        const uint4 *src = reinterpret_cast<const uint4 *>(nck);
        for (uint32_t i = threadIdx.x; i < in.d * in.d; i += blockDim.x)
            dst[i] = src[i];
        __syncthreads();
        tab = reinterpret_cast<const WinRaw *>(smem_wc);
    }
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_win)                    // (the entry behind the last window, seg_begin = n_segs, is the scan's)
        return;
    const uint64_t b = (uint64_t)w * in.window;
    const uint64_t e64 = b + in.window;
    const uint32_t e = e64 < in.n_cov ? (uint32_t)e64 : in.n_cov;
    X87 K;
    K.m = 1ull << 63;                  // 1.0 = 0.5 * 2^1
    K.e = 1;
    uint32_t ct = 0, at = 0;
    uint32_t y[8], yn[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
        y[u] = (uint32_t)b + u < e ? in.rec_cov[(uint32_t)b + u].y : 0u;      // offset 0 = no reads: coefficient 1, counts 0
    for (uint32_t j = (uint32_t)b; j < e; j += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            yn[u] = j + 8 + u < e ? in.rec_cov[j + 8 + u].y : 0u;             // the next turn's, in flight under this turn's products
        WinRaw c[8];                   // the coefficients come normalised from the host: no 64-bit shift by a count here
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t idx = y[u] / 24u, r = idx / in.d, a = idx - r * in.d, cv = r + a;
            ct += cv;
            at += a;
            c[u] = tab[(size_t)cv * in.d + r];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c[u].e > 1)            // times 1 (= 2^63 / 2^64 x 2^1) changes nothing (and c is never 0 for r <= cv)
                K = x87_mul(K, c[u].m, c[u].e);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            y[u] = yn[u];
    }
    raw[w].m = K.m;
    raw[w].e = K.e;
    wconst[w].cov_total = ct;
    wconst[w].alt_total = at;
    atomicMax(&info->ct_max, ct);
}

// K' = K * (1-eps)^reads as {double mantissa, exponent}: one more x87 product, then the
// conversion to double (round to nearest even at 53 bits).
__global__ __launch_bounds__(256) void k_prep_win_kp(uint32_t n_win, const WinRaw *__restrict__ raw,
                                                     const WinRaw *__restrict__ pow_1me, WinConst *__restrict__ wconst)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_win)
        return;
    X87 K;
    K.m = raw[w].m;
    K.e = raw[w].e;
    const WinRaw B = pow_1me[wconst[w].cov_total];
    const X87 Kp = x87_mul(K, B.m, B.e);
    const uint64_t low = Kp.m & 0x7ffull;
    uint64_t q = Kp.m >> 11;
    if (low > 0x400ull || (low == 0x400ull && (q & 1)))
        q += 1;                        // may reach 2^53: mantissa 1.0, as the host's cast would give
    wconst[w].mK = (double)q * 0x1p-53;
    wconst[w].eK = Kp.e;
}

// One thread per segment: the control word of the --LD kernel (ibdg::Seg::flags), which depends on
// the run structure; one thread per run: the largest number of segments in a run.
__global__ __launch_bounds__(256) void k_prep_seg_flags(Seg *__restrict__ segs, const WinConst *__restrict__ wconst,
                                                        const uint32_t *__restrict__ run_begin, uint32_t n_runs,
                                                        uint32_t ring, uint32_t seg_cap, PrepInfo *__restrict__ info)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t n_segs = info->n_segs;
    if (i < n_runs)
        atomicMax(&info->max_seg, wconst[run_begin[i + 1]].seg_begin - wconst[run_begin[i]].seg_begin);
    if (i >= n_segs || n_segs > seg_cap)       // more segments than room: rows out of order, nothing to prepare
        return;
    Seg &sg = segs[i];
    // the run of the segment's window: last r with run_begin[r] <= win
    uint32_t lo = 0, hi = n_runs;      // run_begin[lo] <= win < run_begin[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (run_begin[mid] <= sg.win)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t s0 = wconst[run_begin[lo]].seg_begin, s1 = wconst[run_begin[lo + 1]].seg_begin;
    const uint32_t q0 = segs[s0].tile >> 1;
    uint32_t nc = 0, na = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (sg.cov[k]) nc = k + 1;
        if (sg.alt[k]) na = k + 1;
    }
    uint32_t nslot = 0, nhalf = 0, adv = 0;
    if (i + 1 < s1) {
        const uint32_t tn = segs[i + 1].tile, qn = tn >> 1;
        adv = qn - (sg.tile >> 1);
        if (adv > 255) {
            info->adv_overflow = 1;    // rows too far apart for the record format: strict kernel
            adv = 255;
        }
        nslot = (qn - q0) % ring;
        nhalf = tn & 1;
    }
    sg.flags = nslot | (nhalf << 3) | (adv << 4) | ((nc > 3 || na > 2) ? 1u << 12 : 0u) | (sg.last ? 1u << 13 : 0u) |
               (nc << 16) | (na << 24);
}

// first / last site of every window (ibdg_get_windows)
__global__ __launch_bounds__(256) void k_prep_win_bounds(const uint32_t *__restrict__ cov_site, uint32_t n_cov,
                                                         uint32_t window, uint32_t n_win, uint32_t *__restrict__ first,
                                                         uint32_t *__restrict__ last)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_win)
        return;
    const uint64_t b = (uint64_t)w * window, e64 = b + window;
    const uint32_t e = e64 < n_cov ? (uint32_t)e64 : n_cov;
    first[w] = cov_site[b];
    last[w] = cov_site[e - 1];
}

unsigned blocks_for(size_t n) { return (unsigned)((n + PREP_BLOCK - 1) / PREP_BLOCK); }

}  // namespace

size_t prep_scan_blocks(size_t n) { return blocks_for(n); }

void launch_prep_sites(const PrepSiteArgs &a, hipStream_t st)
{
    if (a.n_sites == 0)
        return;
    SiteIn in;
    in.row_index = a.row_index;
    in.n_ref = a.n_ref;
    in.n_alt = a.n_alt;
    in.n_sites = a.n_sites;
    in.n_rows = a.n_rows;
    in.max_cov = a.max_cov;
    SiteOut out;
    out.rec_all = a.rec_all;
    out.rec_cov = a.rec_cov;
    out.cov_site = a.cov_site;
    const unsigned nb = blocks_for(a.n_sites);
    hipLaunchKernelGGL(k_prep_site_count, dim3(nb), dim3(PREP_THREADS), 0, st, in, a.block_tmp, a.info);
    hipLaunchKernelGGL(k_prep_scan, dim3(1), dim3(1024), 0, st, a.block_tmp, nb, &a.info->n_cov, (WinConst *)nullptr, a.info,
                       a.mirror, a.seq);
    hipLaunchKernelGGL(k_prep_site_scatter, dim3(nb), dim3(PREP_THREADS), 0, st, in, a.block_tmp, out, a.info);
}

void launch_prep_segments(const PrepSegArgs &a, const uint32_t *run_begin, uint32_t n_runs, uint32_t ring, hipStream_t st,
                          hipStream_t st2)
{
    if (a.n_cov == 0)
        return;
    SegIn in;
    in.rec_cov = a.rec_cov;
    in.n_cov = a.n_cov;
    in.window = a.window;
    in.d = a.max_cov + 1;
    in.win_rows = a.compact;
    const unsigned nb = blocks_for(a.n_cov);
    // the per-window constants on the second stream, beside the three segment kernels
    {
        const size_t tab_bytes = (size_t)in.d * in.d * sizeof(WinRaw);
        const dim3 grid((a.n_win + 127) / 128), block(128);
        if (tab_bytes <= 32 * 1024)
            hipLaunchKernelGGL(k_prep_win_const<true>, grid, block, (uint32_t)tab_bytes, st2, in, a.n_win, a.nck, a.wconst, a.raw,
                               a.info);
        else
            hipLaunchKernelGGL(k_prep_win_const<false>, grid, block, 0, st2, in, a.n_win, a.nck, a.wconst, a.raw, a.info);
    }
    hipLaunchKernelGGL(k_prep_seg_count, dim3(nb), dim3(PREP_THREADS), 0, st, in, a.block_tmp, a.info);
    hipLaunchKernelGGL(k_prep_scan, dim3(1), dim3(1024), 0, st, a.block_tmp, nb, &a.info->n_segs, a.wconst + a.n_win,
                       (PrepInfo *)nullptr, (PrepInfo *)nullptr, 0u);
    hipLaunchKernelGGL(k_prep_seg_scatter, dim3(nb), dim3(PREP_THREADS), 0, st, in, a.block_tmp, a.seg_first, a.seg_cap,
                       a.wconst);
    if (a.seg_cap)
        hipLaunchKernelGGL(k_prep_seg_walk, dim3((a.seg_cap + 255) / 256), dim3(256), 0, st, in, a.seg_first, a.segs, a.seg_cap,
                           a.wconst, run_begin, n_runs, ring, a.info);
}

void launch_prep_seg_flags(const PrepSegArgs &a, const uint32_t *run_begin, uint32_t n_runs, uint32_t ring, uint32_t seq,
                           hipStream_t st, bool redo)
{
    const uint32_t n = a.seg_cap > n_runs ? a.seg_cap : n_runs;
    if (n && redo)
        hipLaunchKernelGGL(k_prep_seg_flags, dim3((n + 255) / 256), dim3(256), 0, st, a.segs, a.wconst, run_begin, n_runs,
                           ring, a.seg_cap, a.info);
    hipLaunchKernelGGL(k_prep_mirror, dim3(1), dim3(64), 0, st, a.info, a.mirror, seq, 1);
}

void launch_prep_win_kp(uint32_t n_win, const WinRaw *raw, const WinRaw *pow_1me, WinConst *wconst, hipStream_t st)
{
    if (n_win == 0)
        return;
    hipLaunchKernelGGL(k_prep_win_kp, dim3((n_win + 255) / 256), dim3(256), 0, st, n_win, raw, pow_1me, wconst);
}

void launch_prep_win_bounds(const uint32_t *cov_site, uint32_t n_cov, uint32_t window, uint32_t n_win, uint32_t *first,
                            uint32_t *last, hipStream_t st)
{
    if (n_win == 0)
        return;
    hipLaunchKernelGGL(k_prep_win_bounds, dim3((n_win + 255) / 256), dim3(256), 0, st, cov_site, n_cov, window, n_win,
                       first, last);
}

// ---------------------------------------------------------------------------
// Per comparison individual: its background multiplicities = the run's (base_w: -B list, -N sample excluded) with its own
// lane zeroed (src/ibdgem.c:714: an individual is no background of itself), and the size of its background
// (:742-750: n_refpanel).  On the device so that a run over NEW comparison individuals queues like any other (the host
// used to build these arrays and wait for their copies: one host wait per comparison individual).
// ---------------------------------------------------------------------------
// A few individuals (<= IBDG_TG_INLINE, the usual run of ONE new individual) arrive as kernel ARGUMENTS and the kernel
// writes the index array as well: no host-to-device copy (a blit kernel and an event of its own) in front of it.
__global__ __launch_bounds__(256) void k_target_weights(const double *__restrict__ base_w, uint32_t *__restrict__ targets,
                                                        TargetsInline inl, uint32_t n_inline, uint32_t lanes, int base_sum,
                                                        double *__restrict__ weight, int *__restrict__ n_refpanel)
{
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t t = blockIdx.y;
    uint32_t tgt = n_inline ? inl.v[t] : targets[t];
#ifdef IBDG_DEBUG_TGT
    if (tgt >= lanes) {
        if (n == 0)
            printf("BAD TARGET %u (lanes %u) in k_target_weights t %u inline %u\n", tgt, lanes, t, n_inline);
        tgt = 0;
    }
#endif
    if (n < lanes)
        weight[(size_t)t * lanes + n] = n == tgt ? 0.0 : base_w[n];
    if (n == 0) {
        n_refpanel[t] = base_sum - (int)base_w[tgt];
        // (the individuals again behind the background sizes: a finalising step left to the next run reads both one run later
        //  than anything else of this run is read, from a ring twice as long as the one `targets` lives in)
        n_refpanel[gridDim.y + t] = (int)tgt;
        if (n_inline)
            targets[t] = tgt;
    }
}

void launch_target_weights(const double *base_w, uint32_t *targets, const uint32_t *inline_targets, uint32_t n_targets,
                           uint32_t lanes, int base_sum, double *weight, int *n_refpanel, hipStream_t st)
{
    if (n_targets == 0)
        return;
    TargetsInline inl = {};
    const uint32_t n_inline = inline_targets && n_targets <= IBDG_TG_INLINE ? n_targets : 0u;
    for (uint32_t t = 0; t < n_inline; ++t)
        inl.v[t] = inline_targets[t];
    hipLaunchKernelGGL(k_target_weights, dim3((lanes + 255) / 256, n_targets), dim3(256), 0, st, base_w, targets, inl, n_inline, lanes,
                       base_sum, weight, n_refpanel);
}

}  // namespace ibdg
