// ibdg_prep.hip -- per-comparison site preparation on the device.
//
// What the reference does row by row inside its window loop -- telling covered rows from
// zero-coverage rows (src/ibdgem.c:657-663), cutting windows of `window` covered rows
// (:562-570, :723-730) and looking up the binomial coefficient of every row (src/ibd-math.c:55)
// -- is done here once per ibdg_upload_sites for all rows at once:
//   stage A  covered-row flags -> exclusive scan -> site records, covered-row list      (k_prep_sites)
//   stage B  (window, 32-row tile) segment starts -> scan -> segment masks              (k_prep_segs)
//            per-window constants (k_prep_win_const), control words of the exponent-counting
//            --LD kernel (k_prep_seg_flags; ibdg_ld_popcount.hip), K' (k_prep_win_kp)
// Both scans are single-pass: a workgroup takes the next tile of 1024 elements by ticket, counts its
// flags, publishes the count, adds up its predecessors' published counts (a wave looks back 64 tiles at
// a time until it meets one whose running total is known: "decoupled look-back"), publishes its own
// running total and scatters -- one read of the input instead of the three launches and two reads of a
// count / scan / scatter triple.  Tickets are handed out in start order, so every tile a workgroup
// waits for belongs to a workgroup that is already running.  What the host needs between the stages
// (PrepInfo: a few words) is written to a host-mapped mirror by whichever workgroup of a stage
// finishes last; the host polls its sequence number instead of queueing a copy and waiting for the stream.
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

// ---- single-pass scan across workgroups -------------------------------------------------------
// One 64-bit word per tile: status (2 bits: 1 = the tile's own count, 2 = the running total up to and
// including the tile), the epoch of the launch (30 bits: words of earlier launches read as "nothing yet",
// so the array is never cleared), the value (32 bits).  Agent-scope atomics: the workgroups of a launch
// sit on eight XCDs with an L2 each.
struct ScanChain {
    unsigned long long *state;      // [tiles]
    uint32_t *ticket;               // next tile to hand out (reset by the last tile)
    uint32_t epoch;                 // 1 .. 2^30-1
};

__device__ __forceinline__ unsigned long long chain_word(uint32_t status, uint32_t epoch, uint32_t value)
{
    return ((unsigned long long)status << 62) | ((unsigned long long)epoch << 32) | value;
}

// The tile this workgroup works on (tickets in start order) -- all threads get the same answer.
__device__ __forceinline__ uint32_t chain_take_tile(const ScanChain &ch)
{
    __shared__ uint32_t s_tile;
    if (threadIdx.x == 0)
        s_tile = atomicAdd(ch.ticket, 1u);
    __syncthreads();
    return s_tile;
}

// count = kept elements of this tile (the same in every thread).  Returns the number of kept elements in all
// earlier tiles.  Block-wide call; wave 0 does the look-back.
__device__ __forceinline__ uint32_t chain_exclusive(const ScanChain &ch, uint32_t tile, uint32_t count)
{
    __shared__ uint32_t s_excl;
    const unsigned lane = threadIdx.x & 63;
    if (threadIdx.x < 64) {
        if (lane == 0)
            __hip_atomic_store(&ch.state[tile], chain_word(tile == 0 ? 2u : 1u, ch.epoch, count), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        uint32_t excl = 0;
        if (tile > 0) {
            long long base = (long long)tile - 1;        // lane l looks at tile base - l
            for (;;) {
                const long long idx = base - (long long)lane;
                unsigned long long st = chain_word(2u, ch.epoch, 0u);        // before the first tile: total 0
                if (idx >= 0)
                    st = __hip_atomic_load(&ch.state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t status = (uint32_t)(st >> 62), ep = (uint32_t)(st >> 32) & 0x3fffffffu;
                const bool ready = status != 0 && ep == ch.epoch;
                if (__ballot(ready) != ~0ull) {
                    __builtin_amdgcn_s_sleep(1);                              // a predecessor has not published yet
                    continue;
                }
                const uint64_t totals = __ballot(status == 2u);
                const unsigned first = totals ? (unsigned)__builtin_ctzll(totals) : 63u;
                uint32_t v = lane <= first ? (uint32_t)st : 0u;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1)
                    v += __shfl_xor(v, off);
                excl += v;
                if (totals)
                    break;
                base -= 64;
            }
            if (lane == 0)
                __hip_atomic_store(&ch.state[tile], chain_word(2u, ch.epoch, excl + count), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0)
            s_excl = excl;
    }
    __syncthreads();
    return s_excl;
}

// Called by every workgroup of a launch when its work is done: true (in all threads) for the one that is last.
// Its reads see everything the others wrote before they called.
__device__ __forceinline__ bool chain_last_done(uint32_t *done, uint32_t n_blocks)
{
    __shared__ uint32_t s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const uint32_t k = atomicAdd(done, 1u);
        s_last = k + 1 == n_blocks;
        if (s_last) {
            *done = 0;                 // for the next launch
            __threadfence();
        }
    }
    __syncthreads();
    return s_last != 0;
}

// The host's copy of PrepInfo: all words, then the sequence number (host-mapped, fine-grained memory).
__device__ __forceinline__ void mirror_info(const PrepInfo *info, PrepInfo *mirror, uint32_t seq)
{
    const volatile uint32_t *src = reinterpret_cast<const volatile uint32_t *>(info);
    volatile uint32_t *dst = reinterpret_cast<volatile uint32_t *>(mirror);
    constexpr int N = (int)(offsetof(PrepInfo, seq) / 4);
    for (int i = 0; i < N; ++i)
        dst[i] = __hip_atomic_load(const_cast<const uint32_t *>(&src[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    mirror->seq = seq;
    __threadfence_system();
}

// ---- stage A: rows -> records ----------------------------------------------------------------
struct SiteIn {
    const uint32_t *row_index;      // NULL: row s of the panel is site s
    const uint8_t *n_ref, *n_alt;
    size_t n_sites;
    size_t n_rows;
    uint32_t max_cov;
};

struct SiteOut {
    uint2 *rec_all;         // [n_sites] {row, lut byte offset}
    uint2 *rec_cov;         // [n_cov] the same for covered rows
    uint32_t *cov_site;     // [n_cov] site index of every covered row
};

// A tile = PREP_BLOCK consecutive sites; thread t looks at sites base + i*256 + t, so every access is coalesced.
// Besides its tile's records the workgroup clears its share of the segment array (the masks of stage B are
// built with atomicOr) and takes part in the validation: the smallest offending site of either kind.
__global__ __launch_bounds__(PREP_THREADS) void k_prep_sites(SiteIn in, SiteOut out, ScanChain ch, uint32_t n_tiles,
                                                            PrepInfo *__restrict__ info, PrepCtl *__restrict__ ctl,
                                                            PrepInfo *__restrict__ mirror, uint32_t seq,
                                                            uint4 *__restrict__ clear, size_t clear_units)
{
    constexpr int NW = PREP_THREADS / 64;
    __shared__ uint32_t pre[PREP_ITEMS * NW];
    const uint32_t tile = chain_take_tile(ch);
    const size_t base = (size_t)tile * PREP_BLOCK;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t d = in.max_cov + 1;
    uint64_t ballots[PREP_ITEMS];
    uint2 rec[PREP_ITEMS];
    uint32_t err_row = 0xffffffffu, err_cov = 0xffffffffu, row_min = 0xffffffffu, row_max = 0;
#pragma unroll
    for (int i = 0; i < PREP_ITEMS; ++i) {
        const size_t s = base + (size_t)i * PREP_THREADS + threadIdx.x;
        bool f = false;
        rec[i] = make_uint2(0, 0);
        if (s < in.n_sites) {
            const unsigned r = in.n_ref[s], a = in.n_alt[s];
            rec[i].x = in.row_index ? in.row_index[s] : (uint32_t)s;
            rec[i].y = r + a <= in.max_cov ? (r * d + a) * 24u : 0u;
            if (r + a > in.max_cov)
                err_cov = err_cov < (uint32_t)s ? err_cov : (uint32_t)s;
            if (in.row_index && rec[i].x >= in.n_rows)
                err_row = err_row < (uint32_t)s ? err_row : (uint32_t)s;
            out.rec_all[s] = rec[i];           // every row gets its record (an out-of-range pair is reported and never used)
            f = r + a >= 1;
            if (f) {
                row_min = row_min < rec[i].x ? row_min : rec[i].x;
                row_max = row_max > rec[i].x ? row_max : rec[i].x;
            }
        }
        ballots[i] = __ballot(f);
        if (lane == 0)
            pre[i * NW + wave] = (uint32_t)__popcll(ballots[i]);
    }
    __syncthreads();
    uint32_t count = 0;
    if (wave == 0) {                                   // at most 64 entries: one wave scans them
        static_assert(PREP_ITEMS * NW <= 64, "one wave scans the per-(row, wave) totals");
        const uint32_t v = lane < PREP_ITEMS * NW ? pre[lane] : 0u;
        const uint32_t x = wave_excl_scan(v, &count);
        if (lane < PREP_ITEMS * NW)
            pre[lane] = x;
    }
    __shared__ uint32_t s_count;
    if (threadIdx.x == 0)
        s_count = count;
    __syncthreads();
    count = s_count;
    const uint32_t block_off = chain_exclusive(ch, tile, count);
#pragma unroll
    for (int i = 0; i < PREP_ITEMS; ++i) {
        const size_t s = base + (size_t)i * PREP_THREADS + threadIdx.x;
        if (s < in.n_sites && ((ballots[i] >> lane) & 1)) {
            const uint32_t j = block_off + pre[i * NW + wave] + (uint32_t)__popcll(ballots[i] & ((1ull << lane) - 1));
            out.rec_cov[j] = rec[i];
            out.cov_site[j] = (uint32_t)s;
        }
    }
    // the wave's extremes, one atomic each per wave that has something to say
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t a = __shfl_xor(err_row, off), b = __shfl_xor(err_cov, off), m = __shfl_xor(row_min, off),
                       x = __shfl_xor(row_max, off);
        err_row = err_row < a ? err_row : a;
        err_cov = err_cov < b ? err_cov : b;
        row_min = row_min < m ? row_min : m;
        row_max = row_max > x ? row_max : x;
    }
    if (lane == 0) {
        if (err_row != 0xffffffffu) atomicMin(&info->err_row_site, err_row);
        if (err_cov != 0xffffffffu) atomicMin(&info->err_cov_site, err_cov);
        if (row_min != 0xffffffffu) {
            atomicMin(&info->first_row, row_min);
            atomicMax(&info->last_row, row_max);
        }
    }
    if (tile + 1 == n_tiles && threadIdx.x == 0) {
        info->n_cov = block_off + count;
        *ch.ticket = 0;                                // every ticket of this launch has been taken
    }
    // this workgroup's share of the segment array
    for (size_t u = (size_t)tile * PREP_THREADS + threadIdx.x; u < clear_units; u += (size_t)n_tiles * PREP_THREADS)
        clear[u] = make_uint4(0, 0, 0, 0);
    if (chain_last_done(&ctl->done_a, n_tiles) && threadIdx.x == 0) {
        mirror_info(info, mirror, seq);
        // what the host has now is cleared for the next upload; so is what stage B accumulates into
        info->err_row_site = info->err_cov_site = 0xffffffffu;
        info->first_row = 0xffffffffu;
        info->last_row = 0;
        info->out_of_order = info->n_segs = info->ct_max = info->max_seg = info->adv_overflow = 0;
    }
}

// ---- stage B: covered rows -> segments --------------------------------------------------------
// Covered row j starts a segment when it starts a window (j % window == 0) or lies in another
// 32-row tile than row j-1.
struct SegIn {
    const uint2 *rec_cov;
    uint32_t n_cov;
    uint32_t window;
    uint32_t d;             // max_cov + 1
};

__device__ __forceinline__ bool seg_start(const SegIn &in, size_t j)
{
    if (j % in.window == 0)
        return true;
    return (in.rec_cov[j].x >> 5) != (in.rec_cov[j - 1].x >> 5);
}

// Segment masks, row-parallel: every covered row is a lane.  The lanes of a wave (64 consecutive covered
// rows) OR their bits into a wave-private LDS table of [piece][weight plane] words -- a piece = the rows of
// one segment that fall into this wave's 64, so at most 64 pieces -- with ds_or (LDS atomics; a segment's
// ~20 rows hit the same word, which the LDS serialises in a few tens of cycles), then one lane per piece
// adds the piece's non-zero words to the segment in memory with atomicOr: a segment cut by a wave boundary
// is simply two pieces.  segs[] is zeroed beforehand.  The first row of a segment writes tile and window,
// its last row the end-of-window mark.
__global__ __launch_bounds__(PREP_THREADS) void k_prep_segs(SegIn in, ScanChain ch, uint32_t n_tiles,
                                                            Seg *__restrict__ segs, uint32_t seg_cap,
                                                            WinConst *__restrict__ wconst, uint32_t n_win,
                                                            PrepInfo *__restrict__ info)
{
    constexpr int NW = PREP_THREADS / 64;
    __shared__ uint32_t pre[PREP_ITEMS * NW];
    __shared__ uint32_t s_count;
    __shared__ __attribute__((aligned(16))) uint32_t tbl[NW][64][16];
    const uint32_t tile = chain_take_tile(ch);
    const size_t base = (size_t)tile * PREP_BLOCK;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // pass 1: segment starts per (item, wave) -> exclusive prefix; the tile's total goes down the chain
    bool disorder = false;
#pragma unroll 4
    for (int i = 0; i < PREP_ITEMS; ++i) {
        const size_t j = base + (size_t)i * PREP_THREADS + threadIdx.x;
        if (j > 0 && j < in.n_cov && in.rec_cov[j].x <= in.rec_cov[j - 1].x)
            disorder = true;                           // not in file order: only the strict kernel applies
        const uint64_t b = __ballot(j < in.n_cov && seg_start(in, j));
        if (lane == 0)
            pre[i * NW + wave] = (uint32_t)__popcll(b);
    }
    if (__any(disorder) && lane == 0)
        info->out_of_order = 1;
    __syncthreads();
    if (wave == 0) {
        uint32_t tot;
        const uint32_t v = lane < PREP_ITEMS * NW ? pre[lane] : 0u;
        const uint32_t x = wave_excl_scan(v, &tot);
        if (lane < PREP_ITEMS * NW)
            pre[lane] = x;
        if (lane == 0)
            s_count = tot;
    }
    __syncthreads();
    const uint32_t count = s_count;
    const uint32_t boff = chain_exclusive(ch, tile, count);
    if (tile + 1 == n_tiles && threadIdx.x == 0) {
        info->n_segs = boff + count;
        wconst[n_win].mK = 0.0;                        // the entry behind the last window carries seg_begin = n_segs
        wconst[n_win].eK = 0;
        wconst[n_win].cov_total = wconst[n_win].alt_total = 0;
        wconst[n_win].seg_begin = boff + count;
        *ch.ticket = 0;
    }
    uint4 *my_row = reinterpret_cast<uint4 *>(&tbl[wave][lane][0]);
#pragma unroll 1
    for (int i = 0; i < PREP_ITEMS; ++i) {
        const size_t j = base + (size_t)i * PREP_THREADS + threadIdx.x;
        const bool live = j < in.n_cov;
        if (!__any(live))
            break;
        const uint64_t starts = __ballot(live && seg_start(in, j));
        // piece of this row within the wave: pieces are numbered from 0; when lane 0 does not start a
        // segment, piece 0 is the tail of a segment that began in an earlier wave
        const uint32_t lead = (uint32_t)(~starts & 1);                       // 1: there is such a tail
        const uint32_t piece = (uint32_t)__popcll(starts & ((2ull << lane) - 1)) - 1 + lead;
        const uint32_t n_pieces = (uint32_t)__popcll(starts) + lead;
        // global segment of piece q: the segments that start in this wave follow those counted before it
        const uint32_t seg0 = boff + pre[i * NW + wave] - lead;               // segment of piece 0
        const uint32_t seg = seg0 + piece;
        uint32_t row = 0, cv = 0, al = 0;
        if (live) {
            const uint2 rc = in.rec_cov[j];
            const uint32_t idx = rc.y / 24u, r = idx / in.d;
            row = rc.x;
            al = idx - r * in.d;
            cv = r + al;
        }
        const bool is_start = (starts >> lane) & 1;
        if (live && seg < seg_cap) {
            if (is_start) {
                segs[seg].tile = row >> 5;
                segs[seg].win = (uint32_t)(j / in.window);
            }
            // the segment's final row: the next row starts another one
            if (j + 1 >= in.n_cov || seg_start(in, j + 1))
                segs[seg].last = (j + 1 >= in.n_cov || (j + 1) % in.window == 0) ? 1u : 0u;
        }
        if (live && is_start && j % in.window == 0)
            wconst[j / in.window].seg_begin = seg;
        // the wave's table: clear, OR, read back (a wave's LDS operations execute in order)
        my_row[0] = my_row[1] = my_row[2] = my_row[3] = make_uint4(0, 0, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (live) {
            const uint32_t bit = 1u << (row & 31);
            uint32_t *t = &tbl[wave][piece][0];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if ((cv >> k) & 1) atomicOr(&t[k], bit);
                if ((al >> k) & 1) atomicOr(&t[8 + k], bit);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < n_pieces && seg0 + lane < seg_cap) {
            Seg *sg = &segs[seg0 + lane];
            const uint4 *src = reinterpret_cast<const uint4 *>(&tbl[wave][lane][0]);
            const uint4 c0 = src[0], c1 = src[1], a0 = src[2], a1 = src[3];
            const uint32_t w[16] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (w[k])
                    atomicOr(k < 8 ? &sg->cov[k] : &sg->alt[k - 8], w[k]);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
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

// One thread per window: reads, alt reads and K = prod C(cov, n_ref) of its rows in row order
// (src/ibd-math.c:55 factors of every P(D|G) of the window).
__global__ __launch_bounds__(64) void k_prep_win_const(SegIn in, uint32_t n_win, const unsigned long long *__restrict__ nck,
                                                      WinConst *__restrict__ wconst, WinRaw *__restrict__ raw,
                                                      PrepInfo *__restrict__ info)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_win)
        return;
    const uint64_t b = (uint64_t)w * in.window;
    const uint64_t e64 = b + in.window;
    const uint32_t e = e64 < in.n_cov ? (uint32_t)e64 : in.n_cov;
    X87 K;
    K.m = 1ull << 63;                  // 1.0 = 0.5 * 2^1
    K.e = 1;
    uint32_t ct = 0, at = 0;
    // eight rows per turn: their loads (and the coefficient look-ups behind them) are independent of the
    // product chain and go out together -- a thread's rows are 64 consecutive bytes
    for (uint32_t j = (uint32_t)b; j < e; j += 8) {
        uint32_t y[8];
        uint64_t c[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            y[u] = j + u < e ? in.rec_cov[j + u].y : 0u;      // offset 0 = no reads: coefficient 1, counts 0
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint32_t idx = y[u] / 24u, r = idx / in.d, a = idx - r * in.d, cv = r + a;
            ct += cv;
            at += a;
            c[u] = nck[(size_t)cv * in.d + r];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (c[u] > 1) {            // times 1 changes nothing (and c is never 0 for r <= cv)
                const int z = __clzll((long long)c[u]);
                K = x87_mul(K, c[u] << z, 64 - z);
            }
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
__device__ __forceinline__ void seg_flags_one(uint32_t i, Seg *__restrict__ segs, const WinConst *__restrict__ wconst,
                                              const uint32_t *__restrict__ run_begin, uint32_t n_runs, uint32_t ring,
                                              uint32_t seg_cap, PrepInfo *__restrict__ info)
{
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

// The last kernel of stage B: whichever workgroup finishes last hands PrepInfo to the host's mirror and clears what a
// rerun with another run structure accumulates again.
__global__ __launch_bounds__(256) void k_prep_seg_flags(Seg *__restrict__ segs, const WinConst *__restrict__ wconst,
                                                        const uint32_t *__restrict__ run_begin, uint32_t n_runs,
                                                        uint32_t ring, uint32_t seg_cap, PrepInfo *__restrict__ info,
                                                        PrepCtl *__restrict__ ctl, PrepInfo *__restrict__ mirror, uint32_t seq)
{
    seg_flags_one(blockIdx.x * blockDim.x + threadIdx.x, segs, wconst, run_begin, n_runs, ring, seg_cap, info);
    if (chain_last_done(&ctl->done_f, gridDim.x) && threadIdx.x == 0) {
        mirror_info(info, mirror, seq);
        info->max_seg = info->adv_overflow = 0;
    }
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
    ScanChain ch;
    ch.state = a.chain_state;
    ch.ticket = &a.ctl->ticket;
    ch.epoch = a.epoch;
    const unsigned nb = blocks_for(a.n_sites);
    hipLaunchKernelGGL(k_prep_sites, dim3(nb), dim3(PREP_THREADS), 0, st, in, out, ch, nb, a.info, a.ctl, a.mirror, a.seq,
                       reinterpret_cast<uint4 *>(a.clear), a.clear_bytes / 16);
}

void launch_prep_segments(const PrepSegArgs &a, hipStream_t st)
{
    if (a.n_cov == 0)
        return;
    SegIn in;
    in.rec_cov = a.rec_cov;
    in.n_cov = a.n_cov;
    in.window = a.window;
    in.d = a.max_cov + 1;
    ScanChain ch;
    ch.state = a.chain_state;
    ch.ticket = &a.ctl->ticket;
    ch.epoch = a.epoch;
    const unsigned nb = blocks_for(a.n_cov);
    hipLaunchKernelGGL(k_prep_segs, dim3(nb), dim3(PREP_THREADS), 0, st, in, ch, nb, a.segs, a.seg_cap, a.wconst, a.n_win,
                       a.info);
    hipLaunchKernelGGL(k_prep_win_const, dim3((a.n_win + 63) / 64), dim3(64), 0, st, in, a.n_win, a.nck, a.wconst, a.raw,
                       a.info);
}

void launch_prep_seg_flags(const PrepSegArgs &a, const uint32_t *run_begin, uint32_t n_runs, uint32_t ring, uint32_t seq,
                           hipStream_t st)
{
    uint32_t n = a.seg_cap > n_runs ? a.seg_cap : n_runs;
    if (n == 0)
        n = 1;                                         // the mirror is written by this kernel's last workgroup
    hipLaunchKernelGGL(k_prep_seg_flags, dim3((n + 255) / 256), dim3(256), 0, st, a.segs, a.wconst, run_begin, n_runs,
                       ring, a.seg_cap, a.info, a.ctl, a.mirror, seq);
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

}  // namespace ibdg
