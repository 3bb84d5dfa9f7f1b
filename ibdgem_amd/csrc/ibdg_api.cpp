// ibdg_api.cpp -- the C ABI of include/ibdgem_hip.h: host-side table building,
// validation, device memory and kernel sequencing.  No CPU implementation of
// the likelihood path lives here: without a HIP device every compute entry
// point fails with an error.
#include "../../include/ibdgem_hip.h"
#include "ibdg_kernels.h"

#include <hip/hip_runtime.h>

#include <cerrno>
#include <unistd.h>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// libm pow through a volatile pointer: the compiler must not rewrite
// pow(x, 2.0) as x*x -- glibc's pow differs from x*x in the last bit for some
// inputs and the reference (src/ibd-math.c:93-95) calls the real pow.
double (*volatile libm_pow)(double, double) = std::pow;

std::string g_create_error;          // of the last failed ibdg_create; contexts may be created from several threads
std::mutex g_create_error_mu;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

}  // namespace

struct ibdg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;    // per-site + window-product kernels run beside the --LD kernels
    hipStream_t stream3 = nullptr;    // what a NEW comparison individual needs before its --LD kernel (indices, weights, the
                                      // individual's window / segment images), made under the --LD kernel of the run before
    // Timing events of the last runs (asynchronous runs are timed after the fact).  Every event
    // record is a barrier packet the command processor retires in ~5 us, so the main stream carries
    // one per run (end of the --LD launches) plus a start only when the stream may have been idle.
    static constexpr int EV_RING = 33;      // the last 32 runs can be queried
    struct EvSet {
        hipEvent_t start_own = nullptr;     // recorded when the previous run's end cannot serve as start
        hipEvent_t ld_end = nullptr;        // main stream, after the last --LD launch
        hipEvent_t k_start = nullptr, k_stop = nullptr;   // start / stop of the dominant --LD kernel's dispatch
        bool has_kernel_times = false;
        hipEvent_t s2_start = nullptr;      // stream2: before its first kernel of the run
        hipEvent_t s2[3] = {};              // stream2: after alt-count, per-site, window-product kernels
        hipEvent_t prep = nullptr;          // main stream: the target operands of the matrix-core kernel are built
        hipEvent_t start = nullptr;         // start_own or the previous run's ld_end
        bool recount = false, ld = false;
        bool rows_on_main = false;          // non-LD run: the one kernel went to the main stream, stream2 was not used
    } evs[EV_RING];
    int ev_head = 0;
    long runs_done = 0;
    bool chain_ok = false;      // main stream has been busy since the head run's ld_end was queued
    bool s2_pending = false;    // stream2 holds work the main stream has not waited for yet
    hipEvent_t last_s2 = nullptr;
    std::string err;

    double eps = 0.02;
    unsigned max_cov = 20;
    std::vector<double> lut_h;
    DevBuf lut, pow_tab;

    // panel
    DevBuf panel, alt_count;
    size_t n_rows = 0;
    unsigned n_ids = 0;
    uint32_t n_chunks = 0, stride = 0, n_groups = 0;
    int cpw = 0;
    bool counts_valid = false;

    // sites of the current comparison
    DevBuf rec_all, rec_cov, cov_site, fo;
    DevBuf in_row, in_ref, in_alt;      // device copies of the caller's arrays (ibdg_upload_sites)
    DevBuf scan_tmp, info_dev, wraw, nck_dev, powb, win_first, win_last;
    ibdg::PrepInfo *info_h = nullptr;   // host-mapped mirror of the device's PrepInfo, filled in by the preparation kernels
    uint32_t prep_seq = 0;              // hand-overs so far (info_h->seq == prep_seq: the latest one has arrived)
    size_t seg_room = 0;                // segments the array was cleared for by stage A
    bool have_fo = false;
    size_t n_sites = 0;
    uint32_t n_cov = 0, window = 0, n_win = 0;
    std::vector<uint32_t> win_first_h, win_last_h;   // fetched on the first ibdg_get_windows after an upload
    bool win_bounds_valid = false;
    bool sites_valid = false;           // an upload of sites has succeeded since the last upload of a panel
    std::vector<uint32_t> runs_h;
    // power tables (functions of epsilon only; grown on demand, see grow_pow_tables)
    std::vector<ibdg::PowEntry> p1_h, p2_h, p3_h;      // rho^n, sigma^n, tau^n = (rho / sigma^2)^n (k_ld_mfma)
    std::vector<ibdg::WinRaw> pb_h;
    size_t tab_dev = 0;                 // entries the device copies hold
    size_t tab_fail_from = (size_t)-1;  // first exponent whose power leaves the 32-bit exponent field
    hipEvent_t ev_up[3] = {};           // before the host-to-device copies, after them, after the last prep kernel
    hipEvent_t ev_prep2 = nullptr;      // stream2: the per-window constants of an upload are there
    hipEvent_t ev_prepA = nullptr;      // main stream: the site records of an upload are there (behind k_prep_site_scatter)
    // The finalising step of the last run of single individuals (k_ld_finalize's work) when it has been left to the NEXT
    // run's k_ld_popcount launch (option "finalize_in_next"): whoever reads results or replaces inputs first makes up for
    // it with a launch of its own (flush_finalize).  The partial sums alternate between the two halves of their buffer.
    bool fin_pending = false;
    ibdg::PopFinalArgs fin_args;
    unsigned fin_count = 0;
    uint64_t fin_sites_gen = 0;
    int part_half = 0;
    float up_ms[3] = {0.f, 0.f, 0.f};   // copies, preparation on the device (with its host round trips), whole call
    bool up_ms_pending = false;         // the first two are still to be read from the events

    // fast --LD variant (exponent counting, ibdg_ld_popcount.hip)
    // the compacted tiles of the current site list (k_gather_transpose32) and whether the segments,
    // window constants and control words at hand were cut from them (true) or from the panel's own tiles (false)
    DevBuf t32c;
    uint32_t n_pairs_c = 0;
    bool compact = false;
    uint32_t first_row = 0, last_row = 0;   // panel rows of the first / last site of the upload
    DevBuf seg_first;
    DevBuf t32, segs, wconst, wtarget, twords, wtarget_mt, twords_mt, pow1, pow2, pow3, partial;
    // many comparison individuals (k_ld_mfma): target operands of a batch of groups, window constants per slot,
    // partial sums per half chunk, background multiplicities without the comparison individual's exclusion
    DevBuf aimg, wc_slot, partial_h, base_w;
    // ... and what does NOT depend on the comparison individuals (round 5): every background individual's weighted product of
    // its own genotype factors per window (src/ibdgem.c:715, :743 -- the IBD0 terms) and their sums per chunk, from one pass of
    // k_ld_popcount per site list and background (with some individual's images: the product does not look at them)
    DevBuf p2w, p2c, p2_tw, p2_wt;
    uint64_t p2_gen = 0, p2_bg_gen = 0;     // sites_gen / bg_gen the pass was made for
    int p2_mx = -1;
    uint64_t bg_gen = 1;                    // bumped whenever the background multiplicities change
    // single comparison individuals take their IBD0 terms from that pass too once their runs on one upload and background
    // have added up to "ibd0_after" individuals (the pass costs about one run and saves a fifth of every later one)
    long opt_ibd0_after = 8;                // 0: never
    long opt_mfma_wg_sum = 1;               // the matrix-core kernel's workgroups add their eight waves' sums up themselves (where LDS allows)
    long opt_mfma_batch = 36;               // groups of 15 per launch of the matrix-core kernel (540 individuals)
    size_t dev_mem_bytes = 0;               // the device's memory (hipMemGetInfo at ibdg_create)
    uint64_t ibd0_runs = 0, ibd0_bg_gen = 0;
    int wt_ibd1 = -1;                       // form of the images in wtarget / twords
    DevBuf fragb;                           // [n_segs][3][6 words]: the IBD1 form's fragments that do not depend on the individual (k_frag_base)
    uint64_t fb_gen = 0;                    // sites_gen they were made for
    hipEvent_t ev_fb = nullptr;
    // pow1/pow2: rho^n, sigma^n as {f64 mantissa, i32 exponent}; powb: (1-eps)^n in the x87 format
    uint32_t wpg = 0, max_seg = 0;     // most windows per workgroup run and its largest segment count
    uint32_t n_runs = 0;               // runs of consecutive windows (DevBuf runs: n_runs+1 first windows)
    DevBuf runs;
    int n_cu = 256;
    int tab_in_lds = 0;
    int seg_ring = 4;                  // ring depth the segment control words were built for
    uint32_t n_pairs = 0, n_segs = 0, ct_max = 0;
    int planes = 0;
    bool pop_lut_ok = false;     // P(D|G) table is the unclamped binomial form
    bool pop_sites_ok = false;   // site rows strictly increasing, segments built
    bool pop_dense_enough = true; // the site list went to the layout asked for (false: "compact_tiles" -1 on a sparse pileup)
    std::vector<unsigned long> nck_h;
    int last_variant = 0;
    int last_count_unit = 0;           // 2: the last --LD run's single-individual launches counted on the matrix cores, 1: by (mask, count) pairs, 0: no such launch
    // inputs of the previous ibdg_run whose device copies are still valid
    std::vector<uint32_t> prev_targets;
    std::vector<uint8_t> prev_bg;
    int prev_pu = -2, prev_has_bg = -1;
    size_t prev_lanes = 0;
    // New comparison individuals reach the device without a host wait (the reference's loop hands every individual of the panel
    // to the same rows in turn, src/ibdgem.c:522: a NEW individual per run is the normal case): their indices go through a
    // weights kernel's arguments (a small ring of page-locked slots beyond IBDG_TG_INLINE of them) into one slot of a ring of
    // `targets` buffers (the kernels of earlier runs may still read the others), and the weights / background sizes that follow
    // from them are made on the device (k_target_weights) from the run's background multiplicities `base_w`; `nrefpanel` is a
    // ring as well -- a finalising step left to the next run reads its own run's entry.
    static constexpr int TG_SLOTS = 4;
    uint32_t *tg_stage[TG_SLOTS] = {};
    size_t tg_stage_cap = 0;            // comparison individuals a slot holds
    hipEvent_t tg_stage_ev[TG_SLOTS] = {};
    bool tg_stage_busy[TG_SLOTS] = {};
    int tg_slot = 0;
    // (round 5, later: a RING of four instead of two halves -- the preparation of run i + 1 then waits for the end of run i - 3,
    // not of run i - 1, so it is long done when the --LD kernel of run i ends even on an eighth of a chromosome, where a step is
    // 70 us and the chain "wait, copy, weights, images, record" on stream3 takes 40: profiles/r05_shard_steps.txt)
    static constexpr int TG_RING = 4;
    int tg_cur = 0;                    // the slot of `targets` / `weight` / the images the current comparison individuals sit in
    hipEvent_t tg_s2[TG_RING] = {};     // stream2's last kernel that read that slot
    bool tg_s2_pending[TG_RING] = {};
    hipEvent_t tg_main[TG_RING] = {};   // end of the last run on the main stream that read that slot
    bool tg_main_pending[TG_RING] = {};
    hipEvent_t tg_ready = nullptr;      // stream3: the current comparison individuals' data are complete
    hipEvent_t ev_s3sync = nullptr;     // main stream: the prepared sites stream3's kernels read are complete
    uint64_t s3_gen = 0;                // sites_gen stream3 has been ordered behind
    int nref_slot = 0;                  // `nrefpanel` is a ring of its own, twice as long: a finalising step left to the next run reads
    static constexpr int NREF_SLOTS = 2 * TG_RING;   // its own run's entry one run later than anything else of that run is read
    static constexpr size_t AHEAD_MAX_T = 64;   // runs of up to that many individuals prepare ahead (two halves of every buffer)
    long opt_prep_ahead = 1;
    long opt_end_in_dispatch = 1;    // the end event of a run of single individuals rides in its --LD kernel's dispatch packet (0: an event packet behind it): -7 us of a 91 us step on an eighth of a chromosome, profiles/r05_shard_steps.txt
    int base_sum = 0;                   // sum of base_w
    int wt_slot = -1;                   // the slot the images in wtarget / twords were made in
    bool s3_unsettled = false;          // stream3 holds a preparation the other streams have not been made to wait for yet (an
                                        // ibdg_run that failed half way): the next run settles it before anything else
    // the per-target LDS images of k_win_target (segment records with the target's tile words, window constants) depend
    // on the prepared sites and the targets only: a further run over the same sites and targets reuses them
    uint64_t sites_gen = 0;            // bumped by every upload of sites and every change of layout
    uint64_t relayout_credit = 0;      // what the runs on this upload would have saved on the compacted tiles so far, in
                                       // comparison individuals of the matrix-core kernel (see ibdg_run)
    int wt_mx = -1;                    // ... and the form of the records (option mx_counts)
    uint64_t wt_gen = 0;               // sites_gen the images in wtarget / twords were made for
    uint32_t wt_first = 0, wt_count = 0;   // ... for comparison individuals [wt_first, wt_first + wt_count) of prev_targets

    // run state / results
    DevBuf targets, weight, nrefpanel, af, site_ll, win_ll;
    DevBuf row_tab;                     // [n_sites][4]: the rows' values of an --LD run over several comparison individuals (k_row_table)
    size_t n_targets = 0;
    bool have_results = false;

    // options
    long opt_count_in_run = 0;
    long opt_dispatch_events = 0;   // 1: time the --LD launches through their own dispatch packets (hipExtLaunchKernel);
                                    // gives the dominant kernel's own duration, but costs ~10 us per run more than
                                    // one event record (measured), so it is off unless asked for
    long opt_async = 0;    // 1: ibdg_run returns once its kernels are queued
    long opt_rows_blocks = 0;   // non-LD run: workgroups of k_rows_windows per CU (resident grid, each wave takes several windows); 0 = one wave per pair of windows
    long opt_dev_inputs_ready = 0;   // 1: ibdg_upload_sites_dev trusts the caller that its arrays are complete (no device-wide wait)
    long opt_cpw = 0;      // 0 = auto
    long opt_waves = 8;
    long opt_variant = 0;  // 0 auto, 1 strict products, 2 exponent counting, 3 strict products + serial sums in
                           // the reference's order (bit-identical --LD columns)
    std::vector<uint32_t> bg_order;   // optional: the background list in the reference's order (ibdg_set_background_order)
    DevBuf vals, order;
    long opt_wpg = 16;     // windows per wave in the fast kernel (upper bound unless set explicitly)
    bool opt_wpg_fixed = false;
    long opt_multi_target = 1;   // groups of comparison individuals share a workgroup (k_ld_popcount_mt)
    long opt_mfma_targets = 1;   // 5 or more comparison individuals: groups of IBDG_TG through the matrix cores (k_ld_mfma)
    long opt_mfma_plain_tau = 1; // k_ld_mfma looks tau^G up as a plain double where a window's powers allow it (same bits, half the LDS bytes)
    long opt_mfma_min = 4;       // smallest (last) group worth a launch of its own (round 4: a group of 4 takes 2.09-2.17 ms, four single runs 2.5; a group of 3 2.14 against 1.87 for three single runs since their counts moved to the matrix cores -- 3 until then; of 2: 2.13 against 1.28)
    long opt_guided = 4;   // shrink the runs towards the end of the grid (0 = uniform runs; n scales the
                           // estimate of workgroups in flight by n/4 -- 4 measured best at 500k and 4M rows)
    long opt_ring = 2;     // LDS ring slots per wave (2, 3, 4 or 8); 2 measured fastest (fewest LDS bytes)
    long opt_recbytes = 12 * 1024;   // LDS budget for one run's segment records
    long opt_site_blocks = 4;        // 256-thread workgroups per CU of k_site inside an --LD run (0 = a thread per site)
    long opt_recount_blocks = 4;     // single-wave workgroups per CU of k_alt_count when it runs inside an --LD run
                                     // (0 = the full grid; 4 measured best: tools/recount_sweep.py)
    long opt_compact = 0;            // tiles the --LD kernels read: 0 = chosen per upload (the panel's own where the pileup is
                                     // dense, compacted where it is sparse or the rows are out of file order) and
                                     // per run (many comparison individuals), 1 = always compacted, -1 = never
    long opt_fin_next = 1;           // queued runs of single individuals: a run's finalising step rides in the next run's --LD launch
    long opt_sum_dpp = 1;            // ... its wave sums by DPP moves (0: ds_swizzle, as the vector-ALU form)
    long opt_mx_counts = 1;          // k_ld_popcount: the counts of a haplotype word by one matrix instruction (0: 12 (mask, count) pairs)
    long opt_reserve_compact = 1;    // their buffer is allocated with the panel's (a panel's worth x 1.3 of HBM more per context)
    long opt_compact_align = 1;      // rows a window of the compacted tiles is rounded up to: 1 = the rows back to back (no padding; a
                                     // window straddles tiles like on the panel's own rows), 32 = every window on a tile boundary
                                     // (round 4's layout: 28 % padding at windows of 100 rows)
    long opt_compact_density = 4;    // compacted when fewer than 1 panel row in this many between the first and last site carries reads
                                     // (tools/density_sweep.py: one comparison at 1 row in 3: 0.82 ms in place, 0.94 compacted; in 4: 0.76 / 0.76; in 5: 0.79 / 0.65)
    long opt_compact_targets = 256;  // ... or when the runs on one upload add up to this many comparison individuals of the
                                     // matrix-core kernel k_ld_mfma (the re-layout is paid once: one of them saves 0.007 ms of
                                     // 0.185, the gather costs 1.5; an individual of the counting kernels counts as 16 with
                                     // (mask, count) pairs -- it saves 0.04-0.09 ms of 0.77 -- and as 12 with mx_counts: 0.058 of 0.606)
    long opt_site_results = 1;       // 1: per-site LIBD0/1/2 kept for ibdg_get_site_ll; 0: not -- no T x n_sites x 24 B of HBM,
                                     // no per-site stores (window results only).  (The AF column is made on demand.)
    int res_site_mode = 0;           // the mode the last run's results were produced under
    bool prep_dirty = false;         // the device's PrepInfo may hold the leavings of an upload that did not finish
    long opt_staged_upload = 1;      // panels of 256 MB and more from pageable memory go through the staging team
    // page-locked staging for large panels from pageable memory (staged_upload)
    static constexpr int STAGE_WORKERS = 8;
    static constexpr size_t STAGE_BYTES = (size_t)8 << 20;
    long opt_stage_workers = STAGE_WORKERS;   // host threads of the staging team (two 8 MB page-locked buffers each): a caller with
                                              // several contexts uploading at once gives each a share of the cores
    void *stage[2 * STAGE_WORKERS] = {};
    hipEvent_t stage_ev[2 * STAGE_WORKERS] = {};
};

namespace {

int fail(ibdg_ctx *c, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) {
        c->err = buf;
    } else {
        std::lock_guard<std::mutex> lk(g_create_error_mu);
        g_create_error = buf;
    }
    return 1;
}

#define HIP_TRY(c, call)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((c), "[::] ERROR in %s: %s: %s", __func__, #call, hipGetErrorString(e_)); \
    } while (0)

// The finalising launch a run left to its successor, when no successor took it
int flush_finalize(ibdg_ctx *c)
{
    if (c->fin_pending) {
        c->fin_pending = false;
        ibdg::launch_ld_finalize(c->fin_args, c->fin_count, c->stream, ibdg::KernelEvents());
        HIP_TRY(c, hipGetLastError());
    }
    return 0;
}

// The main stream waits for whatever stream2 still holds (queued, not a host wait).
int join_streams(ibdg_ctx *c)
{
    if (flush_finalize(c)) return 1;
    if (c->s2_pending) {
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->last_s2, 0));
        c->s2_pending = false;
    }
    return 0;
}

// Host wait until both streams are idle (before results are read or inputs replaced).
int quiesce(ibdg_ctx *c)
{
    HIP_TRY(c, hipSetDevice(c->device));
    if (join_streams(c)) return 1;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream3));     // (idle whenever the main stream is: every batch on it ends in an event the main stream waits for)
    c->chain_ok = false;
    return 0;
}

int ensure(ibdg_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes == 0)
        bytes = 16;
    if (b.cap >= bytes)
        return 0;
    if (b.p) {
        // queued runs ("async") may still use the buffer: wait for them rather than rely on hipFree doing so
        if ((c->s2_pending || c->chain_ok) && quiesce(c))
            return 1;
        HIP_TRY(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    HIP_TRY(c, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return 0;
}

void release(DevBuf &b)
{
    if (b.p)
        (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// src/ibd-math.c:5-23: C(i,j) = (i*C(i-1,j-1))/j in unsigned long; 0 for j>i.
std::vector<unsigned long> nck_table(unsigned n)
{
    size_t d = (size_t)n + 1;
    std::vector<unsigned long> t(d * d);
    for (size_t i = 0; i < d; ++i)
        for (size_t j = 0; j < d; ++j)
            t[i * d + j] = j == 0 ? 1ul
                                  : (i == 0 ? 0ul : ((unsigned long)(unsigned)i * t[(i - 1) * d + (j - 1)]) / (unsigned)j);
    return t;
}

// src/ibd-math.c:46-81 over the whole reachable (n_ref, n_alt) grid.
void build_pdg_table(double eps, unsigned M, double *out)
{
    const size_t d = (size_t)M + 1;
    std::vector<unsigned long> nck = nck_table(M);
    for (size_t r = 0; r < d; ++r) {
        for (size_t a = 0; a < d; ++a) {
            double *o = out + (r * d + a) * 3;
            if (r + a > M) {       // never reached: rows with n_ref+n_alt > M are filtered (ibdgem.c:623)
                o[0] = o[1] = o[2] = std::nan("");
                continue;
            }
            if (r == 0 && a == 0) {
                o[0] = o[1] = o[2] = 1.0;
                continue;
            }
            const unsigned long c = nck[(r + a) * d + r];
            const unsigned ur = (unsigned)r, ua = (unsigned)a;
            double p00 = (double)c * libm_pow(1 - eps, ur) * libm_pow(eps, ua);
            double p01 = (double)c * libm_pow(0.5, ur) * libm_pow(0.5, ua);
            double p11 = (double)c * libm_pow(1 - eps, ua) * libm_pow(eps, ur);
            o[0] = p00 == 0.0 ? DBL_MIN : p00;
            o[1] = p01 == 0.0 ? DBL_MIN : p01;
            o[2] = p11 == 0.0 ? DBL_MIN : p11;
        }
    }
}

// x = m * 2^e, m in [0.5,1): extended-precision mantissa so that tables of b^n stay accurate
// to ~1e-19 for any n (plain pow underflows long before n*log2(b) leaves the int range).
struct ME {
    long double m;
    long long e;
};

ME me_norm(long double x, long long e)
{
    int k = 0;
    long double m = frexpl(x, &k);
    return ME{m, e + k};
}

ME me_powl(ME b, uint64_t n)
{
    ME r = me_norm(1.0L, 0);
    while (n) {
        if (n & 1)
            r = me_norm(r.m * b.m, r.e + b.e);
        b = me_norm(b.m * b.m, 2 * b.e);
        n >>= 1;
    }
    return r;
}

ME me_pow(double base, uint64_t n)
{
    ME r = me_norm(1.0L, 0), b = me_norm((long double)base, 0);
    while (n) {
        if (n & 1)
            r = me_norm(r.m * b.m, r.e + b.e);
        b = me_norm(b.m * b.m, 2 * b.e);
        n >>= 1;
    }
    return r;
}

// The fast --LD kernel replaces products of table entries by K*(1-e)^E1*e^E2*2^-E3.  That is
// only the same number when every reachable entry IS C*(1-e)^r*e^a etc. to rounding: no
// DBL_MIN clamp (src/ibd-math.c:77-79), no overflowed coefficient, 0 < e < 1.
bool lut_is_binomial(const std::vector<double> &lut, const std::vector<unsigned long> &nck, double eps,
                     unsigned M)
{
    if (!(eps > 0.0 && eps < 1.0) || M > 50)
        return false;
    const size_t d = (size_t)M + 1;
    const long double b = (long double)(double)(1 - eps), e = (long double)eps;
    for (size_t r = 0; r < d; ++r)
        for (size_t a = 0; a + r <= M; ++a) {
            if (r + a == 0)
                continue;
            const long double c = (long double)nck[(r + a) * d + r];
            const long double want[3] = {c * powl(b, (long double)r) * powl(e, (long double)a),
                                         c * powl(0.5L, (long double)(r + a)),
                                         c * powl(b, (long double)a) * powl(e, (long double)r)};
            for (int g = 0; g < 3; ++g) {
                const long double got = lut[(r * d + a) * 3 + g];
                if (!(want[g] > 1e-280L) || fabsl(got - want[g]) > 1e-14L * want[g])
                    return false;
            }
        }
    return true;
}

int pick_cpw(uint32_t n_chunks, long opt)
{
    if (opt >= 1 && opt <= 5)
        return (int)opt;
    if (n_chunks >= 5)
        return 5;
    return (int)n_chunks;   // 1..4
}

// Lay the panel geometry out for n_ids individuals and allocate device rows.
int prepare_panel(ibdg_ctx *c, size_t n_rows, unsigned n_ids)
{
    if (n_ids == 0)
        return fail(c, "[::] ERROR in ibdg_upload_panel: n_ids must be >= 1");
    c->n_ids = n_ids;
    c->n_rows = n_rows;
    c->n_chunks = (n_ids + 63) / 64;
    c->cpw = pick_cpw(c->n_chunks, c->opt_cpw);
    c->n_groups = (c->n_chunks + c->cpw - 1) / c->cpw;
    c->stride = 2u * c->cpw * c->n_groups;
    c->counts_valid = false;
    c->have_results = false;
    c->pop_sites_ok = false;
    // the device copies of targets / background weights were laid out for the previous panel
    c->prev_targets.clear();
    c->prev_bg.clear();
    c->prev_pu = -2;
    c->prev_has_bg = -1;
    c->prev_lanes = 0;
    c->n_sites = 0;
    c->n_cov = c->n_win = 0;
    c->sites_valid = false;
    c->n_pairs = (uint32_t)(((n_rows + 255) / 256) * 4);     // 64-row tile pairs, padded to whole 8-tile octs
    if (ensure(c, c->panel, n_rows * (size_t)c->stride * 8) || ensure(c, c->alt_count, n_rows * 4))
        return 1;
    if (c->pop_lut_ok && ensure(c, c->t32, (size_t)c->n_chunks * c->n_pairs * 64 * 16))
        return 1;
    // ... and room for the compacted tiles of a site list on all these rows at the usual window sizes (32 ceil(W / 32) / W
    // <= 1.3: W = 100, 50, 75..., 97 and more), so that a re-layout in the middle of a series of runs does not allocate:
    // a hipMalloc of gigabytes is normally 0.2 ms but now and then 270-370 ms (after somebody's large hipFree:
    // tools/hipmalloc_in_process.py), which is 300 runs' worth.  Other windows grow it when their turn comes.
    if (c->pop_lut_ok && c->opt_compact >= 0 && c->opt_reserve_compact &&
        ensure(c, c->t32c, (size_t)c->n_chunks * (size_t)(c->n_pairs * 1.3 + 8) * 64 * 16))
        return 1;
    // pow(1-f,2.0), pow(f,2.0) for every possible alt count (src/ibd-math.c:93-95 with
    // f = k/(2N), src/ibd-parse.c:98)
    const size_t K = 2 * (size_t)n_ids + 1;
    std::vector<double> pt(2 * K);
    for (size_t k = 0; k < K; ++k) {
        const double f = (double)k / (double)(int)(2u * n_ids);
        pt[2 * k] = libm_pow(1 - f, 2.0);
        pt[2 * k + 1] = libm_pow(f, 2.0);
    }
    if (ensure(c, c->pow_tab, pt.size() * 8))
        return 1;
    HIP_TRY(c, hipMemcpyAsync(c->pow_tab.p, pt.data(), pt.size() * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

// A large panel in ordinary (pageable) host memory: a team of host threads copies it piece by piece into
// page-locked staging buffers (two per thread) and every piece goes to the device by DMA as soon as it is
// staged.  The runtime's own path for pageable memory locks the caller's pages chunk by chunk, which is quick
// for memory in huge pages (46 GB/s for a fresh anonymous allocation) and slow for 4 KiB pages -- a mapped
// file from the page cache, the host program's packed-panel cache: 0.18-0.28 s for 2.56 GB.
struct StageJob {
    ibdg_ctx *c;
    const char *src;        // rows in host memory, or
    int fd = -1;            // ... (src == nullptr) in a file, from byte `off` on
    uint64_t off = 0;
    bool io_failed = false;
    size_t rw, dw, n_rows, rows_per_piece;
    int worker, n_workers;
    hipError_t err = hipSuccess;
};

void stage_worker(StageJob *j)
{
    ibdg_ctx *c = j->c;
    if ((j->err = hipSetDevice(c->device)) != hipSuccess)
        return;
    const size_t n_pieces = (j->n_rows + j->rows_per_piece - 1) / j->rows_per_piece;
    int turn = 0;
    for (size_t p = (size_t)j->worker; p < n_pieces; p += (size_t)j->n_workers, turn ^= 1) {
        const int b = 2 * j->worker + turn;
        const size_t r0 = p * j->rows_per_piece, nr = std::min(j->rows_per_piece, j->n_rows - r0);
        if ((j->err = hipEventSynchronize(c->stage_ev[b])) != hipSuccess)     // the buffer's previous piece has left
            return;
        if (j->src) {
            memcpy(c->stage[b], j->src + r0 * j->rw, nr * j->rw);
        } else {
            // straight from the file (the page cache) into the page-locked buffer: no mapping of the file, hence no page
            // faults here and no 2.56 GB of page-table entries to take down when the process ends
            size_t done = 0;
            const size_t want = nr * j->rw;
            while (done < want) {
                const ssize_t k = pread(j->fd, (char *)c->stage[b] + done, want - done, (off_t)(j->off + r0 * j->rw + done));
                if (k <= 0) {
                    if (k < 0 && errno == EINTR)
                        continue;
                    j->io_failed = true;
                    return;
                }
                done += (size_t)k;
            }
        }
        j->err = hipMemcpy2DAsync((char *)c->panel.p + r0 * j->dw, j->dw, c->stage[b], j->rw, j->rw, nr,
                                  hipMemcpyHostToDevice, c->stream);
        if (j->err == hipSuccess)
            j->err = hipEventRecord(c->stage_ev[b], c->stream);
        if (j->err != hipSuccess)
            return;
    }
}

int staged_upload(ibdg_ctx *c, const void *src, size_t n_rows, size_t rw, size_t dw, int fd = -1, uint64_t off = 0)
{
    int T = (int)std::min<unsigned>((unsigned)c->opt_stage_workers, std::max(1u, std::thread::hardware_concurrency()));
    const size_t rows_per_piece = std::max<size_t>(1, ibdg_ctx::STAGE_BYTES / rw);
    for (int b = 0; b < 2 * T; ++b) {
        if (!c->stage[b])
            HIP_TRY(c, hipHostMalloc(&c->stage[b], ibdg_ctx::STAGE_BYTES, hipHostMallocDefault));
        if (!c->stage_ev[b]) {
            HIP_TRY(c, hipEventCreateWithFlags(&c->stage_ev[b], hipEventDisableTiming));
            HIP_TRY(c, hipEventRecord(c->stage_ev[b], c->stream));       // "free" from the start
        }
    }
    std::vector<StageJob> jobs((size_t)T);
    std::vector<std::thread> th;
    for (int w = 0; w < T; ++w) {
        jobs[(size_t)w].c = c;
        jobs[(size_t)w].src = (const char *)src;
        jobs[(size_t)w].fd = fd;
        jobs[(size_t)w].off = off;
        jobs[(size_t)w].rw = rw;
        jobs[(size_t)w].dw = dw;
        jobs[(size_t)w].n_rows = n_rows;
        jobs[(size_t)w].rows_per_piece = rows_per_piece;
        jobs[(size_t)w].worker = w;
        jobs[(size_t)w].n_workers = T;
        th.emplace_back(stage_worker, &jobs[(size_t)w]);
    }
    for (auto &t : th)
        t.join();
    for (const StageJob &j : jobs) {
        if (j.err != hipSuccess)
            return fail(c, "[::] ERROR in ibdg_upload_panel: staged copy: %s", hipGetErrorString(j.err));
        if (j.io_failed)
            return fail(c, "[::] ERROR in ibdg_upload_panel_fd: the file ends before the rows do, or cannot be read");
    }
    return 0;
}

bool is_plain_host_memory(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();          // unknown to the runtime: ordinary memory
        return true;
    }
    return a.type == hipMemoryTypeUnregistered;
}

int copy_rows(ibdg_ctx *c, const void *src, size_t n_rows, hipMemcpyKind kind, int fd = -1, uint64_t off = 0)
{
    const size_t rw = ibdg_row_words(c->n_ids) * 8, dw = (size_t)c->stride * 8;
    if (n_rows == 0)
        return 0;
    if (rw != dw)
        HIP_TRY(c, hipMemsetAsync(c->panel.p, 0, n_rows * dw, c->stream));
    if (fd >= 0) {
        if (staged_upload(c, nullptr, n_rows, rw, dw, fd, off))
            return 1;
    } else if (kind == hipMemcpyHostToDevice && n_rows * rw >= ((size_t)256 << 20) && c->opt_staged_upload &&
        is_plain_host_memory(src)) {
        if (staged_upload(c, src, n_rows, rw, dw))
            return 1;
    } else {
        HIP_TRY(c, hipMemcpy2DAsync(c->panel.p, dw, src, rw, rw, n_rows, kind, c->stream));
    }
    if (!c->opt_count_in_run) {
        ibdg::launch_alt_count((const uint64_t *)c->panel.p, c->stride, n_rows, (uint32_t *)c->alt_count.p,
                               c->stream);
        HIP_TRY(c, hipGetLastError());
        c->counts_valid = true;
    }
    if (c->pop_lut_ok) {
        // second resident layout of the same bits for the fast --LD kernel
        ibdg::launch_transpose32((const uint64_t *)c->panel.p, c->stride, n_rows, c->n_chunks, c->n_pairs,
                                 (uint32_t *)c->t32.p, c->stream);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

// Runs of consecutive windows for the workgroups of the exponent-counting kernel: g windows each,
// except towards the end of the grid, where they shrink (guided self-scheduling: remaining
// windows / workgroups in flight): workgroups are handed out in blockIdx order, so the last
// ones to start are short and the CUs run dry together instead of waiting for one last
// full-length run.  A function of the window count alone.
void make_runs(const ibdg_ctx *c, uint32_t g, std::vector<uint32_t> &runs)
{
    const uint32_t n_cg = (c->n_chunks + 7) / 8, wpg_waves = (c->n_chunks + n_cg - 1) / n_cg;
    const uint32_t in_flight = std::max<uint32_t>(1, (uint32_t)(c->n_cu * (16 / wpg_waves) / n_cg) * (uint32_t)std::max<long>(1, c->opt_guided) / 4);
    runs.clear();
    for (uint32_t w = 0; w < c->n_win;) {
        runs.push_back(w);
        uint32_t len = g;
        if (c->opt_guided)
            len = std::min(g, std::max<uint32_t>(1, (c->n_win - w + in_flight - 1) / in_flight));
        w = std::min(w + len, c->n_win);
    }
    runs.push_back(c->n_win);
}

// rho^n, sigma^n and (1-eps)^n for n < need, in extended precision; the bases come from the doubles
// the reference uses (epsilon and 1-epsilon, src/ibd-math.c:58-61).  The tables depend on epsilon
// only, so a context builds every entry once and keeps it.  Returns false when an exponent leaves
// the range of the 32-bit fields (the strict kernel then serves).
bool grow_pow_tables(ibdg_ctx *c, size_t need)
{
    if (need > c->tab_fail_from)
        return false;
    if (need <= c->p1_h.size())
        return true;
    const size_t old = c->p1_h.size();
    const size_t n = std::max(need, std::min<size_t>(old + old / 2 + 256, c->tab_fail_from));
    const long double one_me = (long double)(double)(1 - c->eps);
    const ME rho = me_norm((long double)c->eps / one_me, 0), sigma = me_norm(0.5L / one_me, 0);
    // tau = rho / sigma^2 = 4 eps (1 - eps): the same long-double eps and 1 - eps as rho and sigma are made of
    const ME tau = me_norm(4.0L * (long double)c->eps * one_me, 0);
    c->p1_h.resize(n);
    c->p2_h.resize(n);
    c->p3_h.resize(n);
    c->pb_h.resize(n);
    for (size_t k = old; k < n; ++k) {
        const ME x = me_powl(rho, k), y = me_powl(sigma, k), z = me_pow(1 - c->eps, k), u = me_powl(tau, k);
        if (x.e < -2000000000LL / 3 || y.e < -2000000000LL / 3 || z.e < -2000000000LL / 3 || u.e < -2000000000LL / 3 ||
            u.e > 2000000000LL / 3) {
            c->tab_fail_from = k;
            c->p1_h.resize(k);
            c->p2_h.resize(k);
            c->p3_h.resize(k);
            c->pb_h.resize(k);
            c->tab_dev = std::min(c->tab_dev, k);
            return need <= k;
        }
        c->p1_h[k].m = (double)x.m; c->p1_h[k].e = (int32_t)x.e; c->p1_h[k].pad = 0;
        c->p2_h[k].m = (double)y.m; c->p2_h[k].e = (int32_t)y.e; c->p2_h[k].pad = 0;
        c->p3_h[k].m = (double)u.m; c->p3_h[k].e = (int32_t)u.e; c->p3_h[k].pad = 0;
        c->pb_h[k].m = (uint64_t)ldexpl(z.m, 64);        // exact: a 64-bit mantissa in [2^63, 2^64)
        c->pb_h[k].e = (int32_t)z.e;
        c->pb_h[k].pad = 0;
    }
    return true;
}

// Poll of one word in host memory until it holds `seq`; every 4096 spins `query` says whether the producer is still
// at work (0 = yes, 1 = it has finished, anything else = it has failed) and the wall clock is looked at.
// Returns 0 = the word arrived, 1 = the producer finished without writing it, 2 = timed out, < 0 = -(query's code).
template <class Q>
int poll_seq(const volatile uint32_t *flag, uint32_t seq, double timeout_s, Q query)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 1; *flag != seq; ++spins) {
        if ((spins & 0xfff) != 0)
            continue;
        const int q = query();
        if (q != 0) {
            if (*flag == seq)
                break;
            return q == 1 ? 1 : -q;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
            return *flag == seq ? 0 : 2;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return 0;
}

double wait_timeout_s()
{
    // ten seconds: a preparation stage is a few hundred microseconds of device work.  (IBDG_WAIT_TIMEOUT_MS: for tests.)
    if (const char *e = getenv("IBDG_WAIT_TIMEOUT_MS"))
        return atof(e) * 1e-3;
    return 10.0;
}

// Wait until the preparation kernels have handed hand-over number `seq` of PrepInfo to the host's mirror: a poll of
// one word in host memory, which the last workgroup of the stage writes -- no copy to queue, no stream to drain
// (each of those is a round trip of 10-15 us; an upload has two).  The stream is asked now and then whether it
// has died under us, and the poll is bounded by wall time: a stream that neither finishes nor fails (a wedged queue)
// ends the call with an error instead of a host thread spinning for ever.  prep_dirty stays set on every failure.
int wait_info(ibdg_ctx *c, uint32_t seq)
{
    hipError_t last = hipSuccess;
    const int rc = poll_seq(&c->info_h->seq, seq, wait_timeout_s(), [&]() {
        last = hipStreamQuery(c->stream);
        return last == hipErrorNotReady ? 0 : (last == hipSuccess ? 1 : 2);
    });
    if (rc == 0)
        return 0;
    if (rc == 1)
        return fail(c, "[::] ERROR in ibdg_upload_sites: the site preparation finished without reporting");
    if (rc == 2)
        return fail(c, "[::] ERROR in ibdg_upload_sites: the site preparation did not report within %.0f s (stream: %s)",
                    wait_timeout_s(), hipGetErrorString(hipStreamQuery(c->stream)));
    return fail(c, "[::] ERROR in ibdg_upload_sites: %s", hipGetErrorString(last));
}

// Segments, per-window constants, control words and power tables of the fast --LD kernel, built on
// the device from the covered-row list (ibdg_prep.hip); the host keeps what depends on the window
// count only (the run structure) and the epsilon-only power tables.
int build_segments(ibdg_ctx *c, bool compact)
{
    c->pop_sites_ok = false;
    if (!c->pop_lut_ok || c->n_cov == 0)
        return 0;
    const ibdg::PrepInfo &I = *c->info_h;
    // virtual rows per window of the compacted layout: the window rounded up to the alignment asked for (1 = the rows back
    // to back, 32 = every window on a tile boundary)
    const uint32_t align = (uint32_t)std::min<long>(32, std::max<long>(1, c->opt_compact_align));
    const uint64_t win_rows = ((uint64_t)c->window + align - 1) / align * align;
    const uint64_t vtiles = ((uint64_t)c->n_win * win_rows + 31) / 32;        // tiles of all its virtual rows
    if (compact && win_rows >= (1ull << 31))
        return 0;
    // segments <= windows + tiles spanned when the rows are in file order (otherwise the device stops
    // writing at the capacity and the exponent-counting kernel is not used); never more than stage A cleared
    uint64_t seg_cap = c->n_cov;
    const uint32_t first_row = c->first_row, last_row = c->last_row;
    if (compact)
        seg_cap = std::min<uint64_t>(seg_cap, (uint64_t)c->n_win + vtiles + 1);
    else if (last_row >= first_row)
        seg_cap = std::min<uint64_t>(seg_cap, (uint64_t)c->n_win + ((last_row >> 5) - (first_row >> 5)) + 1);
    seg_cap = std::min<uint64_t>(seg_cap, c->seg_room);
    if (compact) {
        // the rows with reads, gathered and transposed into tiles that start with their window
        const uint64_t pairs = (vtiles + 1) / 2;
        if (pairs >= (1ull << 31))
            return 0;
        c->n_pairs_c = (uint32_t)((pairs + 3) & ~3ull);
        if (ensure(c, c->t32c, (size_t)c->n_chunks * c->n_pairs_c * 64 * 16))
            return 1;
        ibdg::launch_gather_transpose32((const uint64_t *)c->panel.p, c->stride, (const uint2 *)c->rec_cov.p, c->n_cov,
                                        c->window, (uint32_t)win_rows, c->n_chunks, c->n_pairs_c, (uint32_t *)c->t32c.p, c->stream);
        HIP_TRY(c, hipGetLastError());
    }
    if (ensure(c, c->wconst, ((size_t)c->n_win + 1) * sizeof(ibdg::WinConst)) ||
        ensure(c, c->wraw, (size_t)c->n_win * sizeof(ibdg::WinRaw)))
        return 1;
    if (!c->nck_dev.p) {
        // the coefficients normalised here (C << clz(C), 64 - clz(C)): the device multiplies them as they are
        std::vector<ibdg::WinRaw> nn(c->nck_h.size());
        for (size_t i = 0; i < nn.size(); ++i) {
            const unsigned long v = c->nck_h[i];
            const int z = v ? __builtin_clzl(v) : 63;
            nn[i].m = v ? (uint64_t)v << z : (uint64_t)1 << 63;         // (0 is never looked up: r <= cov)
            nn[i].e = v ? 64 - z : 1;
            nn[i].pad = 0;
        }
        if (ensure(c, c->nck_dev, nn.size() * sizeof(ibdg::WinRaw))) return 1;
        // (a blocking copy, once per context: the kernel that reads the table runs on the second stream)
        HIP_TRY(c, hipMemcpy(c->nck_dev.p, nn.data(), nn.size() * sizeof(ibdg::WinRaw), hipMemcpyHostToDevice));
    }
    ibdg::PrepSegArgs sa;
    sa.rec_cov = (const uint2 *)c->rec_cov.p;
    sa.n_cov = c->n_cov;
    sa.window = c->window;
    sa.n_win = c->n_win;
    sa.max_cov = c->max_cov;
    sa.nck = (const ibdg::WinRaw *)c->nck_dev.p;
    sa.segs = (ibdg::Seg *)c->segs.p;
    sa.seg_first = (uint32_t *)c->seg_first.p;
    sa.seg_cap = (uint32_t)seg_cap;
    sa.wconst = (ibdg::WinConst *)c->wconst.p;
    sa.raw = (ibdg::WinRaw *)c->wraw.p;
    sa.block_tmp = (uint32_t *)c->scan_tmp.p;
    sa.info = (ibdg::PrepInfo *)c->info_dev.p;
    sa.mirror = c->info_h;
    sa.compact = compact ? (uint32_t)win_rows : 0u;
    c->prep_dirty = true;
    // windows per workgroup run: as many as keep the run's records within the LDS budget
    uint32_t g = (uint32_t)std::max<long>(1, c->opt_wpg);
    if (!c->opt_guided && !c->opt_wpg_fixed) {
        // uniform runs and few windows (a shard of a chromosome, a small region): shorter runs, so that
        // the grid still holds several rounds of workgroups for every CU
        const uint64_t rows_of_blocks = (c->n_chunks + 7) / 8;
        const uint64_t want_blocks = (uint64_t)c->n_cu * 2 * 5;          // CUs x resident blocks x rounds
        const uint64_t g_fit = std::max<uint64_t>(1, (uint64_t)c->n_win * rows_of_blocks / want_blocks);
        if (g_fit < g)
            g = (uint32_t)g_fit;
    }
    const uint32_t NS = (uint32_t)c->opt_ring;
    for (bool first_try = true;; g = (g + 1) / 2, first_try = false) {
        make_runs(c, g, c->runs_h);
        c->n_runs = (uint32_t)c->runs_h.size() - 1;
        if (ensure(c, c->runs, c->runs_h.size() * 4))
            return 1;
        // runs_h is a member: it outlives the copy (the next wait is wait_info below)
        HIP_TRY(c, hipMemcpyAsync(c->runs.p, c->runs_h.data(), c->runs_h.size() * 4, hipMemcpyHostToDevice, c->stream));
        if (first_try) {
            // the run structure goes ahead of the segment kernels, whose last one makes the control words for it
            // (stream2 is idle: ibdg_upload_sites drained both streams before it began; it waits for stage A's records)
            HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->ev_prepA, 0));
            ibdg::launch_prep_segments(sa, (const uint32_t *)c->runs.p, c->n_runs, NS, c->stream, c->stream2);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipEventRecord(c->ev_prep2, c->stream2));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_prep2, 0));      // before the hand-over (ct_max) and K'
        }
        ibdg::launch_prep_seg_flags(sa, (const uint32_t *)c->runs.p, c->n_runs, NS, ++c->prep_seq, c->stream, !first_try);
        HIP_TRY(c, hipGetLastError());
        if (wait_info(c, c->prep_seq))
            return 1;
        c->prep_dirty = false;
        if ((!compact && I.out_of_order) || I.n_segs > seg_cap)
            return 0;                      // not in file order: the panel's own tiles do not apply
        if ((size_t)I.max_seg * (sizeof(ibdg::Seg) + 8) <= (size_t)c->opt_recbytes || g == 1)
            break;
    }
    if (I.adv_overflow)
        return 0;                          // rows too far apart for the record format: strict kernel
    c->wpg = g;
    c->max_seg = I.max_seg;
    c->seg_ring = (int)NS;
    c->n_segs = I.n_segs;
    c->ct_max = I.ct_max;
    c->tab_in_lds = (size_t)(c->ct_max + 1) * 32 <= 24 * 1024;
    if (ibdg::ld_popcount_lds_bytes(c->max_seg, c->wpg, c->ct_max + 1, c->tab_in_lds, c->seg_ring, 0) > 150 * 1024)
        return 0;                          // a single window with thousands of tiles: strict kernel
    if (!grow_pow_tables(c, (size_t)c->ct_max + 1))
        return 0;
    if (c->tab_dev < c->p1_h.size()) {     // new entries since the last upload
        const size_t n = c->p1_h.size();
        if (ensure(c, c->pow1, n * sizeof(ibdg::PowEntry)) || ensure(c, c->pow2, n * sizeof(ibdg::PowEntry)) ||
            ensure(c, c->pow3, n * sizeof(ibdg::PowEntry)) || ensure(c, c->powb, n * sizeof(ibdg::WinRaw)))
            return 1;
        HIP_TRY(c, hipMemcpyAsync(c->pow3.p, c->p3_h.data(), n * sizeof(ibdg::PowEntry), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->pow1.p, c->p1_h.data(), n * sizeof(ibdg::PowEntry), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->pow2.p, c->p2_h.data(), n * sizeof(ibdg::PowEntry), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->powb.p, c->pb_h.data(), n * sizeof(ibdg::WinRaw), hipMemcpyHostToDevice, c->stream));
        c->tab_dev = n;
    }
    // K' = K * (1-eps)^(all reads of the window): with it a product is K' * rho^E2 * sigma^E3,
    // rho = eps/(1-eps), sigma = 1/(2(1-eps))  (E1 = reads - E2 - E3 eliminated)
    ibdg::launch_prep_win_kp(c->n_win, (const ibdg::WinRaw *)c->wraw.p, (const ibdg::WinRaw *)c->powb.p,
                             (ibdg::WinConst *)c->wconst.p, c->stream);
    HIP_TRY(c, hipGetLastError());
    c->pop_sites_ok = true;
    c->compact = compact;
    ++c->sites_gen;
    return 0;
}

// Sparse coverage: on the panel's own tiles the exponent-counting kernels stream every 32-row tile between a
// window's first and last row, whether its rows carry reads or not; on the compacted tiles a window costs
// ceil(window / 32) tile words whatever the density, plus its share of the gather (~6 tile words' worth of time per
// 100 rows).  Below about one covered row in four the compacted layout is the faster one for a single run.
bool sparse_sites(const ibdg_ctx *c)
{
    if (c->last_row < c->first_row)
        return true;                       // not even the ends are in file order
    const uint64_t span = (uint64_t)c->last_row - c->first_row + 1;
    return (uint64_t)c->n_cov * (uint64_t)std::max<long>(1, c->opt_compact_density) < span;
}

// Segments for the layout the options and the site list ask for; the compacted one also serves when the panel's own
// tiles turn out not to apply (rows out of file order, rows too far apart for the control words).
int build_layout(ibdg_ctx *c)
{
    c->pop_sites_ok = false;
    c->compact = false;
    c->pop_dense_enough = true;
    if (!c->pop_lut_ok || c->n_cov == 0)
        return 0;
    const bool sparse = sparse_sites(c);
    const bool want_compact = c->opt_compact > 0 || (c->opt_compact == 0 && sparse);
    if (build_segments(c, want_compact))
        return 1;
    if (!c->pop_sites_ok && !want_compact && c->opt_compact == 0 && build_segments(c, true))
        return 1;
    if (c->opt_compact < 0 && sparse)
        c->pop_dense_enough = false;       // the caller forbade the layout this pileup wants: the strict kernel is the faster one
    return 0;
}

}  // namespace

extern "C" {

int ibdg_abi_version(void) { return IBDG_ABI_VERSION; }

int ibdg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

const char *ibdg_last_error(const ibdg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int ibdg_pdg_table(double epsilon, unsigned max_cov, double *out)
{
    if (!out || max_cov < 1 || max_cov > 127)
        return 1;
    build_pdg_table(epsilon, max_cov, out);
    return 0;
}

// src/ibd-math.c:84-101 in k_site's operation order
double ibdg_pdg_ibd0(double f, double p00, double p01, double p11)
{
    if (p00 == 1 || p01 == 1 || p11 == 1)
        return 1.0;
    const double omf = 1 - f;
    const double t1 = libm_pow(omf, 2.0) * p00;
    const double t2 = ((2 * omf) * f) * p01;
    const double t3 = libm_pow(f, 2.0) * p11;
    double v = (t1 + t2) + t3;
    if (v == 0.0)
        v = DBL_MIN;
    return v;
}

// src/ibd-math.c:104-142
double ibdg_pdg_ibd1(unsigned a0, unsigned a1, double f, double p00, double p01, double p11)
{
    if (a0 > 1 || a1 > 1)
        return 1.0;                         // no branch of the reference's switch is taken
    const double omf = 1 - f;
    const unsigned g = a0 + a1;
    double v;
    if (g == 0)
        v = (f * p01) + (omf * p00);
    else if (g == 1)
        v = ((0.5 * p01) + ((0.5 * omf) * p00)) + ((0.5 * f) * p11);
    else
        v = (omf * p01) + (f * p11);
    if (v == 0.0)
        v = DBL_MIN;
    return v;
}

ibdg_ctx *ibdg_create(int device, double epsilon, unsigned max_cov)
{
    if (max_cov < 1 || max_cov > 127) {   // -M >= 1 (ibdgem.c:978); pileup rows have cov < 128 (pileup.c:223)
        fail(nullptr, "[::] ERROR: Invalid maximum estimated coverage (-M) of %u (must be 1..127).", max_cov);
        return nullptr;
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        fail(nullptr, "[::] ERROR in ibdg_create: no HIP device available (%s); this engine has no CPU path",
             e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= n) {
        fail(nullptr, "[::] ERROR in ibdg_create: device %d out of range (0..%d)", device, n - 1);
        return nullptr;
    }
    ibdg_ctx *c = new ibdg_ctx;
    c->device = device;
    c->eps = epsilon;
    c->max_cov = max_cov;
    auto bail = [&](const char *what, hipError_t err) {
        fail(nullptr, "[::] ERROR in ibdg_create: %s: %s", what, hipGetErrorString(err));
        ibdg_destroy(c);
        return (ibdg_ctx *)nullptr;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate", e);
    {
        // stream2 carries the short memory-bound kernels (alt counts, per-site values, window products) beside the
        // --LD kernel, whose workgroups fill every wave slot of the chip: at the highest priority its workgroups
        // take the slots that free up first instead of queueing behind ~17 000 --LD workgroups
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess)
            lo = hi = 0;
        if ((e = hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, hi)) != hipSuccess)
            return bail("hipStreamCreate", e);
        if ((e = hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, hi)) != hipSuccess)
            return bail("hipStreamCreate", e);
        if ((e = hipEventCreateWithFlags(&c->tg_ready, hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&c->ev_s3sync, hipEventDisableTiming)) != hipSuccess)
            return bail("hipEventCreate", e);
    }
    for (auto &E : c->evs) {
        for (hipEvent_t *ev : {&E.start_own, &E.ld_end, &E.k_start, &E.k_stop, &E.s2_start, &E.s2[0], &E.s2[1], &E.s2[2], &E.prep})
            if ((e = hipEventCreate(ev)) != hipSuccess) return bail("hipEventCreate", e);
    }
    for (hipEvent_t &ev : c->ev_up)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_prep2, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreateWithFlags(&c->ev_prepA, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    // (coherent: the kernels' stores must reach the host while the stream is still busy, not at its next drain)
    if ((e = hipHostMalloc((void **)&c->info_h, sizeof(ibdg::PrepInfo), hipHostMallocMapped | hipHostMallocCoherent)) != hipSuccess)
        return bail("hipHostMalloc", e);
    memset(c->info_h, 0, sizeof(ibdg::PrepInfo));
    {
        int n_cu = 0;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0)
            c->n_cu = n_cu;
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess)
            c->dev_mem_bytes = mem_total;
    }
    const size_t d = (size_t)max_cov + 1;
    c->lut_h.resize(d * d * 3);
    build_pdg_table(epsilon, max_cov, c->lut_h.data());
    c->nck_h = nck_table(max_cov);
    c->pop_lut_ok = lut_is_binomial(c->lut_h, c->nck_h, epsilon, max_cov);
    c->planes = 1;
    while ((1u << c->planes) <= max_cov)
        ++c->planes;
    if (ensure(c, c->lut, c->lut_h.size() * 8)) {
        fail(nullptr, "%s", c->err.c_str());
        ibdg_destroy(c);
        return nullptr;
    }
    if ((e = hipMemcpy(c->lut.p, c->lut_h.data(), c->lut_h.size() * 8, hipMemcpyHostToDevice)) != hipSuccess)
        return bail("hipMemcpy(lut)", e);
    return c;
}

void ibdg_destroy(ibdg_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    if (c->stream3)
        (void)hipStreamSynchronize(c->stream3);
    if (c->stream2)
        (void)hipStreamSynchronize(c->stream2);
    if (c->stream)
        (void)hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->lut, &c->pow_tab, &c->panel, &c->alt_count, &c->rec_all, &c->rec_cov, &c->cov_site,
                      &c->fo, &c->targets, &c->weight, &c->nrefpanel, &c->af, &c->site_ll, &c->win_ll, &c->row_tab, &c->t32, &c->t32c, &c->seg_first,
                      &c->segs, &c->runs, &c->wconst, &c->wtarget, &c->twords, &c->wtarget_mt, &c->twords_mt, &c->vals, &c->order, &c->pow1, &c->pow2, &c->pow3, &c->partial, &c->aimg, &c->wc_slot, &c->partial_h, &c->base_w, &c->p2w, &c->p2c, &c->p2_tw, &c->p2_wt, &c->fragb,
                      &c->in_row, &c->in_ref, &c->in_alt, &c->scan_tmp, &c->info_dev, &c->wraw, &c->nck_dev, &c->powb,
                      &c->win_first, &c->win_last})
        release(*b);
    for (hipEvent_t ev : c->ev_up)
        if (ev)
            (void)hipEventDestroy(ev);
    if (c->ev_prep2)
        (void)hipEventDestroy(c->ev_prep2);
    if (c->ev_prepA)
        (void)hipEventDestroy(c->ev_prepA);
    for (int b = 0; b < 2 * ibdg_ctx::STAGE_WORKERS; ++b) {
        if (c->stage_ev[b])
            (void)hipEventDestroy(c->stage_ev[b]);
        if (c->stage[b])
            (void)hipHostFree(c->stage[b]);
    }
    if (c->info_h)
        (void)hipHostFree(c->info_h);
    for (auto &E : c->evs)
        for (hipEvent_t ev : {E.start_own, E.ld_end, E.k_start, E.k_stop, E.s2_start, E.s2[0], E.s2[1], E.s2[2], E.prep})
            if (ev)
                (void)hipEventDestroy(ev);
    for (int i = 0; i < ibdg_ctx::TG_SLOTS; ++i) {
        if (c->tg_stage[i])
            (void)hipHostFree(c->tg_stage[i]);
        if (c->tg_stage_ev[i])
            (void)hipEventDestroy(c->tg_stage_ev[i]);
    }
    for (hipEvent_t ev : {c->tg_ready, c->ev_s3sync, c->ev_fb})
        if (ev)
            (void)hipEventDestroy(ev);
    if (c->stream3)
        (void)hipStreamDestroy(c->stream3);
    if (c->stream2)
        (void)hipStreamDestroy(c->stream2);
    if (c->stream)
        (void)hipStreamDestroy(c->stream);
    delete c;
}

size_t ibdg_row_words(unsigned n_ids) { return 2 * (((size_t)n_ids + 63) / 64); }

void ibdg_pack_alleles(const uint8_t *alleles, unsigned n_ids, uint64_t *row)
{
    const size_t words = ibdg_row_words(n_ids);
    memset(row, 0, words * 8);
    for (unsigned n = 0; n < n_ids; ++n) {
        const size_t w = 2 * (size_t)(n >> 6);
        const uint64_t bit = 1ull << (n & 63);
        if (alleles[2 * n] == 1) row[w] |= bit;
        if (alleles[2 * n + 1] == 1) row[w + 1] |= bit;
    }
}

int ibdg_pack_hap_text(const char *line, unsigned n_ids, uint64_t *row)
{
    const size_t words = ibdg_row_words(n_ids);
    memset(row, 0, words * 8);
    const size_t need = 4 * (size_t)n_ids - 1;
    if (strnlen(line, need) < need)
        return 1;
    int bad = 0;
    for (unsigned n = 0; n < n_ids; ++n) {
        const char c0 = line[4 * (size_t)n], c1 = line[4 * (size_t)n + 2];
        const size_t w = 2 * (size_t)(n >> 6);
        const uint64_t bit = 1ull << (n & 63);
        if (c0 == '1') row[w] |= bit; else if (c0 != '0') bad = 1;
        if (c1 == '1') row[w + 1] |= bit; else if (c1 != '0') bad = 1;
    }
    return bad;
}

int ibdg_upload_panel(ibdg_ctx *c, const uint64_t *rows, size_t n_rows, unsigned n_ids)
{
    if (!c) return 1;
    if (!rows && n_rows) return fail(c, "[::] ERROR in ibdg_upload_panel: rows is NULL");
    if (quiesce(c)) return 1;
    if (prepare_panel(c, n_rows, n_ids)) return 1;
    return copy_rows(c, rows, n_rows, hipMemcpyHostToDevice);
}

int ibdg_upload_panel_fd(ibdg_ctx *c, int fd, uint64_t offset, size_t n_rows, unsigned n_ids)
{
    if (!c) return 1;
    if (fd < 0) return fail(c, "[::] ERROR in ibdg_upload_panel_fd: not an open file");
    if (quiesce(c)) return 1;
    if (prepare_panel(c, n_rows, n_ids)) return 1;
    return copy_rows(c, nullptr, n_rows, hipMemcpyHostToDevice, fd, offset);
}

int ibdg_upload_panel_dev(ibdg_ctx *c, const void *dev_rows, size_t n_rows, unsigned n_ids)
{
    if (!c) return 1;
    if (!dev_rows && n_rows) return fail(c, "[::] ERROR in ibdg_upload_panel_dev: rows is NULL");
    if (quiesce(c)) return 1;
    if (prepare_panel(c, n_rows, n_ids)) return 1;
    // the source may have been produced on another stream (e.g. torch's): make it visible first
    HIP_TRY(c, hipDeviceSynchronize());
    return copy_rows(c, dev_rows, n_rows, hipMemcpyDeviceToDevice);
}

// Everything an upload of sites does once the three input arrays are on the device.
static int upload_sites_core(ibdg_ctx *c, const uint32_t *d_row, const uint8_t *d_ref, const uint8_t *d_alt,
                             const double *f_override, size_t n_sites, unsigned window)
{
    c->n_sites = n_sites;
    c->window = window;
    c->n_cov = 0;
    c->n_win = 0;
    c->sites_valid = false;
    c->have_results = false;
    c->pop_sites_ok = false;
    c->win_bounds_valid = false;
    c->have_fo = false;
    // segments of stage B: at most one per site, and in file order at most windows + tiles of the panel
    // (compacted tiles: windows + tiles of their virtual rows, at most window + 31 per window)
    const size_t n_win_max = (n_sites + window - 1) / window;
    const size_t seg_room = std::min<size_t>(n_sites, std::max<size_t>(n_win_max + (c->n_rows + 31) / 32 + 1,
                                                                      n_win_max + (n_win_max * ((size_t)window + 31) + 31) / 32 + 1));
    const bool fresh_info = !c->info_dev.p;
    if (ensure(c, c->rec_all, n_sites * 8) || ensure(c, c->rec_cov, n_sites * 8) || ensure(c, c->cov_site, n_sites * 4) ||
        ensure(c, c->scan_tmp, ibdg::prep_scan_blocks(std::max<size_t>(n_sites, 1)) * 4) ||
        ensure(c, c->info_dev, sizeof(ibdg::PrepInfo)) ||
        (c->pop_lut_ok && (ensure(c, c->segs, seg_room * sizeof(ibdg::Seg)) || ensure(c, c->seg_first, seg_room * 4))))
        return 1;
    if (fresh_info || c->prep_dirty) {
        // the device's PrepInfo starts clean; afterwards every upload leaves it so (k_prep_mirror) -- unless it
        // stopped half way: prep_dirty
        ibdg::PrepInfo init;
        memset(&init, 0, sizeof init);
        init.err_row_site = init.err_cov_site = init.first_row = 0xffffffffu;
        HIP_TRY(c, hipMemcpyAsync(c->info_dev.p, &init, sizeof init, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));          // `init` is a local
        c->prep_dirty = false;
    }
    ++c->sites_gen;
    c->relayout_credit = 0;
    c->ibd0_runs = 0;
    c->seg_room = c->pop_lut_ok ? seg_room : 0;
    c->compact = false;
    if (n_sites) {
        c->prep_dirty = true;
        ibdg::PrepSiteArgs pa;
        pa.row_index = d_row;
        pa.n_ref = d_ref;
        pa.n_alt = d_alt;
        pa.n_sites = n_sites;
        pa.n_rows = c->n_rows;
        pa.max_cov = c->max_cov;
        pa.rec_all = (uint2 *)c->rec_all.p;
        pa.rec_cov = (uint2 *)c->rec_cov.p;
        pa.cov_site = (uint32_t *)c->cov_site.p;
        pa.block_tmp = (uint32_t *)c->scan_tmp.p;
        pa.info = (ibdg::PrepInfo *)c->info_dev.p;
        pa.mirror = c->info_h;
        pa.seq = ++c->prep_seq;
        ibdg::launch_prep_sites(pa, c->stream);
        HIP_TRY(c, hipGetLastError());
        // (the hand-over comes from the scan, BEFORE the scatter kernel: whatever reads the site records on the second
        // stream waits for this event; the main stream is ordered anyway)
        HIP_TRY(c, hipEventRecord(c->ev_prepA, c->stream));
        if (wait_info(c, c->prep_seq))
            return 1;
        c->prep_dirty = false;                  // the stage's last workgroup has left everything clean
    } else {
        c->prep_dirty = false;
        memset(c->info_h, 0, sizeof(ibdg::PrepInfo));
        c->info_h->err_row_site = c->info_h->err_cov_site = c->info_h->first_row = 0xffffffffu;
        c->info_h->seq = c->prep_seq;
    }
    const ibdg::PrepInfo &I = *c->info_h;
    if (I.err_row_site != 0xffffffffu || I.err_cov_site != 0xffffffffu) {
        // the first offending site in file order, its row checked before its counts (as a loop over the sites would)
        c->n_sites = 0;
        if (I.err_row_site <= I.err_cov_site) {
            uint32_t row = 0;
            HIP_TRY(c, hipMemcpy(&row, d_row + I.err_row_site, 4, hipMemcpyDeviceToHost));
            return fail(c, "[::] ERROR in ibdg_upload_sites: row_index[%zu]=%u outside the panel (%zu rows)",
                        (size_t)I.err_row_site, row, c->n_rows);
        }
        uint8_t r = 0, a = 0;
        HIP_TRY(c, hipMemcpy(&r, d_ref + I.err_cov_site, 1, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(&a, d_alt + I.err_cov_site, 1, hipMemcpyDeviceToHost));
        return fail(c, "[::] ERROR in ibdg_upload_sites: site %zu has n_ref+n_alt=%u > max_cov=%u",
                    (size_t)I.err_cov_site, (unsigned)r + a, c->max_cov);
    }
    c->n_cov = I.n_cov;
    c->first_row = I.first_row;             // (the mirror is overwritten by stage B's hand-over)
    c->last_row = I.last_row;
    c->n_win = (uint32_t)(((uint64_t)c->n_cov + window - 1) / window);
    if (build_layout(c))
        return 1;
    if (f_override) {
        std::vector<double> fo(3 * n_sites);
        for (size_t s = 0; s < n_sites; ++s) {
            const double f = f_override[s];
            fo[3 * s] = f;
            if (f == f) {
                fo[3 * s + 1] = libm_pow(1 - f, 2.0);
                fo[3 * s + 2] = libm_pow(f, 2.0);
                c->have_fo = true;
            } else {
                fo[3 * s + 1] = fo[3 * s + 2] = 0.0;
            }
        }
        if (c->have_fo) {
            if (ensure(c, c->fo, fo.size() * 8)) return 1;
            HIP_TRY(c, hipMemcpyAsync(c->fo.p, fo.data(), fo.size() * 8, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));          // `fo` is a local
        }
    }
    return 0;
}

static int upload_sites_check(ibdg_ctx *c, const void *n_ref, const void *n_alt, bool have_rows, size_t n_sites,
                              unsigned window)
{
    if (!c->panel.p || c->n_ids == 0) return fail(c, "[::] ERROR in ibdg_upload_sites: no panel uploaded");
    if (window < 1) return fail(c, "[::] ERROR: Invalid window size (-w) of %u (must be >= 1).", window);
    if (n_sites && (!n_ref || !n_alt))
        return fail(c, "[::] ERROR in ibdg_upload_sites: NULL input array");
    if (n_sites > 0xffffffffull)
        return fail(c, "[::] ERROR in ibdg_upload_sites: more than 2^32-1 rows in one call");
    if (!have_rows && n_sites > c->n_rows)
        return fail(c, "[::] ERROR in ibdg_upload_sites: row_index[%zu]=%zu outside the panel (%zu rows)", c->n_rows,
                    c->n_rows, c->n_rows);
    return 0;
}

static double wall_ms(std::chrono::steady_clock::time_point t0)
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// After the last kernel of an upload.  The call does not wait for it: everything the host needed it has polled
// for, the caller's arrays have been read, and whatever uses the prepared sites next is queued behind that kernel
// (K' of the windows) on the same stream.  The event clocks are read when somebody asks (ibdg_upload_ms).
static int upload_sites_finish(ibdg_ctx *c, int rc, std::chrono::steady_clock::time_point t0)
{
    if (rc)
        return rc;
    c->sites_valid = true;
    HIP_TRY(c, hipEventRecord(c->ev_up[2], c->stream));
    c->up_ms_pending = true;
    c->up_ms[2] = (float)wall_ms(t0);
    return 0;
}

int ibdg_upload_sites(ibdg_ctx *c, const uint32_t *row_index, const uint8_t *n_ref, const uint8_t *n_alt,
                      const double *f_override, size_t n_sites, unsigned window)
{
    if (!c) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    if (upload_sites_check(c, n_ref, n_alt, row_index != nullptr, n_sites, window)) return 1;
    if (quiesce(c)) return 1;
    if (ensure(c, c->in_ref, n_sites) || ensure(c, c->in_alt, n_sites) || (row_index && ensure(c, c->in_row, n_sites * 4)))
        return 1;
    HIP_TRY(c, hipEventRecord(c->ev_up[0], c->stream));
    if (n_sites) {
        // 6 bytes per row (2 when the rows are the panel's own); pinned arrays (ibdg_host_alloc) go at link speed
        if (row_index)
            HIP_TRY(c, hipMemcpyAsync(c->in_row.p, row_index, n_sites * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->in_ref.p, n_ref, n_sites, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->in_alt.p, n_alt, n_sites, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipEventRecord(c->ev_up[1], c->stream));
    return upload_sites_finish(c, upload_sites_core(c, row_index ? (const uint32_t *)c->in_row.p : nullptr,
                                                    (const uint8_t *)c->in_ref.p, (const uint8_t *)c->in_alt.p,
                                                    f_override, n_sites, window), t0);
}

int ibdg_upload_sites_dev(ibdg_ctx *c, const void *dev_row_index, const void *dev_n_ref, const void *dev_n_alt,
                          const double *f_override, size_t n_sites, unsigned window)
{
    if (!c) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    if (upload_sites_check(c, dev_n_ref, dev_n_alt, dev_row_index != nullptr, n_sites, window)) return 1;
    if (quiesce(c)) return 1;
    // the arrays may have been produced on another stream (e.g. torch's): make them visible first.  This waits for the
    // whole device -- other contexts' kernels included -- so a caller who knows the arrays are complete says so
    // (option "dev_inputs_ready") and its preparation can run under another context's --LD kernel.
    if (!c->opt_dev_inputs_ready)
        HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipEventRecord(c->ev_up[0], c->stream));
    HIP_TRY(c, hipEventRecord(c->ev_up[1], c->stream));
    const int rc = upload_sites_core(c, (const uint32_t *)dev_row_index, (const uint8_t *)dev_n_ref,
                                     (const uint8_t *)dev_n_alt, f_override, n_sites, window);
    // the caller may free or reuse its arrays when this returns: the last kernel that reads them (k_prep_site_scatter) must
    // be done.  Where the layout's second stage was waited for it is (nothing to wait for); a clamped table, an empty
    // site list or a layout without segments leave that stage out
    if (rc == 0 && n_sites)
        HIP_TRY(c, hipEventSynchronize(c->ev_prepA));
    return upload_sites_finish(c, rc, t0);
}

int ibdg_upload_ms(ibdg_ctx *c, float out[3])
{
    if (!c || !out) return 1;
    if (c->up_ms_pending) {
        HIP_TRY(c, hipEventSynchronize(c->ev_up[2]));
        HIP_TRY(c, hipEventElapsedTime(&c->up_ms[0], c->ev_up[0], c->ev_up[1]));
        HIP_TRY(c, hipEventElapsedTime(&c->up_ms[1], c->ev_up[1], c->ev_up[2]));
        c->up_ms_pending = false;
    }
    for (int i = 0; i < 3; ++i) out[i] = c->up_ms[i];
    return 0;
}

void *ibdg_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess)
        return nullptr;
    return p;
}

void ibdg_host_free(void *p)
{
    if (p)
        (void)hipHostFree(p);
}

size_t ibdg_num_sites(const ibdg_ctx *c) { return c ? c->n_sites : 0; }
size_t ibdg_num_windows(const ibdg_ctx *c) { return c ? c->n_win : 0; }
size_t ibdg_num_targets(const ibdg_ctx *c) { return c && c->have_results ? c->n_targets : 0; }

int ibdg_get_windows(ibdg_ctx *c, uint32_t *first, uint32_t *last, uint32_t *n_covered)
{
    if (!c) return 1;
    if (c->n_win && (first || last) && !c->win_bounds_valid) {
        if (quiesce(c)) return 1;
        if (ensure(c, c->win_first, (size_t)c->n_win * 4) || ensure(c, c->win_last, (size_t)c->n_win * 4))
            return 1;
        ibdg::launch_prep_win_bounds((const uint32_t *)c->cov_site.p, c->n_cov, c->window, c->n_win,
                                     (uint32_t *)c->win_first.p, (uint32_t *)c->win_last.p, c->stream);
        HIP_TRY(c, hipGetLastError());
        c->win_first_h.resize(c->n_win);
        c->win_last_h.resize(c->n_win);
        HIP_TRY(c, hipMemcpyAsync(c->win_first_h.data(), c->win_first.p, (size_t)c->n_win * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(c->win_last_h.data(), c->win_last.p, (size_t)c->n_win * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->win_bounds_valid = true;
    }
    for (uint32_t w = 0; w < c->n_win; ++w) {
        const uint64_t b = (uint64_t)w * c->window;
        const uint64_t e = std::min<uint64_t>(b + c->window, c->n_cov);
        if (first) first[w] = c->win_first_h[w];
        if (last) last[w] = c->win_last_h[w];
        if (n_covered) n_covered[w] = (uint32_t)(e - b);
    }
    return 0;
}

int ibdg_run(ibdg_ctx *c, const uint32_t *targets, size_t T, const uint8_t *bg_count, int pu_id, int ld_mode)
{
    if (!c) return 1;
    if (!c->panel.p || c->n_ids == 0) return fail(c, "[::] ERROR in ibdg_run: no panel uploaded");
    if (!c->sites_valid) return fail(c, "[::] ERROR in ibdg_run: no sites uploaded");
    if (T == 0 || !targets) return fail(c, "[::] ERROR in ibdg_run: no targets");
    if (T > 65535) return fail(c, "[::] ERROR in ibdg_run: at most 65535 targets per call");
    for (size_t t = 0; t < T; ++t)
        if (targets[t] >= c->n_ids)
            return fail(c, "[::] ERROR in ibdg_run: target %u is not a panel individual (n_ids=%u)", targets[t],
                        c->n_ids);
    HIP_TRY(c, hipSetDevice(c->device));

    const size_t lanes = (size_t)c->n_groups * c->cpw * 64;
    const bool want_ll = c->opt_site_results != 0;
    // --LD over several comparison individuals: one table of the rows' values for all of them (they differ by the genotype
    // picked), an individual's per-site table is put together when it is fetched (k_row_table / k_site_expand)
    const bool row_table = want_ll && ld_mode && T > 1;
    if (ensure(c, c->targets, ibdg_ctx::TG_RING * T * 4) || (want_ll && ensure(c, c->site_ll, (row_table ? 1 : T) * c->n_sites * 24)) ||
        (row_table && ensure(c, c->row_tab, c->n_sites * 32)) || ensure(c, c->win_ll, T * (size_t)c->n_win * 24))
        return 1;
    // targets / background weights change rarely between calls (a loop over windows sizes, repeated
    // timing steps): their device copies are rebuilt only when the inputs differ
    const bool same_bg = c->prev_pu == pu_id && c->prev_has_bg == (bg_count ? 1 : 0) && c->prev_lanes == lanes &&
                         (!bg_count || (c->prev_bg.size() == c->n_ids &&
                                        std::equal(bg_count, bg_count + c->n_ids, c->prev_bg.begin()))) &&
                         c->base_w.p;
    const bool same_inputs = same_bg && c->prev_targets.size() == T && std::equal(targets, targets + T, c->prev_targets.begin()) &&
                             c->weight.p;
    // a finalising step left to "the next run" is taken along only by a run of the same shape over the same background and
    // prepared sites (it reads the windows' constants and its own run's background sizes): anything else makes up for it first
    if (c->fin_pending && (!same_bg || c->prev_targets.size() != T || !ld_mode || c->fin_sites_gen != c->sites_gen ||
                           !c->opt_fin_next || !c->opt_async) &&
        flush_finalize(c))
        return 1;
    if (!same_bg) {
        // background multiplicity per individual without any comparison individual's own exclusion; the -N sample
        // contributes nothing (src/ibdgem.c:714, :742-750).  Rare (once per program run): a host wait is fine here.
        std::vector<double> wb(lanes, 0.0);
        int sum = 0;
        for (unsigned n = 0; n < c->n_ids; ++n) {
            const unsigned k = bg_count ? bg_count[n] : 1u;
            if ((int)n != pu_id && k != 0) {
                wb[n] = (double)k;
                sum += (int)k;
            }
        }
        if (ensure(c, c->base_w, lanes * 8))
            return 1;
        HIP_TRY(c, hipMemcpyAsync(c->base_w.p, wb.data(), lanes * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));          // the host vector goes out of scope
        c->chain_ok = false;
        ++c->bg_gen;
        c->base_sum = sum;
        c->prev_pu = pu_id;
        c->prev_has_bg = bg_count ? 1 : 0;
        c->prev_lanes = lanes;
        if (bg_count)
            c->prev_bg.assign(bg_count, bg_count + c->n_ids);
        else
            c->prev_bg.clear();
    }
    // Runs of a few individuals keep a ring of TG_RING copies of everything that depends on the individuals, so that the NEXT
    // run's can be made (on stream3) while the runs before still read theirs; larger runs use the buffers whole, on the main stream.
    if (c->s3_unsettled) {          // (left by a run that failed after it had queued its preparation)
        HIP_TRY(c, hipEventRecord(c->tg_ready, c->stream3));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->tg_ready, 0));
        HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->tg_ready, 0));
        c->s3_unsettled = false;
    }
    const bool ahead_cap = T <= ibdg_ctx::AHEAD_MAX_T;
    bool &need_ready = c->s3_unsettled;   // stream3 holds this run's preparation: the other streams wait for tg_ready before they read it
    hipStream_t ps = c->stream;     // where this run's per-individual preparation is queued
    if (!same_inputs) {
        if (ensure(c, c->weight, (ahead_cap ? ibdg_ctx::TG_RING : 1) * T * lanes * 8) || ensure(c, c->nrefpanel, ibdg_ctx::NREF_SLOTS * T * 8))       // (per slot: T background sizes, T individuals)
            return 1;
        // more than a few individuals: a page-locked slot for the indices (so that the copy is a queued one), grown when a run
        // brings more of them; up to IBDG_TG_INLINE of them travel in the weights kernel's arguments instead
        const bool inline_tg = T <= IBDG_TG_INLINE;
        int slot = -1;
        if (!inline_tg) {
            if (c->tg_stage_cap < T) {
                if (quiesce(c)) return 1;
                const size_t cap = std::max<size_t>(64, T);
                for (int i = 0; i < ibdg_ctx::TG_SLOTS; ++i) {
                    if (c->tg_stage[i])
                        (void)hipHostFree(c->tg_stage[i]);
                    c->tg_stage[i] = nullptr;
                    HIP_TRY(c, hipHostMalloc((void **)&c->tg_stage[i], cap * 4, hipHostMallocDefault));
                    if (!c->tg_stage_ev[i])
                        HIP_TRY(c, hipEventCreateWithFlags(&c->tg_stage_ev[i], hipEventDisableTiming));
                    c->tg_stage_busy[i] = false;
                }
                c->tg_stage_cap = cap;
            }
            slot = c->tg_slot;
            c->tg_slot = (slot + 1) % ibdg_ctx::TG_SLOTS;
            if (c->tg_stage_busy[slot])
                HIP_TRY(c, hipEventSynchronize(c->tg_stage_ev[slot]));      // (its copy was queued TG_SLOTS runs ago)
            std::copy(targets, targets + T, c->tg_stage[slot]);
        }
        const bool on_s3 = ahead_cap && c->opt_prep_ahead && c->opt_async;
        const int rs = ahead_cap ? (c->tg_cur + 1) % ibdg_ctx::TG_RING : 0;
        if (on_s3) {
            ps = c->stream3;
            need_ready = true;
        }
        // whoever still reads the ring slot this run's data go to: the run four new individuals back, long done
        for (int h = ahead_cap ? rs : 0; h <= (ahead_cap ? rs : ibdg_ctx::TG_RING - 1); ++h) {
            if (c->tg_main_pending[h] && ps != c->stream)
                HIP_TRY(c, hipStreamWaitEvent(ps, c->tg_main[h], 0));
            if (c->tg_s2_pending[h])
                HIP_TRY(c, hipStreamWaitEvent(ps, c->tg_s2[h], 0));
            c->tg_main_pending[h] = c->tg_s2_pending[h] = false;
        }
        c->tg_cur = rs;
        c->nref_slot = (c->nref_slot + 1) % ibdg_ctx::NREF_SLOTS;
        uint32_t *d_tg = (uint32_t *)((char *)c->targets.p + (size_t)rs * (c->targets.cap / ibdg_ctx::TG_RING / 4 * 4));
        int *d_nref = (int *)((char *)c->nrefpanel.p + (size_t)c->nref_slot * (c->nrefpanel.cap / ibdg_ctx::NREF_SLOTS / 4 * 4));
        double *d_w = (double *)((char *)c->weight.p + (size_t)rs * (c->weight.cap / ibdg_ctx::TG_RING / 8 * 8));
        if (!inline_tg) {
            HIP_TRY(c, hipMemcpyAsync(d_tg, c->tg_stage[slot], T * 4, hipMemcpyHostToDevice, ps));
            HIP_TRY(c, hipEventRecord(c->tg_stage_ev[slot], ps));
            c->tg_stage_busy[slot] = true;
        }
        ibdg::launch_target_weights((const double *)c->base_w.p, d_tg, inline_tg ? targets : nullptr, (uint32_t)T, (uint32_t)lanes,
                                    c->base_sum, d_w, d_nref, ps);
        if (ps == c->stream) {
            // prepared on the main stream (more than AHEAD_MAX_T individuals, "prep_ahead" 0, no queue): the second stream reads
            // the individuals' indices too (k_rows_windows, k_row_table) and, in a queue of runs, starts behind the PREVIOUS
            // run's end only -- it must not overtake this copy / kernel
            HIP_TRY(c, hipEventRecord(c->tg_ready, c->stream));
            HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->tg_ready, 0));
        }
        c->prev_targets.assign(targets, targets + T);
        c->wt_gen = 0;             // the images in wtarget / twords are another individual's
    }
    const uint32_t *const d_targets = (const uint32_t *)((const char *)c->targets.p + (size_t)c->tg_cur * (c->targets.cap / ibdg_ctx::TG_RING / 4 * 4));
    const int *const d_nrefpanel = (const int *)((const char *)c->nrefpanel.p +
                                                 (size_t)c->nref_slot * (c->nrefpanel.cap / ibdg_ctx::NREF_SLOTS / 4 * 4));
    const double *const d_weight = (const double *)((const char *)c->weight.p + (size_t)c->tg_cur * (c->weight.cap / ibdg_ctx::TG_RING / 8 * 8));
    // the other streams join stream3's preparation (once, before the first thing that reads it)
    auto settle_ready = [&]() -> int {
        if (!need_ready)
            return 0;
        need_ready = false;
        HIP_TRY(c, hipEventRecord(c->tg_ready, c->stream3));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->tg_ready, 0));
        HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->tg_ready, 0));
        return 0;
    };

    bool use_pop = false, s2_after_prep = false, side_fast = false, end_recorded = false;
    if (ld_mode && c->pop_lut_ok && c->pop_sites_ok && !c->compact && c->opt_compact == 0 && c->opt_variant != 1 &&
        c->opt_variant != 3 && c->pop_dense_enough) {
        // Comparison individuals over one site list (the site list belongs to the pileup, not to the comparison
        // individual: src/ibdgem.c:522 loops the individuals over the same rows): the compacted tiles' one-off gather
        // is paid back by the fewer segments every later run counts.  The runs on an upload add up -- one run of
        // 256 individuals, nine batches of 30, or sixteen runs of one individual through the counting kernel all
        // reach the point where the re-layout has paid for itself (a rent-or-buy rule: never more than twice the
        // cost of having known the number of runs beforehand).
        // (a group of the matrix-core kernel costs the same whether it holds 3 or IBDG_TG individuals)
        // (round 5: on the site list's rows back to back -- no padding, no rows without reads -- the counting kernel with its sums
        // on the matrix cores takes 0.548 ms where the panel's own tiles take 0.606 and round 4's window-aligned tiles took
        // 0.58-0.59, profiles/r05_layouts.txt: a single run saves 0.058 ms of the 1.3 ms the gather and the new segments
        // cost, i.e. 22 runs pay for them -- an individual counts as 12; with (mask, count) pairs, option mx_counts 0, as 16)
        const bool to_mfma = c->opt_mfma_targets && c->tab_in_lds && T >= (size_t)c->opt_mfma_min;
        // (a group of the matrix-core kernel saves 0.085 ms of 2.2 on the rows back to back -- 8.90 against 8.57 ms per 60 individuals,
        // `many_comparison_individuals` of the bench's detail file, since the launch's groups share the tile words through an
        // XCD's L2 --, i.e. fifteen groups pay for the re-layout: a group counts as 20; 45 earlier in round 5, when a group saved
        // 0.18 ms, 15 until round 5)
        c->relayout_credit += to_mfma ? (uint64_t)((T + IBDG_TG - 1) / IBDG_TG) * 20u : (uint64_t)T * (c->opt_mx_counts ? 12u : 16u);
        if (c->relayout_credit >= (uint64_t)std::max<long>(1, c->opt_compact_targets)) {
            if (quiesce(c)) return 1;
            if (build_segments(c, true)) return 1;
            if (!c->pop_sites_ok && build_segments(c, false)) return 1;     // (cannot happen: it applied a moment ago)
        }
    }
    if (ld_mode) {
        const bool can = c->pop_lut_ok && c->pop_sites_ok && (c->compact ? c->t32c.p : c->t32.p);
        if (c->opt_variant == 2 && !can)
            return fail(c, "[::] ERROR in ibdg_run: ld_variant 2 (exponent counting) is not applicable here "
                           "(clamped P(D|G) table, epsilon outside (0,1), max_cov > 50 or rows out of order)");
        use_pop = can && c->opt_variant != 1 && c->opt_variant != 3 && (c->opt_variant == 2 || c->pop_dense_enough);
    }
    // (a finalising step left to the next run is taken along by the counting kernel only: the strict kernels write the same
    // win_ll entries themselves, and the stale sums must not land behind them)
    if (c->fin_pending && !use_pop && flush_finalize(c))
        return 1;
    c->last_variant = ld_mode ? (use_pop ? 2 : (c->opt_variant == 3 ? 3 : 1)) : 0;
    c->last_count_unit = 0;
    const bool recount = c->opt_count_in_run || !c->counts_valid;
    const int ev_slot = (c->ev_head + 1) % ibdg_ctx::EV_RING;      // becomes the head once the run is queued
    ibdg_ctx::EvSet &E = c->evs[ev_slot];
    E.recount = recount;
    E.ld = ld_mode != 0;
    // a non-LD run is one kernel: it goes to the main stream (no second stream to start, wait for and join)
    const bool rows_on_main = !ld_mode && !recount;
    E.rows_on_main = rows_on_main;
    // Two streams: the per-site kernel and the window products (memory-bound, few waves) run on
    // stream2 beside the --LD kernels (VALU-bound) on the main stream.  stream2 starts a run when
    // the main stream does (a wait in stream2's queue costs the main stream nothing; it also orders
    // stream2 behind an upload of new targets); the main stream waits for stream2 when somebody
    // needs the results (join_streams), not once per run.
    // Timing: normally one event record per run on the main stream (below).  With the option
    // "dispatch_events" the exponent-counting launches carry events in their own dispatch packets
    // instead (hipExtLaunchKernel: start of the first, stop of the last, and both of the dominant
    // kernel -- the only way to time that kernel alone from inside the process).
    const bool dispatch_events = use_pop && c->opt_dispatch_events && c->n_win > 0;
    E.has_kernel_times = dispatch_events;
    if (dispatch_events) {
        E.start = E.start_own;                     // filled in by the first --LD dispatch
        if (c->chain_ok && c->opt_async)           // stream2 keeps one run behind the main stream at most
            HIP_TRY(c, hipStreamWaitEvent(c->stream2, c->evs[c->ev_head].ld_end, 0));
    } else {
        if (c->chain_ok && c->opt_async) {
            E.start = c->evs[c->ev_head].ld_end;   // back-to-back runs: the previous end is this start
        } else {
            HIP_TRY(c, hipEventRecord(E.start_own, c->stream));
            E.start = E.start_own;
        }
        if (!rows_on_main)
            HIP_TRY(c, hipStreamWaitEvent(c->stream2, E.start, 0));
    }

    ibdg::RowsArgs sa;
    sa.panel = (const uint64_t *)c->panel.p;
    sa.stride = c->stride;
    sa.n_ids = c->n_ids;
    sa.rec_all = (const uint2 *)c->rec_all.p;
    sa.n_sites = c->n_sites;
    sa.lut = (const double *)c->lut.p;
    sa.alt_count = (const uint32_t *)c->alt_count.p;
    sa.pow_tab = (const double *)c->pow_tab.p;
    sa.fo = c->have_fo ? (const double *)c->fo.p : nullptr;
    sa.targets = d_targets;
    sa.t32 = c->pop_lut_ok ? (const uint4 *)c->t32.p : nullptr;
    sa.n_pairs = c->n_pairs;
    sa.cov_site = (const uint32_t *)c->cov_site.p;
    sa.n_cov = c->n_cov;
    sa.window = c->window;
    sa.n_win = c->n_win;
    sa.ld_mode = ld_mode ? 1 : 0;
    sa.af = nullptr;
    sa.site_ll = want_ll && !row_table ? (double *)c->site_ll.p : nullptr;
    sa.row_tab = row_table ? (double *)c->row_tab.p : nullptr;
    sa.win_ll = (double *)c->win_ll.p;

    if (use_pop) {
        // Comparison individuals in groups of MT share one workgroup (and the counts that do not
        // depend on them) in k_ld_popcount_mt; what is left over goes one per workgroup.
        const size_t MT = (size_t)ibdg::ld_popcount_mt_width();
        const bool mt_fits = ibdg::ld_popcount_lds_bytes(c->max_seg, c->wpg, c->ct_max + 1, c->tab_in_lds, c->seg_ring,
                                                         1) <= 150 * 1024;
        // Five or more comparison individuals: groups of IBDG_TG through the matrix cores (k_ld_mfma); the
        // last group may be short, fewer than mfma_min individuals take the counting kernels below.
        const size_t TGs = IBDG_TG;
        size_t n_gg = 0, T_g = 0;
        // (one group's partial sums and operands must stay modest: tiny windows over millions of rows go the old way)
        // (a group's partial sums: 16 doubles per window and half chunk -- per group of eight half chunks where the kernel's
        //  workgroups add their waves' sums up themselves, MfmaArgs::wg_sum)
        const bool mfma_wg_sum = c->opt_mfma_wg_sum && ibdg::ld_mfma_wg_sum(c->wpg, c->ct_max + 1, c->max_seg);
        const size_t ph_group = (size_t)c->n_win * (mfma_wg_sum ? (size_t)((2 * c->n_chunks + 7) / 8) * 128 : (size_t)c->n_chunks * 2 * 128) + 128;
        const size_t group_bytes = ph_group + (size_t)c->n_segs * 1024 + (size_t)c->n_win * 512;
        if (c->opt_mfma_targets && c->tab_in_lds && !dispatch_events && T >= (size_t)c->opt_mfma_min && (c->compact ? c->n_pairs_c : c->n_pairs) < (1u << 21) && c->n_segs < (1u << 21) &&     // (32-bit byte offsets of its buffer loads)
            group_bytes <= ((size_t)4 << 30) &&
            ibdg::ld_mfma_lds_bytes(c->wpg, c->ct_max + 1, c->max_seg) <= 64 * 1024) {
            n_gg = T / TGs;
            T_g = n_gg * TGs;
            if (T - T_g >= (size_t)c->opt_mfma_min) {
                n_gg++;
                T_g = T;
            }
        }
        const size_t T_cnt = T - T_g;      // comparison individuals of the counting kernels: [T_g, T)
        const size_t n_grp = (c->opt_multi_target && mt_fits && T_cnt >= MT) ? T_cnt / MT : 0, T_one = T_cnt - n_grp * MT;
        // target operands (1 KiB per segment and group) and partial sums (32 B per window, chunk and individual)
        // exist for one batch of groups at a time: about 1 GiB of operands, eight groups at most
        size_t gg_batch = n_gg;
        if (n_gg) {
            // (option "mfma_batch_groups": at most that many groups per launch, within 1/16 of the device's memory for each of
            //  the two buffers)
            const size_t per_group = (size_t)c->n_segs * 1024;
            const size_t mem_cap = std::max<size_t>((size_t)1 << 30, c->dev_mem_bytes / 16);
            size_t fit = per_group ? mem_cap / per_group : n_gg;
            const size_t fit_p = mem_cap / ph_group;     // partial sums
            fit = fit < fit_p ? fit : fit_p;
            const size_t cap_g = (size_t)std::max<long>(1, c->opt_mfma_batch);
            fit = fit < 1 ? 1 : (fit > cap_g ? cap_g : fit);
            gg_batch = fit < n_gg ? fit : n_gg;
        }
        // one comparison individual per workgroup: the counts of a haplotype word on the matrix cores where the larger
        // records leave the run's LDS image within reach (option "mx_counts")
        // ... and its power tables in LDS are plain doubles, rho^n as rho^n 2^(s n): s = the integer nearest to -log2 rho keeps
        // every entry, and every product of a rho and a sigma entry whose exponents add up to a window's reads, a normal number
        const double log2_rho = std::log2(c->eps / (1 - c->eps)), log2_sigma = std::log2(0.5 / (1 - c->eps));
        // (8 where the table allows it: the window end then makes the exponent up with one subtraction)
        const bool shift8 = (double)(c->ct_max + 1) * std::max(std::fabs(log2_rho + 8.0), std::fabs(log2_sigma)) <= 1000.0;
        const long rho_shift = shift8 ? 8 : std::lround(-log2_rho);
        const double per_read = std::max(std::fabs(log2_rho + (double)rho_shift), std::fabs(log2_sigma));
        const int mx_counts = c->opt_mx_counts && rho_shift >= 0 && rho_shift <= 40 &&
                              (!c->tab_in_lds || (double)(c->ct_max + 1) * per_read <= 1000.0) &&
                              ibdg::ld_popcount_lds_bytes(c->max_seg, c->wpg, c->ct_max + 1, c->tab_in_lds, c->seg_ring, 2) <= 150 * 1024;
        const size_t part_bytes = T * (size_t)c->n_win * c->n_chunks * 16;       // the counting kernels' sums per chunk; two halves taken in turn
        // (the single individuals' images in two halves like the other per-individual data, see above)
        const size_t img_slots = ahead_cap ? ibdg_ctx::TG_RING : 1;
        const size_t wt_cap0 = c->wtarget.cap, tw_cap0 = c->twords.cap;
        if (ensure(c, c->wtarget, img_slots * T_one * (size_t)c->n_win * 32) ||
            ensure(c, c->twords, img_slots * T_one * (size_t)c->n_segs * ibdg::ld_popcount_rec_bytes(mx_counts)) ||
            ensure(c, c->wtarget_mt, n_grp * (size_t)c->n_win * ibdg::ld_popcount_mt_wc_bytes()) ||
            ensure(c, c->twords_mt, n_grp * (size_t)c->n_segs * ibdg::ld_popcount_mt_rec_bytes()) ||
            ensure(c, c->partial, T_cnt ? 2 * part_bytes : 0) ||
            ensure(c, c->aimg, gg_batch * (size_t)c->n_segs * 1024) ||
            ensure(c, c->wc_slot, gg_batch * (size_t)c->n_win * 512) ||
            ensure(c, c->partial_h, gg_batch * ph_group))
            return 1;
        ibdg::PopArgs pa;
        pa.t32 = (const uint32_t *)(c->compact ? c->t32c.p : c->t32.p);
        pa.n_pairs = c->compact ? c->n_pairs_c : c->n_pairs;
        pa.n_chunks = c->n_chunks;
        pa.segs = (const ibdg::Seg *)c->segs.p;
        pa.n_segs = c->n_segs;
        pa.max_seg = c->max_seg;
        if (c->wtarget.cap != wt_cap0 || c->twords.cap != tw_cap0)
            c->wt_gen = 0;                    // new buffers: no images in them
        pa.rec_ready = (const uint32_t *)((const char *)c->twords.p + (size_t)c->tg_cur * (c->twords.cap / ibdg_ctx::TG_RING / 16 * 16));
        pa.wconst = (const ibdg::WinConst *)c->wconst.p;
        pa.n_win = c->n_win;
        pa.win_per_group = c->wpg;
        pa.run_begin = (const uint32_t *)c->runs.p;
        pa.n_runs = c->n_runs;
        pa.n_cgroups = (c->n_chunks + 7) / 8;
        pa.waves_per_group = (c->n_chunks + pa.n_cgroups - 1) / pa.n_cgroups;   // 40 chunks: 5 x 8; 9: 5 + 4; 2: 1 x 2
        pa.wc_ready = (const uint32_t *)((const char *)c->wtarget.p + (size_t)c->tg_cur * (c->wtarget.cap / ibdg_ctx::TG_RING / 16 * 16));
        pa.pow_1me = (const ibdg::PowEntry *)c->pow1.p;
        pa.pow_eps = (const ibdg::PowEntry *)c->pow2.p;
        pa.targets = sa.targets;
        pa.t_base = (uint32_t)T_g;
        pa.weight = d_weight;
        pa.lanes = (uint32_t)lanes;
        // Queued runs of single individuals (the timed steps of a shard, a caller's loop over the same comparison): this run's
        // finalising step -- one wave per window, 5 us, but a launch of its own with its gap and the event packet behind it:
        // a tenth of a step on an eighth of a chromosome -- is left to the NEXT run's --LD launch, whose first workgroups
        // do it on the way (the kernel boundary between the two launches is all the ordering it needs), and this run's
        // launch does the same for its predecessor.  The partial sums alternate between two halves of their buffer.
        // The IBD0 terms of this site list and background, once: a pass of the counting kernel that keeps every lane's weighted
        // product (p2_out) and its sums per chunk; the images it reads are those of the run's first individual.
        const bool p2_stale = c->p2_gen != c->sites_gen || c->p2_bg_gen != c->bg_gen || c->p2_mx != mx_counts;
        auto ibd0_pass = [&]() -> int {
            if (c->fin_pending && flush_finalize(c))
                return 1;
            if (settle_ready())
                return 1;
            if (ensure(c, c->p2w, (size_t)c->n_win * lanes * 8) || ensure(c, c->p2c, (size_t)c->n_win * c->n_chunks * 16) ||
                ensure(c, c->p2_tw, (size_t)c->n_segs * ibdg::ld_popcount_rec_bytes(mx_counts)) ||
                ensure(c, c->p2_wt, (size_t)c->n_win * 32))
                return 1;
            ibdg::PopArgs pp = pa;
            pp.rec_ready = (const uint32_t *)c->p2_tw.p;
            pp.wc_ready = (const uint32_t *)c->p2_wt.p;
            pp.weight = (const double *)c->base_w.p;
            pp.t_base = 0;
            pp.partial = (double *)c->p2c.p;
            pp.p2_out = (double *)c->p2w.p;
            pp.fin_prev = nullptr;
            pp.ibd1 = 0;
            pp.ring_slots = (uint32_t)c->seg_ring;
            pp.tab_len = c->ct_max + 1;
            pp.tab_in_lds = (uint32_t)c->tab_in_lds;
            pp.mx_counts = (uint32_t)mx_counts;
            pp.rho_shift = (uint32_t)rho_shift;
            pp.sum_dpp = (uint32_t)c->opt_sum_dpp;
            ibdg::launch_win_target(pp, 1, c->stream);
            if (ibdg::launch_ld_popcount(pp, 1, c->planes, c->stream))
                return fail(c, "[::] ERROR in ibdg_run: unsupported number of weight bit-planes %d", c->planes);
            c->p2_gen = c->sites_gen;
            c->p2_bg_gen = c->bg_gen;
            c->p2_mx = mx_counts;
            return 0;
        };
        // single individuals in the IBD1 form (counts on the matrix cores, tables in LDS): at once where the pass exists,
        // otherwise when the runs on this upload and background have added up
        bool ibd1 = false;
        if (T_one && mx_counts && c->tab_in_lds && c->opt_ibd0_after > 0) {
            if (c->ibd0_bg_gen != c->bg_gen) {
                c->ibd0_bg_gen = c->bg_gen;
                c->ibd0_runs = 0;
            }
            c->ibd0_runs += T_one;
            ibd1 = !p2_stale || (n_gg > 0) || c->ibd0_runs >= (uint64_t)c->opt_ibd0_after;
        }
        if (p2_stale && (n_gg > 0 || ibd1) && ibd0_pass())
            return 1;
        const bool fin_in_next = c->opt_fin_next && c->opt_async && T_one > 0 && T_one == T_cnt && n_gg == 0;
        if (c->fin_pending && (!fin_in_next || c->fin_count != (unsigned)T_cnt || c->fin_args.t_base != (uint32_t)T_g) &&
            flush_finalize(c))
            return 1;
        pa.partial = (double *)((char *)c->partial.p + (fin_in_next ? (size_t)c->part_half * part_bytes : 0));
        if (c->fin_pending) {
            pa.fin_prev = c->fin_args.partial;
            pa.n_refpanel = c->fin_args.n_refpanel;      // (of the run that left it: its entry of the ring)
            pa.win_ll = (double *)c->win_ll.p;
            pa.fin_p2c = c->fin_args.p2c;                // (non-null: that run was of the IBD1 form)
            pa.fin_p2w = c->fin_args.p2w;
            pa.fin_targets = c->fin_args.targets;
            c->fin_pending = false;
        }
        pa.ring_slots = (uint32_t)c->seg_ring;
        pa.tab_len = c->ct_max + 1;
        pa.tab_in_lds = (uint32_t)c->tab_in_lds;
        pa.mx_counts = (uint32_t)mx_counts;
        pa.rho_shift = (uint32_t)rho_shift;
        pa.sum_dpp = (uint32_t)c->opt_sum_dpp;
        ibdg::KernelEvents first, dominant, last;  // all null unless dispatch_events
        if (dispatch_events) {
            first.start = E.start_own;
            dominant.start = E.k_start;
            dominant.stop = E.k_stop;
            last.stop = E.ld_end;
        }
        if (n_gg) {
            if (settle_ready()) return 1;
            ibdg::MfmaArgs ma;
            ma.t32 = pa.t32;
            ma.n_pairs = pa.n_pairs;
            ma.n_chunks = pa.n_chunks;
            ma.segs = pa.segs;
            ma.n_segs = pa.n_segs;
            ma.wconst = pa.wconst;
            ma.n_win = pa.n_win;
            ma.run_begin = pa.run_begin;
            ma.n_runs = pa.n_runs;
            ma.win_per_group = pa.win_per_group;
            ma.max_seg = pa.max_seg;
            ma.aimg = (uint4 *)c->aimg.p;
            ma.wc_slot = (uint4 *)c->wc_slot.p;
            ma.pow_1me = pa.pow_1me;
            ma.pow_eps = pa.pow_eps;
            ma.pow_tau = (const ibdg::PowEntry *)c->pow3.p;
            ma.tab_len = pa.tab_len;
            ma.plain_tau = c->opt_mfma_plain_tau ? 1u : 0u;
            ma.targets = pa.targets;
            ma.base_weight = (const double *)c->base_w.p;
            ma.p2w = (const double *)c->p2w.p;
            ma.p2c = (const double *)c->p2c.p;
            ma.lanes = (uint32_t)lanes;
            ma.wg_sum = mfma_wg_sum ? 1u : 0u;
            {
                // one batch's partial sums: t1 [groups][windows][half chunks][16], t0 [groups][windows][half chunks], ov [groups][windows][16]
                const size_t nh = (size_t)c->n_chunks * 2;
                ma.part_t1 = (double *)c->partial_h.p;
                (void)nh;
            }
            for (size_t g0 = 0; g0 < n_gg; g0 += gg_batch) {
                const size_t nb = n_gg - g0 < gg_batch ? n_gg - g0 : gg_batch;
                ma.t_base = (uint32_t)(g0 * TGs);
                ma.n_targets = (uint32_t)((T_g - g0 * TGs) < nb * TGs ? (T_g - g0 * TGs) : nb * TGs);
                ibdg::launch_win_target_g(ma, (unsigned)nb, c->stream);
                if (g0 == 0) {
                    // the second stream (per-site values, window products; high priority) starts behind these
                    // two short kernels rather than beside them: it starved them (0.57 ms instead of 0.06)
                    HIP_TRY(c, hipEventRecord(E.prep, c->stream));
                    s2_after_prep = true;
                }
                if (ibdg::launch_ld_mfma(ma, (unsigned)nb, c->stream, ibdg::KernelEvents()))
                    return fail(c, "[::] ERROR in ibdg_run: the matrix-core --LD kernel could not be launched");
                ibdg::launch_ld_finalize_g(ma, (unsigned)nb, d_nrefpanel, (double *)c->win_ll.p, c->stream);
            }
        }
        // k_ld_mfma (4 waves per SIMD) leaves wave slots to the second stream: its kernels run in their fast forms
        // (also with a few individuals left to the counting kernels: T = 16 5.5 ms against 6.0; beside
        // k_ld_popcount_mt alone it makes no difference)
        side_fast = n_gg > 0;
        if (n_grp) {
            if (settle_ready()) return 1;
            ibdg::PopArgs pm = pa;
            pm.mx_counts = 0;
            pm.rec_ready = (const uint32_t *)c->twords_mt.p;
            pm.wc_ready = (const uint32_t *)c->wtarget_mt.p;
            ibdg::launch_win_target_mt(pm, (unsigned)n_grp, c->stream, first);
            first = ibdg::KernelEvents();
            if (ibdg::launch_ld_popcount_mt(pm, (unsigned)n_grp, c->stream, T_one ? ibdg::KernelEvents() : dominant))
                return fail(c, "[::] ERROR in ibdg_run: the multi-target --LD kernel could not be launched");
        }
        if (T_one) {
            pa.t_base = (uint32_t)(T_g + n_grp * MT);
            // (skipped when the previous run made the very same images: same prepared sites, same comparison individuals --
            // a caller that runs a comparison again, e.g. timed steps: one launch of ~10 us less per run, which on an
            // eighth of a chromosome is a tenth of the step)
            const bool wt_cached = same_inputs && c->wt_gen == c->sites_gen && c->wt_first == pa.t_base &&
                                   c->wt_count == (uint32_t)T_one && c->wt_mx == mx_counts && c->wt_slot == c->tg_cur &&
                                   c->wt_ibd1 == (int)ibd1 && !dispatch_events;
            pa.ibd1 = ibd1 ? 1u : 0u;
            // a new individual's images: on stream3 with its weights (under the --LD kernel of the run before) unless the
            // launch carries the run's start event (dispatch_events: that belongs on the main stream)
            const bool wt_ahead = need_ready && !dispatch_events;
            if (!wt_ahead && settle_ready()) return 1;
            if (wt_ahead && c->s3_gen != c->sites_gen) {
                // once per upload / change of layout: stream3's kernel reads the prepared sites (segments, window constants),
                // whose kernels were queued on the main stream
                HIP_TRY(c, hipEventRecord(c->ev_s3sync, c->stream));
                HIP_TRY(c, hipStreamWaitEvent(c->stream3, c->ev_s3sync, 0));
                c->s3_gen = c->sites_gen;
            }
            if (ibd1) {
                if (c->fb_gen != c->sites_gen) {
                    // once per site list (and layout): the three fragments per segment an individual's images select between
                    if (ensure(c, c->fragb, (size_t)c->n_segs * 72))
                        return 1;
                    if (!c->ev_fb)
                        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fb, hipEventDisableTiming));
                    hipStream_t fs = wt_ahead ? c->stream3 : c->stream;
                    ibdg::launch_frag_base(pa, (uint32_t *)c->fragb.p, fs);
                    HIP_TRY(c, hipEventRecord(c->ev_fb, fs));
                    HIP_TRY(c, hipStreamWaitEvent(fs == c->stream ? c->stream3 : c->stream, c->ev_fb, 0));   // (whichever makes the next images)
                    c->fb_gen = c->sites_gen;
                }
                pa.frag_base = (const uint32_t *)c->fragb.p;
            }
            if (!wt_cached)
                ibdg::launch_win_target(pa, (unsigned)T_one, wt_ahead ? c->stream3 : c->stream, first);
            if (settle_ready()) return 1;
            c->wt_slot = c->tg_cur;
            c->wt_gen = c->sites_gen;
            c->wt_first = pa.t_base;
            c->wt_count = (uint32_t)T_one;
            c->wt_mx = mx_counts;
            c->wt_ibd1 = (int)ibd1;
            c->last_count_unit = mx_counts ? (ibd1 ? 3 : 2) : 1;
            // (option "end_in_dispatch": the run's end event is the --LD kernel's own completion signal -- no event packet
            // of its own behind the kernel -- where that kernel is the run's last launch on the main stream)
            if (c->opt_end_in_dispatch && fin_in_next && !dispatch_events && T_one == T) {
                dominant.stop = E.ld_end;
                end_recorded = true;
            }
            if (ibdg::launch_ld_popcount(pa, (unsigned)T_one, c->planes, c->stream, dominant))
                return fail(c, "[::] ERROR in ibdg_run: unsupported number of weight bit-planes %d", c->planes);
        }
        ibdg::PopFinalArgs fa;
        fa.wconst = pa.wconst;
        fa.n_win = c->n_win;
        fa.n_chunks = c->n_chunks;
        fa.n_refpanel = d_nrefpanel;
        fa.win_ll = (double *)c->win_ll.p;
        if (T_cnt) {
            fa.partial = pa.partial;
            fa.t_base = (uint32_t)T_g;
            fa.halves = 0;
            if (ibd1) {                      // (whatever kernel made an individual's IBD1 sums)
                fa.p2c = (const double *)c->p2c.p;
                fa.p2w = (const double *)c->p2w.p;
                fa.targets = (const uint32_t *)(d_nrefpanel + T);      // (this run's individuals, from the longer ring: k_target_weights)
                fa.lanes = (uint32_t)lanes;
            }
            if (fin_in_next) {
                c->fin_pending = true;
                c->fin_args = fa;
                c->fin_count = (unsigned)T_cnt;
                c->fin_sites_gen = c->sites_gen;
                c->part_half ^= 1;
                if (dispatch_events)         // (no finalising launch to carry the run's end in its dispatch packet)
                    HIP_TRY(c, hipEventRecord(E.ld_end, c->stream));
            } else {
                ibdg::launch_ld_finalize(fa, (unsigned)T_cnt, c->stream, last);
            }
        }
    } else if (ld_mode) {
        if (settle_ready()) return 1;
        ibdg::LdArgs la;
        la.panel = sa.panel;
        la.stride = c->stride;
        la.rec_cov = (const uint2 *)c->rec_cov.p;
        la.n_cov = c->n_cov;
        la.window = c->window;
        la.n_win = c->n_win;
        la.n_groups = c->n_groups;
        la.lut = sa.lut;
        la.targets = sa.targets;
        la.weight = d_weight;
        la.n_refpanel = d_nrefpanel;
        la.win_ll = (double *)c->win_ll.p;
        la.t_base = 0;
        la.vals = nullptr;
        if (c->opt_variant == 3) {
            // reference order: per comparison individual, the per-individual products of every window,
            // then serial sums over the background list in the reference's order
            std::vector<uint32_t> order;
            if (!c->bg_order.empty()) {
                order = c->bg_order;               // checked against bg_count below
                std::vector<unsigned> cnt(c->n_ids, 0);
                for (uint32_t n : order) {
                    if (n >= c->n_ids)
                        return fail(c, "[::] ERROR in ibdg_run: background order names individual %u of %u", n, c->n_ids);
                    cnt[n]++;
                }
                for (unsigned n = 0; n < c->n_ids; ++n)
                    if (cnt[n] != (bg_count ? bg_count[n] : 1u))
                        return fail(c, "[::] ERROR in ibdg_run: background order and bg_count disagree for individual %u", n);
            } else {
                for (unsigned n = 0; n < c->n_ids; ++n)
                    for (unsigned k = bg_count ? bg_count[n] : 1u; k > 0; --k)
                        order.push_back(n);
            }
            if (ensure(c, c->vals, (size_t)c->n_win * lanes * 16) || ensure(c, c->order, order.size() * 4))
                return 1;
            HIP_TRY(c, hipMemcpyAsync(c->order.p, order.data(), order.size() * 4, hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));        // `order` is a local
            la.vals = (double2 *)c->vals.p;
            for (size_t t = 0; t < T; ++t) {
                la.t_base = (uint32_t)t;
                if (ibdg::launch_ld(la, 1, c->cpw, (unsigned)c->opt_waves, c->stream))
                    return fail(c, "[::] ERROR in ibdg_run: unsupported chunks_per_wave %d", c->cpw);
                ibdg::OrdArgs oa;
                oa.vals = la.vals;
                oa.lanes = (uint32_t)lanes;
                oa.n_win = c->n_win;
                oa.order = (const uint32_t *)c->order.p;
                oa.n_order = (uint32_t)order.size();
                oa.target = targets[t];
                oa.pu_id = pu_id;
                oa.win_ll = (double *)c->win_ll.p + t * (size_t)c->n_win * 3;
                ibdg::launch_ld_ordered_sum(oa, c->stream);
            }
        } else if (ibdg::launch_ld(la, (unsigned)T, c->cpw, (unsigned)c->opt_waves, c->stream))
            return fail(c, "[::] ERROR in ibdg_run: unsupported chunks_per_wave %d", c->cpw);
    }
    if (settle_ready()) return 1;
    if (rows_on_main) {
        if (join_streams(c)) return 1;           // an earlier run's kernel on stream2 may still write the results
        // alone on the chip: a wave per pair of windows (the default).  A resident grid whose waves walk over several windows -- even
        // with the next window's records kept in flight -- measured slower at every size (rows_blocks_per_cu 4..28:
        // 0.048-0.041 ms against 0.040)
        unsigned blocks = 0;
        if (c->opt_rows_blocks > 0)
            blocks = std::max<unsigned>(1u, (unsigned)((size_t)c->n_cu * c->opt_rows_blocks / T));
        ibdg::launch_rows_windows(sa, (unsigned)T, c->stream, blocks);
    }
    if (!dispatch_events && !end_recorded)
        HIP_TRY(c, hipEventRecord(E.ld_end, c->stream));

    if (!rows_on_main) {
        // stream2, queued after the critical path so that the --LD launches reach the device first
        if (s2_after_prep)
            HIP_TRY(c, hipStreamWaitEvent(c->stream2, E.prep, 0));
        HIP_TRY(c, hipEventRecord(E.s2_start, c->stream2));
        if (recount) {
            // beside the --LD kernel: few long-lived waves (opt_recount_blocks per CU), so that the recount does not
            // take the wave slots the --LD workgroups need -- it is bound by HBM, they by instruction issue
            ibdg::launch_alt_count((const uint64_t *)c->panel.p, c->stride, c->n_rows, (uint32_t *)c->alt_count.p,
                                   c->stream2, ld_mode ? (unsigned)(c->n_cu * c->opt_recount_blocks) : 0u);
            c->counts_valid = true;
            HIP_TRY(c, hipEventRecord(E.s2[0], c->stream2));
        }
        // the per-row values and the window products, one launch (k_rows_windows).  Beside the exponent-counting --LD
        // kernel, which holds every wave slot, it gets few long-lived workgroups (opt_site_blocks per CU, shared among
        // the targets): its gathers wait on memory either way, and the --LD workgroups keep their wave slots
        // (not when the alt counts are recounted in this run: the second stream's chain count -> rows is then the
        // longer one of the two, and its kernels should be short; and not beside the matrix-core kernel, which leaves
        // half of the wave slots free)
        const bool shadow = ld_mode && !recount && !side_fast;
        unsigned row_blocks = 0;
        if (shadow && c->opt_site_blocks > 0)
            row_blocks = std::max<unsigned>(1u, (unsigned)((size_t)c->n_cu * c->opt_site_blocks / T));
        if (row_table)
            ibdg::launch_row_table(sa, c->stream2);
        ibdg::launch_rows_windows(sa, (unsigned)T, c->stream2, row_blocks);
        HIP_TRY(c, hipEventRecord(E.s2[2], c->stream2));
        c->last_s2 = E.s2[2];
        c->s2_pending = true;
        c->tg_s2[c->tg_cur] = E.s2[2];
        c->tg_s2_pending[c->tg_cur] = true;
        if (!ahead_cap)
            for (int h = 1; h < ibdg_ctx::TG_RING; ++h) {
                c->tg_s2[h] = E.s2[2];
                c->tg_s2_pending[h] = true;
            }
    }
    // (who reads this run's slot of the per-individual buffers on the main stream)
    for (int h = ahead_cap ? c->tg_cur : 0; h <= (ahead_cap ? c->tg_cur : ibdg_ctx::TG_RING - 1); ++h) {
        c->tg_main[h] = E.ld_end;
        c->tg_main_pending[h] = true;
    }
    HIP_TRY(c, hipGetLastError());
    c->ev_head = ev_slot;
    ++c->runs_done;
    c->chain_ok = true;
    if (!c->opt_async && quiesce(c))
        return 1;
    c->n_targets = T;
    c->have_results = true;
    c->res_site_mode = row_table ? 2 : (int)c->opt_site_results;
    return 0;
}

static int fetch(ibdg_ctx *c, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return 0;
    HIP_TRY(c, hipSetDevice(c->device));
    if (join_streams(c)) return 1;
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    return quiesce(c);
}

int ibdg_get_site_af(ibdg_ctx *c, double *af)
{
    if (!c) return 1;
    if (!c->have_results) return fail(c, "[::] ERROR in ibdg_get_site_af: no results (call ibdg_run)");
    if (!c->counts_valid) return fail(c, "[::] ERROR in ibdg_get_site_af: alt counts not computed yet");
    // made when asked for: it depends on the panel row (or the -A value) only, and no run needs it
    HIP_TRY(c, hipSetDevice(c->device));
    if (ensure(c, c->af, c->n_sites * 8)) return 1;
    if (join_streams(c)) return 1;
    ibdg::RowsArgs ra = {};
    ra.rec_all = (const uint2 *)c->rec_all.p;
    ra.n_sites = c->n_sites;
    ra.n_ids = c->n_ids;
    ra.alt_count = (const uint32_t *)c->alt_count.p;
    ra.fo = c->have_fo ? (const double *)c->fo.p : nullptr;
    ra.af = (double *)c->af.p;
    ibdg::launch_site_af(ra, c->stream);
    HIP_TRY(c, hipGetLastError());
    return fetch(c, af, c->af.p, c->n_sites * 8);
}

int ibdg_get_site_ll(ibdg_ctx *c, size_t t, double *out)
{
    if (!c) return 1;
    if (!c->have_results || t >= c->n_targets) return fail(c, "[::] ERROR in ibdg_get_site_ll: no results for target %zu", t);
    if (c->res_site_mode == 0) return fail(c, "[::] ERROR in ibdg_get_site_ll: the run kept no per-site results (option site_results)");
    if (c->res_site_mode == 2) {
        // the run kept one table of the rows' values for all its comparison individuals: this one's per-site table from it
        if (t >= c->prev_targets.size()) return fail(c, "[::] ERROR in ibdg_get_site_ll: no results for target %zu", t);
        HIP_TRY(c, hipSetDevice(c->device));
        if (join_streams(c)) return 1;
        ibdg::RowsArgs ra = {};
        ra.panel = (const uint64_t *)c->panel.p;
        ra.stride = c->stride;
        ra.n_ids = c->n_ids;
        ra.rec_all = (const uint2 *)c->rec_all.p;
        ra.n_sites = c->n_sites;
        ra.lut = (const double *)c->lut.p;
        ra.t32 = c->pop_lut_ok ? (const uint4 *)c->t32.p : nullptr;
        ra.n_pairs = c->n_pairs;
        ra.row_tab = (double *)c->row_tab.p;
        ibdg::launch_site_expand(ra, c->prev_targets[t], (double *)c->site_ll.p, c->stream);
        HIP_TRY(c, hipGetLastError());
        return fetch(c, out, c->site_ll.p, c->n_sites * 24);
    }
    return fetch(c, out, (const char *)c->site_ll.p + t * c->n_sites * 24, c->n_sites * 24);
}

int ibdg_get_window_ll(ibdg_ctx *c, size_t t, double *out)
{
    if (!c) return 1;
    if (!c->have_results || t >= c->n_targets) return fail(c, "[::] ERROR in ibdg_get_window_ll: no results for target %zu", t);
    return fetch(c, out, (const char *)c->win_ll.p + t * (size_t)c->n_win * 24, (size_t)c->n_win * 24);
}

int ibdg_get_window_ll_all(ibdg_ctx *c, double *out)
{
    if (!c) return 1;
    if (!c->have_results) return fail(c, "[::] ERROR in ibdg_get_window_ll_all: no results (call ibdg_run)");
    return fetch(c, out, c->win_ll.p, c->n_targets * (size_t)c->n_win * 24);
}

int ibdg_get_alt_counts(ibdg_ctx *c, size_t first_row, size_t n, uint32_t *out)
{
    if (!c) return 1;
    if (!c->counts_valid) return fail(c, "[::] ERROR in ibdg_get_alt_counts: counts not computed yet");
    if (first_row + n > c->n_rows) return fail(c, "[::] ERROR in ibdg_get_alt_counts: range outside the panel");
    return fetch(c, out, (const char *)c->alt_count.p + first_row * 4, n * 4);
}

int ibdg_run_ms(ibdg_ctx *c, unsigned back, float out[5])
{
    if (!c || !out) return 1;
    if (back + 1 >= (unsigned)ibdg_ctx::EV_RING || (long)back >= c->runs_done)
        return fail(c, "[::] ERROR in ibdg_run_ms: no timing kept for the run %u calls back", back);
    const ibdg_ctx::EvSet &E = c->evs[(c->ev_head + ibdg_ctx::EV_RING - (int)back) % ibdg_ctx::EV_RING];
    if (quiesce(c)) return 1;
    float v, w;
    HIP_TRY(c, hipEventElapsedTime(&v, E.start, E.ld_end));
    out[1] = out[4] = 0.f;
    if (E.rows_on_main) {                        // non-LD: one kernel between the two events of the main stream
        out[0] = out[2] = v;
        out[3] = 0.f;
        return 0;
    }
    HIP_TRY(c, hipEventElapsedTime(&w, E.start, E.s2[2]));
    out[0] = v > w ? v : w;                  // the run ends when both streams are done
    out[3] = E.ld ? v : 0.f;
    if (E.recount)
        HIP_TRY(c, hipEventElapsedTime(&out[1], E.s2_start, E.s2[0]));
    HIP_TRY(c, hipEventElapsedTime(&v, E.recount ? E.s2[0] : E.s2_start, E.s2[2]));
    out[2] = v;                              // per-row values and window products are one kernel
    return 0;
}

int ibdg_run_kernel_ms(ibdg_ctx *c, unsigned back, float *ms)
{
    if (!c || !ms) return 1;
    if (back + 1 >= (unsigned)ibdg_ctx::EV_RING || (long)back >= c->runs_done)
        return fail(c, "[::] ERROR in ibdg_run_kernel_ms: no timing kept for the run %u calls back", back);
    const ibdg_ctx::EvSet &E = c->evs[(c->ev_head + ibdg_ctx::EV_RING - (int)back) % ibdg_ctx::EV_RING];
    if (!E.has_kernel_times)
        return fail(c, "[::] ERROR in ibdg_run_kernel_ms: that run did not use the exponent-counting --LD kernel");
    if (quiesce(c)) return 1;
    HIP_TRY(c, hipEventElapsedTime(ms, E.k_start, E.k_stop));
    return 0;
}

int ibdg_last_run_ms(ibdg_ctx *c, float out[5])
{
    if (!c || !out) return 1;
    if (c->runs_done == 0) {
        for (int i = 0; i < 5; ++i) out[i] = 0.f;
        return 0;
    }
    return ibdg_run_ms(c, 0, out);
}

int ibdg_last_ld_variant(const ibdg_ctx *c) { return c ? c->last_variant : 0; }

int ibdg_last_count_unit(const ibdg_ctx *c) { return c ? c->last_count_unit : 0; }

int ibdg_ld_layout(const ibdg_ctx *c)
{
    if (!c || !c->sites_valid || !c->pop_sites_ok)
        return 0;
    return c->compact ? 2 : 1;
}

int ibdg_set_option(ibdg_ctx *c, const char *name, long value)
{
    if (!c || !name) return 1;
    if (!strcmp(name, "count_in_run")) { c->opt_count_in_run = value != 0; return 0; }
    if (!strcmp(name, "multi_target")) { c->opt_multi_target = value != 0; return 0; }
    if (!strcmp(name, "mfma_targets")) { c->opt_mfma_targets = value != 0; return 0; }
    if (!strcmp(name, "mfma_plain_tau")) { c->opt_mfma_plain_tau = value != 0; return 0; }
    if (!strcmp(name, "mfma_min")) {
        if (value < 1 || value > IBDG_TG) return fail(c, "[::] ERROR in ibdg_set_option: mfma_min must be 1..%d", IBDG_TG);
        c->opt_mfma_min = value;
        return 0;
    }
    if (!strcmp(name, "guided_runs")) { c->opt_guided = value; return 0; }
    if (!strcmp(name, "compact_align")) {
        if (value != 1 && value != 2 && value != 4 && value != 8 && value != 16 && value != 32)
            return fail(c, "[::] ERROR in ibdg_set_option: compact_align must be 1, 2, 4, 8, 16 or 32");
        c->opt_compact_align = value;
        return 0;
    }
    if (!strcmp(name, "mfma_wg_sum")) { c->opt_mfma_wg_sum = value != 0; return 0; }
    if (!strcmp(name, "mfma_batch_groups")) { c->opt_mfma_batch = value < 1 ? 1 : (value > 64 ? 64 : value); return 0; }
    if (!strcmp(name, "ibd0_after")) { c->opt_ibd0_after = value < 0 ? 0 : value; return 0; }
    if (!strcmp(name, "end_in_dispatch")) { c->opt_end_in_dispatch = value != 0; return 0; }
    if (!strcmp(name, "prep_ahead")) { c->opt_prep_ahead = value != 0; return 0; }
    if (!strcmp(name, "dispatch_events")) { c->opt_dispatch_events = value != 0; return 0; }
    if (!strcmp(name, "async")) { c->opt_async = value != 0; return 0; }
    if (!strcmp(name, "dev_inputs_ready")) { c->opt_dev_inputs_ready = value != 0; return 0; }
    if (!strcmp(name, "staged_upload")) { c->opt_staged_upload = value != 0; return 0; }
    if (!strcmp(name, "stage_workers")) {
        if (value < 1 || value > ibdg_ctx::STAGE_WORKERS) return fail(c, "[::] ERROR in ibdg_set_option: stage_workers must be 1..%d", ibdg_ctx::STAGE_WORKERS);
        c->opt_stage_workers = value; return 0;
    }
    if (!strcmp(name, "compact_tiles")) {
        if (value < -1 || value > 1) return fail(c, "[::] ERROR in ibdg_set_option: compact_tiles must be -1 (never), 0 (auto) or 1 (always)");
        c->opt_compact = value; return 0;
    }
    if (!strcmp(name, "finalize_in_next")) {
        if (value < 0 || value > 1) return fail(c, "[::] ERROR in ibdg_set_option: finalize_in_next must be 0 or 1");
        if (join_streams(c)) return 1;
        c->opt_fin_next = value; return 0;
    }
    if (!strcmp(name, "sum_dpp")) {
        if (value < 0 || value > 1) return fail(c, "[::] ERROR in ibdg_set_option: sum_dpp must be 0 or 1");
        c->opt_sum_dpp = value; return 0;
    }
    if (!strcmp(name, "mx_counts")) {
        if (value < 0 || value > 1) return fail(c, "[::] ERROR in ibdg_set_option: mx_counts must be 0 or 1");
        c->opt_mx_counts = value; return 0;
    }
    if (!strcmp(name, "reserve_compact")) {
        if (value < 0 || value > 1) return fail(c, "[::] ERROR in ibdg_set_option: reserve_compact must be 0 or 1");
        c->opt_reserve_compact = value; return 0;
    }
    if (!strcmp(name, "compact_density")) {
        if (value < 1 || value > 1000000) return fail(c, "[::] ERROR in ibdg_set_option: compact_density must be 1..1000000");
        c->opt_compact_density = value; return 0;
    }
    if (!strcmp(name, "compact_targets")) {
        if (value < 1 || value > 65536) return fail(c, "[::] ERROR in ibdg_set_option: compact_targets must be 1..65536");
        c->opt_compact_targets = value; return 0;
    }
    if (!strcmp(name, "site_results")) {
        if (value < 0 || value > 1) return fail(c, "[::] ERROR in ibdg_set_option: site_results must be 0 or 1");
        c->opt_site_results = value; return 0;
    }
    if (!strcmp(name, "rows_blocks_per_cu")) {
        if (value < 0 || value > 128) return fail(c, "[::] ERROR in ibdg_set_option: rows_blocks_per_cu must be 0..128");
        c->opt_rows_blocks = value; return 0;
    }
    if (!strcmp(name, "site_blocks_per_cu")) {
        if (value < 0 || value > 128) return fail(c, "[::] ERROR in ibdg_set_option: site_blocks_per_cu must be 0..128");
        c->opt_site_blocks = value; return 0;
    }
    if (!strcmp(name, "recount_blocks_per_cu")) {
        if (value < 0 || value > 128) return fail(c, "[::] ERROR in ibdg_set_option: recount_blocks_per_cu must be 0..128");
        c->opt_recount_blocks = value; return 0;
    }
    if (!strcmp(name, "chunks_per_wave")) {
        if (value < 0 || value > 5) return fail(c, "[::] ERROR in ibdg_set_option: chunks_per_wave must be 0..5");
        c->opt_cpw = value; return 0;
    }
    if (!strcmp(name, "waves_per_block")) {
        if (value < 1 || value > 8) return fail(c, "[::] ERROR in ibdg_set_option: waves_per_block must be 1..8");
        c->opt_waves = value; return 0;
    }
    if (!strcmp(name, "ld_variant")) {
        if (value < 0 || value > 3) return fail(c, "[::] ERROR in ibdg_set_option: ld_variant must be 0 (auto), 1 (strict), 2 (exponent counting) or 3 (reference order)");
        c->opt_variant = value; return 0;
    }
    if (!strcmp(name, "ring_slots")) {
        if (value != 2 && value != 3 && value != 4 && value != 8) return fail(c, "[::] ERROR in ibdg_set_option: ring_slots must be 2, 3, 4 or 8");
        c->opt_ring = value; return 0;
    }
    if (!strcmp(name, "record_lds_bytes")) {
        if (value < 1024 || value > 96 * 1024) return fail(c, "[::] ERROR in ibdg_set_option: record_lds_bytes must be 1024..98304");
        c->opt_recbytes = value; return 0;
    }
    if (!strcmp(name, "windows_per_wave")) {
        if (value < 1 || value > 65536) return fail(c, "[::] ERROR in ibdg_set_option: windows_per_wave must be 1..65536");
        c->opt_wpg = value; c->opt_wpg_fixed = true; return 0;
    }
    return fail(c, "[::] ERROR in ibdg_set_option: unknown option '%s'", name);
}

int ibdg_set_background_order(ibdg_ctx *c, const uint32_t *ids, size_t n)
{
    if (!c) return 1;
    if (n && !ids) return fail(c, "[::] ERROR in ibdg_set_background_order: ids is NULL");
    c->bg_order.assign(ids, ids + n);
    return 0;
}

int ibdg_selftest(const char *what)
{
    if (!what)
        return 1;
    if (!strcmp(what, "wait_info")) {
        // the three ways the bounded poll ends, without a device: the word arrives (from another thread); the producer
        // reports completion / failure without having written it; neither -- the wall-clock bound
        volatile uint32_t flag = 0;
        std::thread setter([&]() {
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
            flag = 7;
        });
        const int arrived = poll_seq(&flag, 7u, 5.0, []() { return 0; });
        setter.join();
        flag = 0;
        const int done = poll_seq(&flag, 9u, 5.0, []() { return 1; });
        const int failed = poll_seq(&flag, 9u, 5.0, []() { return 5; });
        const auto t0 = std::chrono::steady_clock::now();
        const int timed_out = poll_seq(&flag, 9u, 0.05, []() { return 0; });
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return (arrived == 0 && done == 1 && failed == -5 && timed_out == 2 && waited >= 0.05 && waited < 2.0) ? 0 : 2;
    }
    return 1;
}

int ibdg_sync(ibdg_ctx *c)
{
    if (!c) return 1;
    return quiesce(c);
}

}  // extern "C"
