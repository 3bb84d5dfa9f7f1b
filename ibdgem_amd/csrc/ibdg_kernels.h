// ibdg_kernels.h -- kernel argument blocks and launch wrappers (internal).
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stddef.h>
#include <stdint.h>

namespace ibdg {

// per-row values + window products (k_rows_windows)
struct RowsArgs {
    const uint64_t *panel;      // [n_rows][stride]
    uint32_t stride;            // u64 words per device row
    uint32_t n_ids;
    const uint2 *rec_all;       // [n_sites] {row, lut byte offset}
    size_t n_sites;
    const double *lut;          // [(M+1)^2][3]
    const uint32_t *alt_count;  // [n_rows]
    const double *pow_tab;      // [2*n_ids+1][2] = pow(1-f,2), pow(f,2) at f=k/(2N)
    const double *fo;           // NULL or [n_sites][3] = f, pow(1-f,2), pow(f,2) (-A)
    const uint32_t *targets;    // [T]
    const uint4 *t32;           // tile-transposed panel (NULL if not built) and its pairs per chunk
    uint32_t n_pairs;
    const uint32_t *cov_site;   // [n_cov] site index of each covered row
    uint32_t n_cov;
    uint32_t window;
    uint32_t n_win;
    int ld_mode;                // 1: of the window products only LIBD2 is written
    double *af;                 // [n_sites]: written by k_site_af only (ibdg_get_site_af)
    double *site_ll;            // [T][n_sites][3], or NULL: per-site results not wanted
    double *row_tab;            // [n_sites][4] {LIBD0, LIBD1 under genotype 0, 1, 2}: k_row_table / k_site_expand only
    double *win_ll;             // [T][n_win][3]
};

struct LdArgs {
    const uint64_t *panel;
    uint32_t stride;
    const uint2 *rec_cov;       // [n_cov] {row, lut byte offset} of covered rows
    uint32_t n_cov;
    uint32_t window;
    uint32_t n_win;
    uint32_t n_groups;          // chunk groups per row = stride / (2*CPW)
    const double *lut;
    const uint32_t *targets;    // [T]
    const double *weight;       // [T][n_groups*CPW*64] background multiplicity (0 = excluded)
    const int *n_refpanel;      // [T]
    double *win_ll;             // [T][n_win][3]
    uint32_t t_base;            // comparison individual of blockIdx.y == 0
    double2 *vals;              // reference-order mode: [n_win][lanes] {P2, (Q00+Q01)+Q10)+Q11} per background
                                // individual of ONE comparison individual, written instead of the window sums
};

// Reference-order mode (ld_variant 3): the window averages of src/ibdgem.c:736-753 summed serially over
// the background list in the reference's own order, from the per-individual products of k_ld_window.
struct OrdArgs {
    const double2 *vals;        // [n_win][lanes]
    uint32_t lanes, n_win;
    const uint32_t *order;      // [n_order] background individuals in list order (duplicates kept)
    uint32_t n_order;
    uint32_t target;            // individual excluded as the comparison individual
    int pu_id;                  // individual excluded as the pileup's own, or -1
    double *win_ll;             // this comparison individual's [n_win][3]
};
void launch_ld_ordered_sum(const OrdArgs &a, hipStream_t st);

// ---- fast --LD variant: exponent counting on the tile-transposed panel -------------------
// One segment = the covered rows of ONE window that fall into ONE 32-row tile.
// cov[k]/alt[k]: bit j set <=> row 32*tile+j is such a row and bit k of its
// n_ref+n_alt (resp. n_alt) is set.
struct Seg {
    uint32_t tile;
    uint32_t win;        // window index
    uint32_t last;       // 1 = last segment of its window
    uint32_t flags;      // bits 0-2 ring slot of the next segment's tile pair (relative to its run's first
                         // pair, modulo the ring depth), bit 3 its tile parity, bits 4-11 pairs between this
                         // segment and the next, bit 12 weight planes beyond cov 0-2 / alt 0-1 present,
                         // bit 13 last segment of its window, bits 16-23 / 24-31 number of cov / alt planes
    uint32_t cov[8];
    uint32_t alt[8];
};                       // 80 bytes

// rho^n or sigma^n (rho = eps/(1-eps), sigma = 1/(2(1-eps))) as m * 2^e with m in [0.5,1]
struct PowEntry {
    double m;
    int32_t e;
    int32_t pad;
};

// per window, target independent
struct WinConst {
    double mK;           // K' = prod C(cov,n_ref) * (1-eps)^cov_total = mK * 2^eK
    int32_t eK;
    uint32_t cov_total;  // sum of n_ref+n_alt over the window's rows
    uint32_t alt_total;  // sum of n_alt
    uint32_t seg_begin;  // first segment of the window
};                       // 24 bytes

struct PopArgs {
    const uint32_t *t32;        // [n_chunks][n_pairs][64] uint4: tile-transposed panel (see k_transpose32)
    uint32_t n_pairs;           // tile pairs per chunk, a multiple of 4 (whole octs)
    uint32_t n_chunks;          // chunks that hold individuals
    const Seg *segs;
    uint32_t n_segs;
    uint32_t max_seg;           // most segments in one run of win_per_group windows (LDS sizing)
    const uint32_t *rec_ready;  // [T][n_segs][8] LDS-ready segment records (written by k_win_target)
    const WinConst *wconst;     // [n_win + 1] (the extra entry carries seg_begin = n_segs)
    uint32_t n_win;
    uint32_t win_per_group;     // most windows in one run (LDS sizing)
    const uint32_t *run_begin;  // [n_runs + 1] first window of every run; workgroup = (run, group of 8 chunks)
    uint32_t n_runs;
    uint32_t n_cgroups;         // groups of waves_per_group chunks (one workgroup each per run)
    uint32_t waves_per_group;   // 1..8: chunks = waves per workgroup, balanced over the groups
    const uint32_t *wc_ready;   // [T][n_win][8] LDS-ready window constants (written by k_win_target)
    const PowEntry *pow_1me;    // [(max cov_total)+1]
    const PowEntry *pow_eps;
    const uint32_t *targets;    // [T]
    uint32_t t_base;            // first comparison individual of this launch (blockIdx.z / group 0)
    const double *weight;       // [T][lanes] background multiplicity (0 = excluded)
    uint32_t lanes;             // stride of weight per target
    double *partial;            // [T][n_win][n_chunks][2]
    uint32_t ring_slots;        // 4 or 8 tile pairs of LDS ring per wave
    uint32_t tab_len;           // entries per power table = (max cov_total) + 1
    uint32_t tab_in_lds;        // 1: the workgroup keeps both tables in LDS
    uint32_t rho_shift = 0;     // ... whose rho^n table (in LDS) holds rho^n * 2^(rho_shift n) as plain doubles
    const double *fin_prev = nullptr;   // k_ld_popcount: the partial sums of the PREVIOUS run of the same shape, finalised by this launch's first workgroups
    const int *n_refpanel = nullptr;    // ... what that takes (k_ld_finalize's arguments)
    double *win_ll = nullptr;
    uint32_t sum_dpp = 0;       // ... and whose wave sums exchange by DPP moves instead of ds_swizzle
    uint32_t mx_counts = 0;     // 1: k_ld_popcount takes the counts of a haplotype word on the matrix cores (records of 128 bytes)
    double *p2_out = nullptr;   // k_ld_popcount: [n_win][lanes] every lane's weighted product of its individual's OWN genotype
                                // factors (src/ibdgem.c:715, :743) -- what does not depend on the comparison individual; the
                                // matrix-core kernel's launches take it from there (once per site list and background)
    // Round 5: with that pass made, a single comparison individual's launch leaves the IBD0 terms out altogether
    // (k_ld_popcount<.., IBD1 = true>): its matrix instructions return the table exponents of the four IBD1 products
    // themselves, and the finalising step takes IBD0 from the pass (ibd0_from_pass in ibdg_ld_dev.h).
    uint32_t ibd1 = 0;          // 1: images (k_win_target_mx) and kernel of that form; needs mx_counts and tab_in_lds
    const double *fin_p2c = nullptr, *fin_p2w = nullptr;   // fin_prev's run was of that form: the pass's sums per chunk / per individual
    const uint32_t *fin_targets = nullptr;                 // ... and its comparison individuals (that run's entry of the ring)
    const uint32_t *frag_base = nullptr;   // ibd1: [n_segs][3][6] the fragments COV, F0, F1 of the site list (k_frag_base); null: k_win_target_mx builds everything
};

// events a dispatch updates with its own start / stop time (either may be null)
struct KernelEvents {
    hipEvent_t start = nullptr, stop = nullptr;
};

struct PopFinalArgs {
    const WinConst *wconst;
    const double *partial;      // [T][n_win][n_chunks][2], or per half chunk [T][n_win][2 * n_chunks][2] (halves = 1)
    uint32_t n_win, n_chunks;
    const int *n_refpanel;
    double *win_ll;
    uint32_t t_base = 0;        // first comparison individual of the launch
    uint32_t halves = 0;        // 1: partial sums per half chunk (k_ld_mfma); a chunk's sum is half 0 + half 1
    uint32_t p_t0 = 0;          // comparison individual whose partial sums come first in `partial`
    const double *p2c = nullptr, *p2w = nullptr;   // non-null: IBD0 from the one pass over the site list (PopArgs::ibd1 runs)
    const uint32_t *targets = nullptr;
    uint32_t lanes = 0;
};

// ---- many comparison individuals against one panel: the G(x,t) sums as integer matrix products ----------
// (ibdg_ld_mfma.hip; BASELINE.json configs[4]).  Groups of IBDG_TG comparison individuals.
#define IBDG_TG 15
struct MfmaArgs {
    const uint32_t *t32;        // tile-transposed panel (PopArgs::t32)
    uint32_t n_pairs, n_chunks;
    const Seg *segs;
    uint32_t n_segs;
    const WinConst *wconst;     // [n_win + 1]
    uint32_t n_win;
    const uint32_t *run_begin;  // [n_runs + 1]
    uint32_t n_runs, win_per_group;
    uint32_t max_seg;           // most segments in one run (LDS sizing)
    uint4 *aimg;                // [groups][n_segs][64] target operand of every segment (k_win_target_g)
    uint4 *wc_slot;             // [groups][n_win][16][2] per comparison individual: U_t0, U_t1 mantissas | their exponents
                                // (U_t = rho^-<t,alt> sigma^<t,cov>)
    const PowEntry *pow_1me, *pow_eps, *pow_tau;    // rho^n, sigma^n, tau^n = (rho / sigma^2)^n
    uint32_t tab_len;
    uint32_t plain_tau;         // 1: windows whose powers tau^G are all ordinary doubles look them up as such (8 bytes); 0: never
    const uint32_t *targets;    // [T]
    uint32_t t_base;            // first comparison individual of group 0 of this launch
    uint32_t n_targets;         // comparison individuals of this launch (groups of IBDG_TG, the last may be short)
    const double *base_weight;  // [lanes] background multiplicity without the comparison individual's exclusion
    // partial sums of the launch's groups per window and half chunk (2 * n_chunks of them), see k_ld_mfma:
    double *part_t1;            // [groups][n_win][2 * n_chunks][16 slots]  IBD1 sums
    // IBD0 does not depend on the comparison individual except for its own exclusion: the products and their sums per chunk of
    // 64 individuals come from ONE pass of k_ld_popcount per site list and background (PopArgs::p2_out and its partial sums)
    uint32_t wg_sum = 0;        // 1: part_t1 is [groups][n_win][groups of eight half chunks][16]: a workgroup adds its waves' sums up (ld_mfma_wg_sum)
    uint32_t n_groups = 1;      // groups of the launch (set by launch_ld_mfma: the kernel deals its workgroups itself)
    const double *p2w;          // [n_win][lanes] weight x product of every background individual
    const double *p2c;          // [n_win][n_chunks][2]: [0] = the chunk's sum of p2w
    uint32_t lanes;
};
size_t ld_mfma_lds_bytes(uint32_t win_per_group, uint32_t tab_len, uint32_t max_seg);
int ld_mfma_wg_sum(uint32_t win_per_group, uint32_t tab_len, uint32_t max_seg);
void launch_win_target_g(const MfmaArgs &a, unsigned n_groups, hipStream_t st);
int launch_ld_mfma(const MfmaArgs &a, unsigned n_groups, hipStream_t st, KernelEvents ev);
// win_ll[t][w][0..1] of the launch's comparison individuals from those partial sums (src/ibdgem.c:751-752)
void launch_ld_finalize_g(const MfmaArgs &a, unsigned n_groups, const int *n_refpanel, double *win_ll, hipStream_t st);

void launch_transpose32(const uint64_t *panel, uint32_t stride, size_t n_rows, uint32_t n_chunks,
                        uint32_t n_pairs, uint32_t *t32, hipStream_t st);
// the compacted tiles of a site list: covered row j -> virtual row (j / window) * win_rows + j % window (win_rows = window:
// back to back; 32 * ceil(window / 32): every window on a tile boundary); n_pairs tile pairs per chunk (whole octs, zero
// beyond the last window)
void launch_gather_transpose32(const uint64_t *panel, uint32_t stride, const uint2 *rec_cov, uint32_t n_cov,
                               uint32_t window, uint32_t win_rows, uint32_t n_chunks, uint32_t n_pairs, uint32_t *t32,
                               hipStream_t st);
void launch_win_target(const PopArgs &a, unsigned n_targets, hipStream_t st, KernelEvents ev = {});
void launch_frag_base(const PopArgs &a, uint32_t *frag_base, hipStream_t st);      // PopArgs::frag_base of a site list
size_t ld_popcount_rec_bytes(int mx_counts);      // bytes of one segment record in rec_ready
int launch_ld_popcount(const PopArgs &a, unsigned n_targets, int planes, hipStream_t st, KernelEvents ev = {});
size_t ld_popcount_lds_bytes(uint32_t max_seg, uint32_t win_per_group, uint32_t tab_len, int tab_in_lds,
                             int ring_slots, int multi_target);
// groups of ld_popcount_mt_width() comparison individuals per workgroup (shared target-independent counts)
int ld_popcount_mt_width(void);
size_t ld_popcount_mt_rec_bytes(void);
size_t ld_popcount_mt_wc_bytes(void);
void launch_win_target_mt(const PopArgs &a, unsigned n_groups, hipStream_t st, KernelEvents ev = {});
int launch_ld_popcount_mt(const PopArgs &a, unsigned n_groups, hipStream_t st, KernelEvents ev = {});
void launch_ld_finalize(const PopFinalArgs &a, unsigned n_targets, hipStream_t st, KernelEvents ev = {});

// ---- per-comparison site preparation on the device (ibdg_prep.hip) ------------------------
// What the host learns from it (one small device-to-host copy per stage).
struct PrepInfo {
    uint32_t n_cov;          // covered rows (n_ref+n_alt >= 1)
    uint32_t err_row_site;   // smallest site whose row_index lies outside the panel, 0xffffffff if none
    uint32_t err_cov_site;   // smallest site with n_ref+n_alt > max_cov, 0xffffffff if none
    uint32_t out_of_order;   // 1: some covered row is not behind its predecessor in the panel
    uint32_t first_row, last_row;   // panel rows of the first / last covered site
    uint32_t n_segs;         // (window, 32-row tile) segments
    uint32_t ct_max;         // most reads in one window
    uint32_t max_seg;        // most segments in one run of windows
    uint32_t adv_overflow;   // 1: two consecutive segments of a run are more than 255 tile pairs apart
    uint32_t seq;            // host mirror only: written last, the number of the hand-over the words above belong to
    uint32_t pad;
};

// a number in the x87 extended format, normalised: value = m / 2^64 * 2^e, m in [2^63, 2^64)
struct WinRaw {
    uint64_t m;
    int32_t e;
    int32_t pad;
};

struct PrepSiteArgs {
    const uint32_t *row_index;  // device; NULL: site s is panel row s
    const uint8_t *n_ref, *n_alt;
    size_t n_sites, n_rows;
    uint32_t max_cov;
    uint2 *rec_all, *rec_cov;
    uint32_t *cov_site;
    uint32_t *block_tmp;        // prep_scan_blocks(n_sites) words
    PrepInfo *info;             // device
    PrepInfo *mirror;           // host-mapped copy a one-wave kernel fills in at the end of the stage, then mirror->seq = seq
    uint32_t seq;
};

struct PrepSegArgs {
    const uint2 *rec_cov;
    uint32_t n_cov, window, n_win, max_cov;
    const WinRaw *nck;          // [(max_cov+1)^2] binomial coefficients, normalised: C = m / 2^64 x 2^e, m in [2^63, 2^64)
    Seg *segs;                  // room for seg_cap segments
    uint32_t *seg_first;        // [seg_cap] first covered row of every segment
    uint32_t seg_cap;           // segments beyond it are counted but not written (rows out of file order only)
    WinConst *wconst;           // [n_win + 1]
    WinRaw *raw;                // [n_win] K = prod C(cov, n_ref) per window
    uint32_t *block_tmp;        // prep_scan_blocks(n_cov) words
    PrepInfo *info;
    PrepInfo *mirror;
    uint32_t compact = 0;       // 0: segments of the panel's own tiles; otherwise of the compacted tiles of the site list
                                // (k_gather_transpose32) with this many virtual rows per window (>= window)
};

size_t prep_scan_blocks(size_t n);
// stage A: rec_all, rec_cov, cov_site, info->{n_cov, err_*, first_row, last_row}; the mirror gets them with seq
void launch_prep_sites(const PrepSiteArgs &a, hipStream_t st);
// stage B: segment masks (on st), per-window reads / alt reads / K (on st2, which must be idle and see stage A's
// results: the caller has waited for stage A), info->{n_segs, ct_max, out_of_order}.  The caller makes st wait for st2
// before it queues anything that reads the per-window constants.
void launch_prep_segments(const PrepSegArgs &a, const uint32_t *run_begin, uint32_t n_runs, uint32_t ring, hipStream_t st,
                          hipStream_t st2);
// control words for ANOTHER run structure than the one given to launch_prep_segments, info->{max_seg, adv_overflow}; the mirror
// gets all of stage B with seq
// redo = false: the segment kernels have made the control words for this very run structure already, only the hand-over is queued
void launch_prep_seg_flags(const PrepSegArgs &a, const uint32_t *run_begin, uint32_t n_runs, uint32_t ring, uint32_t seq,
                           hipStream_t st, bool redo);
// wconst[w].{mK, eK} = K * pow_1me[reads of w]
void launch_prep_win_kp(uint32_t n_win, const WinRaw *raw, const WinRaw *pow_1me, WinConst *wconst, hipStream_t st);
// weight[t][n] = n == targets[t] ? 0 : base_w[n];  n_refpanel[t] = base_sum - base_w[targets[t]]  (src/ibdgem.c:714, :742-750);
// n_refpanel[n_targets + t] = targets[t] (n_refpanel holds 2 n_targets entries)
// (inline_targets != NULL and n_targets <= IBDG_TG_INLINE: the indices travel as kernel arguments and the kernel writes
// `targets` too; otherwise `targets` must hold them already)
#define IBDG_TG_INLINE 16
struct TargetsInline {
    uint32_t v[IBDG_TG_INLINE];
};
void launch_target_weights(const double *base_w, uint32_t *targets, const uint32_t *inline_targets, uint32_t n_targets,
                           uint32_t lanes, int base_sum, double *weight, int *n_refpanel, hipStream_t st);
void launch_prep_win_bounds(const uint32_t *cov_site, uint32_t n_cov, uint32_t window, uint32_t n_win, uint32_t *first,
                            uint32_t *last, hipStream_t st);

// max_blocks: 0 = as many single-wave workgroups as keep every CU's wave slots full (the kernel alone on the
// chip); a small number (e.g. 8 per CU) when it runs beside a kernel that should keep most of the slots
void launch_alt_count(const uint64_t *panel, uint32_t stride, size_t n_rows, uint32_t *alt_count,
                      hipStream_t st, unsigned max_blocks = 0);
// max_blocks: 0 = a wave per pair of consecutive windows (four waves per workgroup); otherwise at most that many workgroups per target, which
// walk the windows grid-stride (few long-lived waves: for running beside the --LD kernel)
void launch_rows_windows(const RowsArgs &a, unsigned n_targets, hipStream_t st, unsigned max_blocks = 0);
// af[s] of every site (k_site_af)
void launch_site_af(const RowsArgs &a, hipStream_t st);
// the row table of a run over several comparison individuals, and one individual's per-site table from it
void launch_row_table(const RowsArgs &a, hipStream_t st);
void launch_site_expand(const RowsArgs &a, uint32_t tgt, double *out, hipStream_t st);
int launch_ld(const LdArgs &a, unsigned n_targets, int cpw, unsigned waves, hipStream_t st);

}  // namespace ibdg
