/*
 * include/ibdgem_hip.h -- C ABI of the MI355X IBD-likelihood engine.
 *
 * Drop-in boundary for ONE path of Paleogenomics/IBDGem: the per-SNP
 * P(D|IBD0,1,2) arithmetic of src/ibd-math.c and the window / --LD
 * background-panel loop that the reference writes inline in
 * compare_impute()/compare_vcf() (src/ibdgem.c:558-760 and :247-462).
 *
 * The reference has no FFI for this path: the host program
 * calls find_pDgG/find_pDgf/find_pDgIBD1 (src/ibd-math.h:14-63) once per row
 * and runs the LD loop in place.  This header is the seam a maintainer binds
 * instead (INTEGRATION.md shows the patch to compare_impute): the host keeps
 * parsing and row filtering (integer work, src/ibdgem.c:573-630), hands the
 * engine the rows that survive, and prints what comes back.
 *
 * Conventions (mirroring the reference's, src/ibdgem.c:947-1041):
 *   - every call returns 0 on success, non-zero on error; the message is
 *     available from ibdg_last_error(); the library never exits the process
 *     and never falls back to a CPU implementation;
 *   - the caller owns all host buffers; the context owns all device memory;
 *   - one context per device, used from one host thread at a time.
 *
 * Plain C: pointers and sizes only.
 */
#ifndef IBDGEM_HIP_H
#define IBDGEM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IBDG_ABI_VERSION 5   /* 5: ibdg_num_targets, ibdg_upload_panel_fd; a run over new comparison individuals queues without a host wait; options ibd0_after, mfma_batch_groups, mfma_wg_sum; ibdg_last_count_unit 3.  4: ibdg_ld_layout, ibdg_last_count_unit, ibdg_get_window_ll_all; options compact_tiles, compact_density, compact_targets; the strict kernel is no
                              * longer what a sparse pileup gets.  3: options site_results, stage_workers; ibdg_get_site_af
                              * computes on demand; ibdg_last_run_ms out[4] is 0 */

typedef struct ibdg_ctx ibdg_ctx;

int ibdg_abi_version(void);

/* Number of visible HIP devices (0 if none / no driver). */
int ibdg_device_count(void);

/* Create an engine on `device`.  Replaces init_nCk() + the per-row
 * find_pDgG() calls (src/ibdgem.c:1168, :632-634; src/ibd-math.c:13-81):
 * the (max_cov+1)^2 x 3 table of P(D|G) is built here on the host with libm
 * pow(), exactly as the reference evaluates it, and kept on the device.
 * epsilon = -e, max_cov = -M.  Returns NULL on failure (see
 * ibdg_last_error(NULL)). */
ibdg_ctx *ibdg_create(int device, double epsilon, unsigned max_cov);

/* Replaces destroy_nCk() (src/ibd-math.c:34-43) and frees all device memory. */
void ibdg_destroy(ibdg_ctx *ctx);

/* Last error message of `ctx` (or of the last failed ibdg_create when ctx is
 * NULL).  Never NULL; empty string when there was no error. */
const char *ibdg_last_error(const ibdg_ctx *ctx);

/* Host-side helper: the P(D|G) table the context uses, as
 * out[(n_ref*(max_cov+1)+n_alt)*3 + g], g = 0:(0,0) 1:het 2:(1,1)
 * (src/ibd-math.c:46-81).  out holds (max_cov+1)^2*3 doubles. */
int ibdg_pdg_table(double epsilon, unsigned max_cov, double *out);

/* Host-side twins of find_pDgf and find_pDgIBD1 (src/ibd-math.h:40-63, src/ibd-math.c:84-142) for one
 * row: P(D|IBD0) and P(D|IBD1) from the allele frequency f, the three P(D|G) of the row (a row of
 * ibdg_pdg_table) and, for IBD1, the comparison individual's two alleles (0/1; anything else leaves the
 * value at 1.0 like the reference).  Same operations in the same order as the device kernel (libm pow,
 * no fused multiply-add, zero -> DBL_MIN), so they return the bits ibdg_get_site_ll returns.  For spot
 * checks and bindings that want single values; the engine itself never computes on the host. */
double ibdg_pdg_ibd0(double f, double p00, double p01, double p11);
double ibdg_pdg_ibd1(unsigned a0, unsigned a1, double f, double p00, double p01, double p11);

/* ---- phased panel (the .hap rows / VCF GT columns) ---------------------- */

/* A packed row is ibdg_row_words(n_ids) 64-bit words: for individual n,
 * chunk = n/64, bit = n%64; word[2*chunk] holds the first haplotype
 * (reference hap_buf[4n], src/ibdgem.c:638), word[2*chunk+1] the second
 * (hap_buf[4n+2], :639).  Unused high bits are zero. */
size_t ibdg_row_words(unsigned n_ids);

/* Pack one row of 2*n_ids alleles (0/1 bytes, [2n]=first, [2n+1]=second). */
void ibdg_pack_alleles(const uint8_t *alleles, unsigned n_ids, uint64_t *row);

/* Pack one IMPUTE .hap text row ("0 1 1 0 ...", the reference's hap_buf):
 * characters at even offsets are alleles.  Returns 0, or 1 if the row is
 * shorter than 4*n_ids-1 characters or holds a character other than '0'/'1'
 * at an allele offset. */
int ibdg_pack_hap_text(const char *hap_line, unsigned n_ids, uint64_t *row);

/* Upload n_rows packed rows (host memory, row-major, ibdg_row_words each).
 * Also computes the per-row alternate-allele counts on the device
 * (find_f_impute/find_f_vcf, src/ibd-parse.c:91-110) unless deferred to
 * ibdg_run (see ibdg_set_option "count_in_run"). */
int ibdg_upload_panel(ibdg_ctx *ctx, const uint64_t *rows, size_t n_rows, unsigned n_ids);

/* Same, with the packed rows in a FILE: n_rows x ibdg_row_words(n_ids) x 8 bytes from byte `offset` of the open file `fd`
 * (the host program's packed-panel cache).  The engine's staging threads read the file (pread) straight into their
 * page-locked buffers: the caller maps nothing, so there are no page faults on 2.56 GB of mapping during the upload and
 * no page-table entries to take down when the process ends (80 ms of a 0.45 s run of the host program).  The file
 * offset of `fd` is neither used nor changed; an error if the file ends before the rows do. */
int ibdg_upload_panel_fd(ibdg_ctx *ctx, int fd, uint64_t offset, size_t n_rows, unsigned n_ids);

/* Same, from memory that is already on this context's device
 * (e.g. a torch tensor's data_ptr()); copied device-to-device. */
int ibdg_upload_panel_dev(ibdg_ctx *ctx, const void *dev_rows, size_t n_rows, unsigned n_ids);

/* ---- the rows of one comparison ------------------------------------------ */

/* The rows that passed the reference's filter chain (src/ibdgem.c:584-626)
 * in file order: row_index into the uploaded panel (NULL: site s is panel row
 * s, n_sites <= panel rows), read counts after -D culling
 * (src/ibdgem.c:620-628; n_ref+n_alt <= max_cov), optional -A
 * frequency (NaN = use the panel's own; pointer may be NULL), window = -w.
 * Rows with n_ref+n_alt == 0 get per-site values but join no window
 * (src/ibdgem.c:657-663).  Windows are consecutive runs of `window` covered
 * rows; the last may be shorter (src/ibdgem.c:572-578, :736).
 * The three arrays are copied to the device (6 bytes per row) and everything
 * else -- the covered-row list, the windows, the per-window constants of the
 * --LD kernel -- is derived there (ibdgem_amd/csrc/ibdg_prep.hip); arrays from
 * ibdg_host_alloc are copied at link speed.  Returns when the device is done. */
int ibdg_upload_sites(ibdg_ctx *ctx, const uint32_t *row_index, const uint8_t *n_ref,
                      const uint8_t *n_alt, const double *f_override, size_t n_sites,
                      unsigned window);

/* Same, with row_index / n_ref / n_alt already in this context's device memory
 * (uint32 / uint8 / uint8; dev_row_index may be NULL as above).  f_override
 * stays a HOST array (or NULL): its powers are taken with the host's libm.
 * The call first waits for the whole device (the arrays may come from any stream); with option
 * "dev_inputs_ready" 1 the caller vouches that they are complete and the wait is left out -- the
 * preparation of one context can then run under the --LD kernel of another on the same GPU. */
int ibdg_upload_sites_dev(ibdg_ctx *ctx, const void *dev_row_index, const void *dev_n_ref,
                          const void *dev_n_alt, const double *f_override, size_t n_sites,
                          unsigned window);

/* Clocks of the last ibdg_upload_sites[_dev] (ms): out[0] host-to-device copies
 * of the arrays, out[1] preparation on the device including its two small
 * read-backs, out[2] the whole call on the host's clock. */
int ibdg_upload_ms(ibdg_ctx *ctx, float out[3]);

/* Page-locked host memory for arrays handed to ibdg_upload_* / ibdg_get_*:
 * copies to and from it run at the link's speed instead of going through a
 * staging buffer.  NULL on failure.  Any host memory works; this is optional. */
void *ibdg_host_alloc(size_t bytes);
void ibdg_host_free(void *p);

size_t ibdg_num_sites(const ibdg_ctx *ctx);
size_t ibdg_num_windows(const ibdg_ctx *ctx);
/* Comparison individuals of the last ibdg_run: what ibdg_get_window_ll_all copies is
 * ibdg_num_targets x ibdg_num_windows x 3 doubles. */
size_t ibdg_num_targets(const ibdg_ctx *ctx);

/* Per window: index (into the uploaded site list) of its first and last
 * covered row, and NUM_SITES (src/ibdgem.c:723-730, :751-756).  Any pointer
 * may be NULL. */
int ibdg_get_windows(ibdg_ctx *ctx, uint32_t *first, uint32_t *last, uint32_t *n_covered);

/* ---- run ------------------------------------------------------------------- */

/* Evaluate n_targets comparisons (the reference's loop over data->uids,
 * src/ibdgem.c:522) in one call.
 *   targets[t]  individual index of the compared sample (cmp_idx/2)
 *   bg_count    NULL = every individual is background once
 *               (src/ibdgem.c:517-520), else n_ids bytes: how many times the
 *               individual appears in the -B list (read_rf keeps duplicates,
 *               src/ibd-parse.c:262-308); one byte each, so at most 255
 *               listings of one individual (the host program refuses more)
 *   pu_id       individual whose name equals -N, or -1 (src/ibdgem.c:501-506)
 *   ld_mode     --LD: LIBD0/LIBD1 of each window are the background averages
 *               of src/ibdgem.c:736-753; otherwise the plain products (:755).
 * Results stay on the device until fetched. */
int ibdg_run(ibdg_ctx *ctx, const uint32_t *targets, size_t n_targets, const uint8_t *bg_count,
             int pu_id, int ld_mode);

/* AF column: alt-allele fraction per uploaded site (or the -A override): alt count / (2 n_ids), src/ibd-parse.c:98.
 * Computed by this call (it depends on the panel row only), after an ibdg_run. */
int ibdg_get_site_af(ibdg_ctx *ctx, double *af);
/* LIBD0, LIBD1, LIBD2 per site of target t: out[n_sites][3] (tab columns 12-14). */
int ibdg_get_site_ll(ibdg_ctx *ctx, size_t t, double *out);
/* LIBD0, LIBD1, LIBD2 per window of target t: out[n_windows][3] (summary columns 4-6). */
int ibdg_get_window_ll(ibdg_ctx *ctx, size_t t, double *out);
/* The same for every target of the last ibdg_run in one copy: out[ibdg_num_targets][n_windows][3] (a caller that runs batches of
 * comparison individuals takes a batch's tables off the device at once and can queue the next batch before it goes
 * through them: the host program's --summary-only loop, reference src/ibdgem.c:522 with :751-756). */
int ibdg_get_window_ll_all(ibdg_ctx *ctx, double *out);
/* Alt-allele count of panel rows [first_row, first_row+n): out[n] (for tests). */
int ibdg_get_alt_counts(ibdg_ctx *ctx, size_t first_row, size_t n, uint32_t *out);

/* ---- measurement / tuning --------------------------------------------------- */

/* Device time of the last ibdg_run, from HIP events on the engine's streams:
 * out[0] total (first launch to last completion), out[1] alt-count kernel
 * (0 if not run), out[2] the kernel of the per-row values and window products,
 * out[3] the --LD launches, out[4] 0 (ms; up to ABI 2 the window products were
 * a kernel of their own).  The per-row kernel runs on a second stream beside
 * the --LD kernels, so the parts overlap and need not add up to the total. */
int ibdg_last_run_ms(ibdg_ctx *ctx, float out[5]);

/* The same for the run `back` calls ago (0 = the last one); the engine keeps the
 * events of its last 32 runs.  Waits for the engine's stream first, so it is the
 * way to time runs issued with the "async" option after ibdg_sync. */
int ibdg_run_ms(ibdg_ctx *ctx, unsigned back, float out[5]);

/* Duration (ms) of the dominant --LD kernel alone in that run (k_ld_popcount, or its
 * multi-individual form), from the start / stop times of its own dispatch packet.  An error if the
 * run used the strict kernel or was not an --LD run. */
int ibdg_run_kernel_ms(ibdg_ctx *ctx, unsigned back, float *ms);

/* Which --LD kernel the last ibdg_run used: 0 none (non-LD), 1 the strict
 * kernel (sequential fp64 products in the reference's order), 2 the
 * exponent-counting kernel (same values to ~1e-14), 3 the
 * reference-order mode (strict products, then the background sums taken
 * serially in the reference's order: LIBD0/LIBD1 of every window bit-identical
 * to the reference; ~12x the time of 2). */
int ibdg_last_ld_variant(const ibdg_ctx *ctx);

/* Which tiles the exponent-counting / matrix-core --LD kernels read for the site list at hand: 0 none prepared
 * (no sites, or only the strict kernel applies), 1 the panel's own 32-row tiles (every tile between a window's
 * first and last panel row is streamed), 2 the compacted tiles of this site list: only the rows that carry reads (the
 * rows the reference's window loop multiplies, src/ibdgem.c:596-601, :657-663), back to back (option "compact_align" 1, the
 * default since ABI 5; 32 = every window on a tile boundary), gathered and transposed once per ibdg_upload_sites -- or by
 * the ibdg_run with which the runs on this upload have added up to "compact_targets" comparison individuals -- so that a
 * window costs window / 32 tile words whatever the pileup's density.  Chosen per upload and per run (option
 * "compact_tiles"); the results are the same bits from either. */
int ibdg_ld_layout(const ibdg_ctx *ctx);

/* How the last ibdg_run's exponent-counting launches for single comparison individuals (k_ld_popcount: one individual,
 * or the one to four left over beside the groups of the other kernels) took the weighted sums of a haplotype word:
 * 2 = one matrix instruction per word (v_mfma_scale_f32_16x16x128_f8f6f4: the rows' weights as FP6, the word's bits as
 * FP4; option "mx_counts", the default), 1 = twelve (mask, count) pairs on the vector ALU, 0 = no such launch in that
 * run; 3 = form 2 counting the IBD1 sums only (the instruction returns the table exponents of the four IBD1 products
 * themselves), IBD0 taken from ONE pass over the site list that keeps every background individual's own product per
 * window -- src/ibdgem.c:715, :743: it does not depend on the comparison individual, whose only trace in that sum is its
 * own exclusion, :714 -- (option "ibd0_after").  The sums are exact integers and the floating-point additions the same
 * in the same order in every form: same results, bit for bit. */
int ibdg_last_count_unit(const ibdg_ctx *ctx);

/* Options: "dispatch_events" (0/1: time the --LD launches through their own
 * dispatch packets, which makes ibdg_run_kernel_ms available; costs ~10 us per
 * run more than the default single event record); "async" (0/1: ibdg_run returns as soon as its kernels are queued;
 * ibdg_sync, ibdg_run_ms and every ibdg_get_* wait for them -- lets a caller
 * queue one run per comparison individual without a host round trip between
 * them); "dev_inputs_ready" (0/1: ibdg_upload_sites_dev does not wait for the whole device first, see there);
 * "count_in_run" (0/1: recompute alt counts inside every ibdg_run,
 * so the timed region covers it; beside the --LD kernel the recount runs with "recount_blocks_per_cu"
 * single-wave workgroups per CU, default 4, 0 = its full grid); "site_blocks_per_cu" (default 4: workgroups per CU of the
 * per-row kernel beside the exponent-counting --LD kernel, 0 = its full grid) and "rows_blocks_per_cu" (default 0 = full
 * grid: the same for a non-LD run, where it has the chip to itself); "site_results" (what ibdg_run keeps per row: 1, the default,
 * LIBD0/1/2 of every row and comparison individual for ibdg_get_site_ll; 0 nothing -- no n_targets x n_sites x 24 bytes
 * of device memory, no per-row stores, and in --LD mode only the IBD2 pick of a row is computed at all: for callers that
 * want the window table only, e.g. hundreds of comparison individuals in one call; ibdg_get_site_ll then fails.  The AF
 * column never costs a run anything: ibdg_get_site_af computes it when called); "stage_workers" (1..8, default 8: host threads of that staging team -- a caller
 * whose contexts upload at the same time gives each its share); "staged_upload" (0/1, default 1: a panel of 256 MB or more in
 * ordinary host memory goes to the device through page-locked staging buffers filled by a team of host
 * threads instead of the runtime's pageable-memory path); "ld_variant" (0 = pick automatically,
 * 1 = strict, 2 = exponent counting, an error if not applicable, 3 = reference
 * order);
 * "compact_tiles" (set before ibdg_upload_sites: 0, the default = the compacted tiles of ibdg_ld_layout when fewer than
 * one panel row in "compact_density" (default 4) between the first and the last site carries reads, when the rows
 * are not in file order, or once the runs on one upload have added up to "compact_targets" (default 256) comparison
 * individuals -- the site list belongs to the pileup, src/ibdgem.c:522 runs every individual over the same rows; a
 * group of the matrix-core kernel counts as 20, an individual of the counting kernels as 12 (16 with "mx_counts" 0): what
 * each saves on the compacted tiles -- a single run 0.058 of 0.606 ms at chr1 x 2504 -- against the 1.3 ms of the gather
 * and the new segments, i.e. the 22nd single run on an upload re-lays it out -- the panel's own tiles otherwise; 1 =
 * always; -1 = never: sparse or unordered site lists then take the strict kernel);
 * "reserve_compact" (0/1, default 1, set before ibdg_upload_panel: the buffer of the compacted tiles, 1.3 x the
 * panel's, is allocated with the panel so that a re-layout never allocates);
 * "ibd0_after" (default 8; 0 = never: the single comparison individuals run on one upload and background add up, and
 * the run that reaches this number makes the pass of ibdg_last_count_unit's form 3 -- about the cost of one run, a fifth
 * of every later one saved; at once where a run of groups of 15 has made the pass already.  708 MB of device memory at
 * chr1 x 2504);
 * "mx_counts" (0/1, default 1: see ibdg_last_count_unit; applies where a run's records of 128 bytes per (window, tile)
 * segment fit the workgroup's LDS and the powers rho^n 2^(s n) of a window's table stay normal doubles, always so at the
 * table sizes kept in LDS); "sum_dpp" (0/1, default 1: the wave sums of that form exchange by DPP moves instead of
 * ds_swizzle -- same additions, same bits);
 * "finalize_in_next" (0/1, default 1; with "async" only: a run of single comparison individuals leaves its finalising
 * step -- the sum over the chunks and the background average, src/ibdgem.c:751-752 -- to the next run's --LD launch when
 * that run has the same shape, background and prepared sites (ABI 5: the individuals may differ); whoever reads results or
 * replaces inputs first gets a launch of its own for it: one launch, its gap and an event packet less per queued run, same
 * results);
 * "compact_align" (1, 2, 4, 8, 16 or 32, default 1, set before ibdg_upload_sites: the rows a window of the compacted tiles
 * is rounded up to -- 1 = the site list's rows with reads back to back, no padding; 32 = every window on a tile boundary,
 * round 4's layout);
 * "prep_ahead" (0/1, default 1; with "async" only: what a NEW comparison individual needs before its --LD kernel -- its
 * background weights and window / segment images -- is made on a third stream into a ring of four buffers while the runs
 * before still read theirs; 0: on the main stream, in front of the kernel);
 * "end_in_dispatch" (0/1, default 1: the end event of a run of single individuals is its --LD kernel's own completion
 * signal instead of an event packet behind the kernel -- 7 us less between two queued runs);
 * "chunks_per_wave" (strict kernel tiling, set before ibdg_upload_panel),
 * "waves_per_block" (strict kernel), "windows_per_wave", "guided_runs",
  * "ring_slots" (2, 3, 4 or 8), "record_lds_bytes" (exponent-counting kernel;
 * set before ibdg_upload_sites); "multi_target" (0/1, default 1: with four or more
 * comparison individuals in one ibdg_run, groups of four share a workgroup of
 * the exponent-counting kernel -- same results, ~1.4x the throughput); "mfma_targets" (0/1,
 * default 1: with "mfma_min" (1..15, default 4) or more comparison individuals in one ibdg_run,
 * groups of 15 go through the matrix-core kernel -- the sums that depend on the comparison
 * individual as integer matrix products; same results, ~2.5x the throughput of single runs; "mfma_plain_tau"
 * (0/1, default 1: that kernel looks the powers tau^G of a window up as plain doubles where none of them leaves the
 * double range -- the same bits as the {mantissa, exponent} table it uses otherwise, half the LDS traffic);
 * "mfma_batch_groups" (1..64, default 36: groups of 15 per launch of that kernel -- the groups' workgroups of one run sit
 * next to each other on one XCD, whose L2 then serves the panel's tile words to all but the first of them; bounded by a
 * sixteenth of the device's memory for the groups' operands and partial sums); "mfma_wg_sum" (0/1, default 1: a workgroup of
 * that kernel adds its eight waves' window sums up itself where its LDS allows: an eighth of the partial sums written)).
 * Returns non-zero for an unknown name or a value out of range. */
int ibdg_set_option(ibdg_ctx *ctx, const char *name, long value);

/* Reference-order mode only ("ld_variant" 3): the background individuals in the
 * order the reference walks them -- the lines of the -B file, duplicates kept
 * (data->refids, src/ibd-parse.c:262-308; src/ibdgem.c:741).  Must agree with
 * the bg_count given to ibdg_run.  n = 0 clears it: ascending individuals, each
 * bg_count[n] times, which is the reference's order when there is no -B. */
int ibdg_set_background_order(ibdg_ctx *ctx, const uint32_t *ids, size_t n);

/* Host-side self checks that need no device (for the CPU test tier): "wait_info" = the bounded poll the uploads
 * wait for the site preparation with (the word arrives / the stream ends without it / the stream fails / neither:
 * wall-clock bound).  0 = passed, 1 = unknown check, 2 = failed. */
int ibdg_selftest(const char *what);

/* Block until all work queued on the engine's stream is done. */
int ibdg_sync(ibdg_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* IBDGEM_HIP_H */
