/*
 * ibdgem.c -- host program of the MI355X IBD-likelihood engine: the reference's
 * command line, input formats and output files, with the likelihood arithmetic done by
 * libibdgem_hip.so (include/ibdgem_hip.h).
 *
 * What stays on the CPU is what the north star leaves there: option parsing, text parsing
 * and the integer row filter chain of compare_impute (reference src/ibdgem.c:572-630).
 * What the reference computes per row and per window in that loop (:632-667, :669-722,
 * :736-756) is one ibdg_run per batch of comparison individuals.  On a machine without a HIP
 * device a non-LD run (BASELINE configs[0]: "plumbing, runs without a GPU") takes the per-row
 * values and the window products from the library's own host twins of the reference's math seam
 * (ibdg_pdg_table / ibdg_pdg_ibd0 / ibdg_pdg_ibd1, src/ibd-math.h:14-63) in the reference's
 * sequential order (host_nonld below); an --LD run without a device stops with the engine's
 * error -- the background-panel loop exists on the device only.
 *
 * Differences from the reference, all outside the BASELINE configs:
 *   - VCF input (-V) follows the IMPUTE semantics: any number of comparison individuals (the
 *     reference frees its genotype regex inside the per-individual loop, src/ibdgem.c:472, and
 *     crashes on the second one) and the -B background indexed by individual (the reference's VCF
 *     loop indexes genotypes by list position, src/ibdgem.c:374-375);
 *   - no 30720-byte line limit (src/file-io.h:10); .hap rows with a character other than
 *     '0'/'1' at an allele offset are counted as skipped instead of read as garbage;
 *   - the genotype files are read once, not once per comparison individual
 *     (src/ibdgem.c:771-772);
 *   - extra long option --plan: print, per comparison individual, the rows that pass the
 *     filter chain with their integer columns and the window boundaries, and exit.  It is the
 *     hook the CPU-only tests use to check the host logic against the reference's outputs.
 */
#define _GNU_SOURCE
#include <ctype.h>
#include <float.h>
#include <getopt.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#ifdef __GLIBC__
#include <malloc.h>
#include <sys/mman.h>
#endif
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <fcntl.h>
#include <errno.h>

#include "../../include/ibdgem_hip.h"
#include "ingest.h"
#include "lineio.h"
#include "pileup.h"

/* ---- options (names and defaults: reference src/ibdgem.c:21-66) ------------------- */
static double opt_eps = 0.02, opt_min_qual = 0, opt_max_af = 1, opt_min_af = 0, opt_target_dp = 0;
static unsigned opt_max_cov = 20;
static int opt_window = 100;
static const char *opt_sq = "UNKWN";
static int opt_ld = 0, opt_plan = 0, opt_summary_only = 0, opt_ref_order = 0, in_impute = 0, in_vcf = 0;
static int has_S = 0, has_s = 0, has_B = 0, has_A = 0, has_p = 0, has_v = 0, has_D = 0;
static int opt_threads = 0;
static const char *cache_fn = NULL, *dump_panel_fn = NULL;

static struct option longopts[] = {
    {"LD", no_argument, &opt_ld, 1},
    {"plan", no_argument, &opt_plan, 1},
    {"summary-only", no_argument, &opt_summary_only, 1},
    {"reference-order", no_argument, &opt_ref_order, 1},
    {"rand-stream", required_argument, 0, 1000},
    {"fmt-check", required_argument, 0, 1005},
    {"devices", required_argument, 0, 1001},
    {"threads", required_argument, 0, 1002},
    {"panel-cache", required_argument, 0, 1003},
    {"dump-panel", required_argument, 0, 1004},
    {"vcf", required_argument, 0, 'V'},
    {"hap", required_argument, 0, 'H'},
    {"legend", required_argument, 0, 'L'},
    {"indv", required_argument, 0, 'I'},
    {"pileup", required_argument, 0, 'P'},
    {"pileup-name", required_argument, 0, 'N'},
    {"window-size", required_argument, 0, 'w'},
    {"allele-freqs", required_argument, 0, 'A'},
    {"sample-list", required_argument, 0, 'S'},
    {"sample", required_argument, 0, 's'},
    {"background-list", required_argument, 0, 'B'},
    {"max-cov", required_argument, 0, 'M'},
    {"downsample-cov", required_argument, 0, 'D'},
    {"out-dir", required_argument, 0, 'O'},
    {"max-af", required_argument, 0, 'F'},
    {"min-af", required_argument, 0, 'f'},
    {"positions", required_argument, 0, 'p'},
    {"min-qual", required_argument, 0, 'q'},
    {"chromosome", required_argument, 0, 'c'},
    {"error-rate", required_argument, 0, 'e'},
    {"variable-sites-only", no_argument, 0, 'v'},
    {"help", no_argument, 0, 'h'},
    {0, 0, 0, 0}};

static void usage(int code)
{
    fputs("ibdgem (MI355X engine): likelihood that low-coverage reads (pileup) and a genotyped\n"
          "individual share 0, 1 or 2 chromosomes identical by descent, per SNP and per window.\n\n"
          "Usage: ibdgem [--LD] -H hap -L legend -I indv -P pileup [options]\n"
          "  --LD                      background-panel (linkage-aware) window likelihoods\n"
          "  -H/--hap, -L/--legend, -I/--indv FILE   IMPUTE genotype input (plain or .gz)\n"
          "  -V/--vcf FILE             VCF genotype input (plain or .gz), biallelic SNP rows with 0/1 genotypes\n"
          "  -P/--pileup FILE          samtools pileup of the unknown sample (required)\n"
          "  -N/--pileup-name STR      name of the pileup sample (default UNKWN)\n"
          "  -A/--allele-freqs FILE    CHROM POS AF table overriding the panel's own frequencies\n"
          "  -S/--sample-list FILE     individuals to compare against, one per line\n"
          "  -s/--sample STR           the same, comma separated\n"
          "  -B/--background-list FILE individuals forming the --LD background panel\n"
          "  -p/--positions FILE       restrict to these sites (CHROM POS or BED)\n"
          "  -q/--min-qual FLOAT       minimum QUAL (VCF input only)\n"
          "  -M/--max-cov INT          skip sites with more informative reads (default 20)\n"
          "  -F/--max-af, -f/--min-af FLOAT   alternate-allele frequency range (default 0..1)\n"
          "  -D/--downsample-cov FLOAT thin the reads to this mean depth\n"
          "  -w/--window-size INT      covered sites per summary window (default 100)\n"
          "  -O/--out-dir DIR          output directory (default: current directory)\n"
          "  -c/--chromosome STR       use only this chromosome of the pileup / -A / -p files\n"
          "  -e/--error-rate FLOAT     sequencing error rate (default 0.02)\n"
          "  -v/--variable-sites-only  skip sites where the compared individual is 0/0\n"
          "  --devices LIST            GPUs to spread the windows of each comparison over, e.g. 0,1,2,3\n"
          "                            (default 0); every GPU holds the whole panel, the windows are cut\n"
          "                            into contiguous ranges, one per GPU, and gathered on the host\n"
          "  --threads INT             host threads for reading the .hap file and writing the output files\n"
          "                            (default: the CPUs available to the process, at most 16)\n"
          "  --panel-cache FILE        keep the bit-packed panel of the .hap file in FILE and reuse it\n"
          "                            while the .hap file is unchanged\n"
          "  --summary-only            write the *.summary.txt files only (no per-site *.tab.txt files: the\n"
          "                            per-site values are neither copied back from the device nor formatted)\n"
          "  --reference-order         --LD: sum over the background panel serially in the reference's order\n"
          "                            (LIBD0/LIBD1 bit-identical to the reference; ~12x slower than the default,\n"
          "                            whose values agree to ~1e-15)\n"
          "  --plan                    print the filtered rows and windows only (no device needed)\n"
          "  -h/--help\n\n"
          "Outputs <out>/<pileup-name>.<individual>.tab.txt with columns\n"
          "CHR rsID POS REF ALT AF DP SQ_NREF SQ_NALT GT_A0 GT_A1 LIBD0 LIBD1 LIBD2\n"
          "and <out>/<pileup-name>.<individual>.summary.txt with columns\n"
          "SEGMENT START END LIBD0 LIBD1 LIBD2 NUM_SITES\n",
          stderr);
    exit(code);
}

/* ---- panel individuals -------------------------------------------------------------- */
typedef struct {
    char **names;
    size_t n;
} names_t;

static void chomp1(char *s)
{
    size_t n = strlen(s);
    if (n)
        s[n - 1] = '\0';               /* the reference drops the last character (the newline) */
}

static int read_names(const char *fn, names_t *out)
{
    line_src *ls = ls_open(fn);
    if (!ls)
        return 1;
    char *line;
    size_t cap = 0;
    out->n = 0;
    out->names = NULL;
    while ((line = ls_next(ls, NULL))) {
        chomp1(line);
        if (out->n == cap) {
            cap = cap ? cap * 2 : 256;
            out->names = ls_xrealloc(out->names, cap * sizeof *out->names);
        }
        out->names[out->n++] = strdup(line);
    }
    ls_close(ls);
    return 0;
}

static long find_name(const names_t *ids, const char *name)
{
    for (size_t i = 0; i < ids->n; ++i)
        if (strcmp(ids->names[i], name) == 0)
            return (long)i;
    return -1;
}

/* a list of panel individuals (by index) in list order, unknown names reported and dropped
 * (read_sf / read_rf, reference src/ibd-parse.c:176-219, :262-308) */
typedef struct {
    uint32_t *idx;
    size_t n;
} idlist_t;

static int read_idlist_file(const char *fn, const names_t *ids, const char *what_missing, const char *err_fn,
                            idlist_t *out)
{
    names_t l;
    if (read_names(fn, &l))
        return 1;
    out->idx = malloc((l.n ? l.n : 1) * sizeof *out->idx);
    out->n = 0;
    for (size_t i = 0; i < l.n; ++i) {
        long k = find_name(ids, l.names[i]);
        if (k < 0) {
            fprintf(stderr, what_missing, l.names[i]);
            continue;
        }
        out->idx[out->n++] = (uint32_t)k;
    }
    for (size_t i = 0; i < l.n; ++i)
        free(l.names[i]);
    free(l.names);
    if (out->n == 0) {
        fprintf(stderr, "[::] ERROR in %s(): No matching samples found in %s.\n", err_fn, fn);
        return 1;
    }
    return 0;
}

static int read_idlist_csv(const char *csv, const names_t *ids, idlist_t *out)
{
    char *buf = strdup(csv);
    out->idx = malloc((strlen(csv) / 2 + 2) * sizeof *out->idx);
    out->n = 0;
    /* strtok_r: the device contexts start on a thread of their own while this runs, and the runtime's start-up splits
     * strings with strtok too -- the two shared its hidden state, and every now and then the list ended early in one of
     * the runtime's strings ("Sample 00000000 not found ...", fewer comparisons than asked for, exit code 0) */
    char *save = NULL;
    for (char *tok = strtok_r(buf, ",", &save); tok; tok = strtok_r(NULL, ",", &save)) {
        long k = find_name(ids, tok);
        if (k < 0) {
            fprintf(stderr, "Sample %s not found in reference panel.\n", tok);
            continue;
        }
        out->idx[out->n++] = (uint32_t)k;
    }
    free(buf);
    if (out->n == 0) {
        fprintf(stderr, "[::] ERROR in read_scmd(): No matching samples found.\n");
        return 1;
    }
    return 0;
}

/* ---- -A and -p tables (reference src/ibd-parse.c:311-421) --------------------------- */
typedef struct {
    unsigned long pos;
    double f;
} af_rec;

static af_rec *af_tab;
static size_t af_n;
static unsigned long *pos_tab;
static size_t pos_n;

static int read_af_file(const char *fn, const char *chr)
{
    line_src *ls = ls_open(fn);
    if (!ls)
        return 1;
    size_t cap = 0;
    char *line, c[129];
    while ((line = ls_next(ls, NULL))) {
        af_rec r;
        if (sscanf(line, "%128s %lu %lf", c, &r.pos, &r.f) != 3)
            continue;
        if (chr && strcmp(c, chr) != 0)
            continue;
        if (af_n == cap) {
            cap = cap ? cap * 2 : 4096;
            af_tab = ls_xrealloc(af_tab, cap * sizeof *af_tab);
        }
        af_tab[af_n++] = r;
    }
    ls_close(ls);
    if (af_n == 0) {
        fprintf(stderr, "[::] ERROR in read_af(): Cannot parse lines from %s.\n", fn);
        return 1;
    }
    return 0;
}

static const af_rec *find_af(unsigned long pos)      /* bsearch over the (sorted) table */
{
    size_t lo = 0, hi = af_n;
    while (lo < hi) {
        size_t mid = lo + (hi - lo) / 2;
        if (af_tab[mid].pos == pos)
            return &af_tab[mid];
        if (af_tab[mid].pos < pos)
            lo = mid + 1;
        else
            hi = mid;
    }
    return NULL;
}

static int read_pos_file(const char *fn, const char *chr)
{
    line_src *ls = ls_open(fn);
    if (!ls)
        return 1;
    size_t cap = 0;
    char *line, c[129];
    while ((line = ls_next(ls, NULL))) {
        unsigned long p;
        /* BED (chr start end -> end) or chr pos */
        if (!(sscanf(line, "%128s %*d %lu", c, &p) == 2 || sscanf(line, "%128s %lu", c, &p) == 2))
            continue;
        if (chr && strcmp(c, chr) != 0)
            continue;
        if (pos_n == cap) {
            cap = cap ? cap * 2 : 4096;
            pos_tab = ls_xrealloc(pos_tab, cap * sizeof *pos_tab);
        }
        pos_tab[pos_n++] = p;
    }
    ls_close(ls);
    if (pos_n == 0) {
        fprintf(stderr, "[::] ERROR in read_pos(): Cannot parse lines from %s.\n", fn);
        return 1;
    }
    return 0;
}

static int cmp_ul(const void *a, const void *b)
{
    unsigned long x = *(const unsigned long *)a, y = *(const unsigned long *)b;
    return x < y ? -1 : x > y;
}

static int pos_listed(unsigned long pos)             /* membership (the reference scans linearly) */
{
    return bsearch(&pos, pos_tab, pos_n, sizeof *pos_tab, cmp_ul) != NULL;
}

/* ---- genotype rows -------------------------------------------------------------------- */
typedef struct {
    uint32_t id_off, ref_off, alt_off;   /* into the string arena */
    unsigned long pos;
    uint8_t legend_ok;                   /* the legend row had its 4 fields (src/ibdgem.c:589) */
    uint8_t hap_ok;                      /* the .hap row packed cleanly / the VCF row parsed, is biallelic, GTs ok */
    uint8_t gt_failed;                   /* VCF: a genotype field did not look like [01][/|][01] (message per run) */
    double qual;                         /* VCF QUAL column (-q) */
} row_t;

static row_t *rows;
static size_t n_rows;
static char *arena;
static size_t arena_len, arena_cap;
static uint64_t *packed;                /* the packed rows in host memory -- or, with a panel cache, NULL until something on the host
                                         * asks for a row (packed_rows): the engine takes the rows from the cache FILE */
static int packed_fd = -1;              /* the open panel cache and where its rows start */
static uint64_t packed_off;
static size_t packed_mapped_bytes;      /* > 0: `packed` is a mapping of the panel cache file of that many bytes */
static size_t packed_file_bytes;
static pthread_once_t packed_once = PTHREAD_ONCE_INIT;

static void packed_map(void)
{
    packed = ingest_cache_map(packed_fd, packed_off, packed_file_bytes);
    if (!packed) {
        fprintf(stderr, "[::] ERROR: cannot map the panel cache.\n");
        _exit(1);
    }
    packed_mapped_bytes = packed_file_bytes;
}

/* the packed rows for the few things the host itself reads them for (a row's alleles of the comparison individual in the
 * per-site table, -v, a run without a device): with a panel cache the file is mapped at the first such call */
static inline const uint64_t *packed_rows(void)
{
    if (__builtin_expect(!packed && packed_fd >= 0, 0))
        pthread_once(&packed_once, packed_map);
    return packed;
}
static size_t row_words;
static uint32_t *alt_count_h;            /* alt alleles per panel row, counted on the host (ingest.c) or read from the cache */

static uint32_t arena_add(const char *s)
{
    size_t l = strlen(s) + 1;
    if (arena_len + l > arena_cap) {
        arena_cap = arena_cap ? arena_cap * 2 : (1 << 20);
        while (arena_len + l > arena_cap)
            arena_cap *= 2;
        arena = ls_xrealloc(arena, arena_cap);
    }
    memcpy(arena + arena_len, s, l);
    arena_len += l;
    return (uint32_t)(arena_len - l);
}

/* The legend rows (small text) are parsed on a thread of their own while the team of ingest.c
 * packs the .hap rows; both files advance together in the reference (src/ibdgem.c:573-578), so the
 * panel has as many rows as the shorter of the two. */
typedef struct {
    const char *fn;
    size_t n;
    int rc;
    int threads;
} legend_job;

/* one legend line into a row and a string arena (src/ibdgem.c:589) */
typedef struct {
    row_t *rows;
    size_t n, cap;
    char *arena;
    size_t arena_len, arena_cap;
} legend_part;

static uint32_t part_arena_add(legend_part *t, const char *s)
{
    const size_t l = strlen(s) + 1;
    if (t->arena_len + l > t->arena_cap) {
        t->arena_cap = t->arena_cap ? t->arena_cap * 2 : (1 << 20);
        while (t->arena_len + l > t->arena_cap)
            t->arena_cap *= 2;
        t->arena = ls_xrealloc(t->arena, t->arena_cap);
    }
    memcpy(t->arena + t->arena_len, s, l);
    t->arena_len += l;
    return (uint32_t)(t->arena_len - l);
}

static void part_add_legend_line(legend_part *t, const char *l)
{
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : (1 << 16);
        t->rows = ls_xrealloc(t->rows, t->cap * sizeof *t->rows);
    }
    row_t *r = &t->rows[t->n++];
    memset(r, 0, sizeof *r);
    char id[129], ref[129], alt[129];
    if (sscanf(l, "%128s %lu %128s %128s", id, &r->pos, ref, alt) == 4) {
        r->legend_ok = 1;
        r->id_off = part_arena_add(t, id);
        r->ref_off = part_arena_add(t, ref);
        r->alt_off = part_arena_add(t, alt);
    }
}

/* A large plain legend is cut into byte ranges at line starts, one thread each.  Pass 1 counts the
 * lines of every range, so pass 2 can write its rows straight into their final places; the strings go
 * to an arena per thread, and once all sizes are known every thread copies its arena into the common
 * one and shifts the offsets of its own rows.  Small or gzip files go line by line. */
typedef struct {
    const char *base;
    size_t a, b;
    size_t lines;               /* pass 1 */
    size_t first;               /* pass 2: index of the range's first row */
    legend_part part;           /* .rows unused here: rows are written in place */
    size_t arena_base;          /* pass 3 */
} legend_range;

static void *legend_count(void *arg)
{
    legend_range *j = arg;
    size_t n = 0;
    for (const char *p = j->base + j->a, *e = j->base + j->b; p < e;) {
        const char *nl = memchr(p, '\n', (size_t)(e - p));
        ++n;
        if (!nl)
            break;
        p = nl + 1;
    }
    j->lines = n;
    return NULL;
}

static void *legend_parse(void *arg)
{
    legend_range *j = arg;
    char *buf = NULL;
    size_t cap = 0, n = j->first;
    for (size_t p = j->a; p < j->b;) {
        const char *nl = memchr(j->base + p, '\n', j->b - p);
        const size_t len = nl ? (size_t)(nl - (j->base + p)) + 1 : j->b - p;
        if (len + 1 > cap) {
            cap = (len + 1) * 2;
            buf = ls_xrealloc(buf, cap);
        }
        memcpy(buf, j->base + p, len);
        buf[len] = 0;
        row_t *r = &rows[n++];
        memset(r, 0, sizeof *r);
        char id[129], ref[129], alt[129];
        if (sscanf(buf, "%128s %lu %128s %128s", id, &r->pos, ref, alt) == 4) {
            r->legend_ok = 1;
            r->id_off = part_arena_add(&j->part, id);
            r->ref_off = part_arena_add(&j->part, ref);
            r->alt_off = part_arena_add(&j->part, alt);
        }
        p += len;
    }
    free(buf);
    return NULL;
}

static void *legend_place(void *arg)
{
    legend_range *j = arg;
    if (j->part.arena_len)
        memcpy(arena + j->arena_base, j->part.arena, j->part.arena_len);
    free(j->part.arena);
    const uint32_t shift = (uint32_t)j->arena_base;
    for (size_t i = j->first; i < j->first + j->lines; ++i)
        if (rows[i].legend_ok) {
            rows[i].id_off += shift;
            rows[i].ref_off += shift;
            rows[i].alt_off += shift;
        }
    return NULL;
}

static void team_run(void *(*fn)(void *), legend_range *rg, int T)
{
    pthread_t th[64];
    for (int t = 0; t < T; ++t)
        if (pthread_create(&th[t], NULL, fn, &rg[t]) != 0) {
            fn(&rg[t]);
            th[t] = pthread_self();
        }
    for (int t = 0; t < T; ++t)
        if (!pthread_equal(th[t], pthread_self()))
            pthread_join(th[t], NULL);
}

static void *read_legend(void *arg)
{
    legend_job *j = arg;
    size_t size = 0;
    const char *base = j->threads > 1 ? ls_map(j->fn, &size) : NULL;
    if (base && size >= ls_mt_min_bytes()) {
        const char *nl = memchr(base, '\n', size);              /* legend header (src/ibdgem.c:555) */
        const size_t from = nl ? (size_t)(nl - base) + 1 : size;
        const int T = j->threads > 64 ? 64 : j->threads;
        size_t cut[65];
        legend_range rg[64];
        ls_split_lines(base, size, from, T, cut);
        for (int t = 0; t < T; ++t) {
            memset(&rg[t], 0, sizeof rg[t]);
            rg[t].base = base;
            rg[t].a = cut[t];
            rg[t].b = cut[t + 1];
        }
        team_run(legend_count, rg, T);
        size_t total = 0;
        for (int t = 0; t < T; ++t) {
            rg[t].first = total;
            total += rg[t].lines;
        }
        rows = ls_xrealloc(rows, (total ? total : 1) * sizeof *rows);
        team_run(legend_parse, rg, T);
        size_t atotal = arena_len;
        for (int t = 0; t < T; ++t) {
            rg[t].arena_base = atotal;
            atotal += rg[t].part.arena_len;
        }
        if (atotal > arena_cap) {
            arena_cap = atotal;
            arena = ls_xrealloc(arena, arena_cap);
        }
        arena_len = atotal;
        team_run(legend_place, rg, T);
        ls_unmap(base, size);
        j->n = total;
        return NULL;
    }
    ls_unmap(base, size);
    line_src *leg = ls_open(j->fn);
    if (!leg) {
        j->rc = 1;
        return NULL;
    }
    legend_part t;
    memset(&t, 0, sizeof t);
    t.arena = arena;
    t.arena_len = arena_len;
    t.arena_cap = arena_cap;
    ls_next(leg, NULL);                                   /* legend header (src/ibdgem.c:555) */
    for (;;) {
        char *l = ls_next(leg, NULL);
        if (!l)
            break;
        part_add_legend_line(&t, l);
    }
    ls_close(leg);
    rows = t.rows;
    arena = t.arena;
    arena_len = t.arena_len;
    arena_cap = t.arena_cap;
    j->n = t.n;
    return NULL;
}

static int default_threads(void)
{
    cpu_set_t set;
    int n = 1;
    if (sched_getaffinity(0, sizeof set, &set) == 0)
        n = CPU_COUNT(&set);
    return n < 1 ? 1 : n > 16 ? 16 : n;
}

static int read_genotypes(const char *hap_fn, const char *legend_fn, unsigned n_ids)
{
    row_words = ibdg_row_words(n_ids);
    /* the .hap team and the legend team share the threads: the legend is ~1/400 of the text, but its rows
     * cost a sscanf each; with the packed-panel cache the legend is all there is to parse */
    legend_job lj = {legend_fn, 0, 0, opt_threads > 0 ? opt_threads : default_threads()};
    pthread_t lt;
    const int threaded = pthread_create(&lt, NULL, read_legend, &lj) == 0;
    if (!threaded)
        read_legend(&lj);
    uint8_t *ok = NULL;
    size_t n_hap = 0;
    int rc = 1;
    if (cache_fn)
        rc = ingest_cache_open(cache_fn, hap_fn, n_ids, &packed_fd, &packed_off, &ok, &alt_count_h, &n_hap);
    if (!rc) {
        packed_file_bytes = n_hap * (size_t)row_words * 8;
        if (getenv("IBDGEM_CACHE_MAP"))         /* measurement only: the rows as a mapping from the start, as until round 4 */
            (void)packed_rows();
    }
    if (rc) {
        packed_fd = -1;
        const int team = opt_threads > 0 ? opt_threads : default_threads();
        rc = ingest_hap(hap_fn, n_ids, team, &packed, &ok, &n_hap);
        if (!rc) {
            alt_count_h = malloc((n_hap ? n_hap : 1) * sizeof *alt_count_h);
            if (!alt_count_h)
                rc = 1;
            else
                ingest_alt_counts(packed, n_hap, n_ids, team, alt_count_h);
        }
        if (!rc && cache_fn && ingest_cache_store(cache_fn, hap_fn, n_ids, packed, ok, alt_count_h, n_hap))
            fprintf(stderr, "Could not write the panel cache %s (continuing without it).\n", cache_fn);
    }
    if (threaded)
        pthread_join(lt, NULL);
    if (rc || lj.rc) {
        fprintf(stderr, "[::] ERROR parsing hap/legend/indv data; make sure inputs are valid.\n");
        return 1;
    }
    n_rows = n_hap < lj.n ? n_hap : lj.n;
    for (size_t r = 0; r < n_rows; ++r)
        rows[r].hap_ok = ok[r];
    free(ok);
    if (dump_panel_fn) {
        FILE *f = fopen(dump_panel_fn, "wb");
        if (!f)
            return 1;
        for (size_t r = 0; r < n_rows; ++r)
            fputc(rows[r].hap_ok, f);
        fwrite(packed_rows(), 8, n_rows * row_words, f);
        fclose(f);
    }
    return 0;
}

/* VCF rows -> the same row table (restates the parsing of compare_vcf, reference
 * src/ibdgem.c:272-296, src/ibd-parse.c:113-173): sample names from the #CHROM line; per row the
 * eight fixed columns, FORMAT, then one field per sample whose first three characters must be
 * [01][/|][01].  Rows that do not split into the fixed columns, multi-allelic rows (comma in ALT)
 * and rows with an unparsable genotype are "skipped" rows. */
static int read_genotypes_vcf(const char *vcf_fn, names_t *ids)
{
    line_src *vcf = ls_open(vcf_fn);
    if (!vcf)
        return 1;
    char *line;
    while ((line = ls_next(vcf, NULL)) && strncmp(line, "##", 2) == 0)
        ;
    static const char hdr[] = "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t";
    if (!line || strncmp(line, hdr, sizeof hdr - 1) != 0) {
        fprintf(stderr, "[::] ERROR parsing VCF header.\n");
        return 1;
    }
    {
        char *p = line + sizeof hdr - 1;
        p[strcspn(p, "\n")] = 0;
        size_t cap = 0;
        ids->n = 0;
        ids->names = NULL;
        char *save = NULL;
        for (char *tok = strtok_r(p, "\t", &save); tok; tok = strtok_r(NULL, "\t", &save)) {
            if (ids->n == cap) {
                cap = cap ? cap * 2 : 256;
                ids->names = ls_xrealloc(ids->names, cap * sizeof *ids->names);
            }
            ids->names[ids->n++] = strdup(tok);
        }
        if (ids->n == 0) {
            fprintf(stderr, "[::] ERROR: No samples found.\n");
            return 1;
        }
    }
    const unsigned n_ids = (unsigned)ids->n;
    row_words = ibdg_row_words(n_ids);
    uint8_t *alle = malloc(2 * (size_t)n_ids);
    size_t cap = 0;
    while ((line = ls_next(vcf, NULL))) {
        if (n_rows == cap) {
            cap = cap ? cap * 2 : (1 << 16);
            rows = ls_xrealloc(rows, cap * sizeof *rows);
            packed = ls_xrealloc(packed, cap * row_words * 8);
        }
        row_t *r = &rows[n_rows];
        memset(r, 0, sizeof *r);
        memset(packed + n_rows * row_words, 0, row_words * 8);
        n_rows++;
        line[strcspn(line, "\n")] = 0;
        char *f[9], *p = line;
        int nf = 0;
        while (nf < 9) {                                  /* CHROM POS ID REF ALT QUAL FILTER INFO FORMAT */
            char *t = strchr(p, '\t');
            if (!t)
                break;
            *t = 0;
            f[nf++] = p;
            p = t + 1;
        }
        char *endp;
        r->pos = strtoul(nf > 1 ? f[1] : "", &endp, 10);
        if (nf < 9 || endp == f[1] || !*p)
            continue;                                     /* the reference's sscanf != 8 branch */
        r->legend_ok = 1;
        r->id_off = arena_add(f[2]);
        r->ref_off = arena_add(f[3]);
        r->alt_off = arena_add(f[4]);
        r->qual = atof(f[5]);
        if (strchr(f[4], ','))
            continue;                                     /* not biallelic (src/ibdgem.c:275) */
        unsigned i = 0;
        for (char *tok = p; i < n_ids && tok; ++i) {
            char *t = strchr(tok, '\t');
            if (t)
                *t = 0;
            if (!((tok[0] == '0' || tok[0] == '1') && (tok[1] == '/' || tok[1] == '|') && (tok[2] == '0' || tok[2] == '1')))
                break;
            alle[2 * i] = (uint8_t)(tok[0] - '0');
            alle[2 * i + 1] = (uint8_t)(tok[2] - '0');
            tok = t ? t + 1 : NULL;
        }
        if (i < n_ids) {
            r->gt_failed = 1;
            continue;
        }
        ibdg_pack_alleles(alle, n_ids, packed + (n_rows - 1) * row_words);
        r->hap_ok = 1;
    }
    free(alle);
    ls_close(vcf);
    alt_count_h = malloc((n_rows ? n_rows : 1) * sizeof *alt_count_h);
    if (!alt_count_h)
        return 1;
    ingest_alt_counts(packed, n_rows, n_ids, 1, alt_count_h);
    return 0;
}

static int is_snp(const char *ref, const char *alt)   /* one of A C G T each (src/ibdgem.c:113-119) */
{
    return ref[0] && !ref[1] && strchr("ACGT", ref[0]) && alt[0] && !alt[1] && strchr("ACGT", alt[0]);
}

static inline unsigned row_allele(size_t r, unsigned indiv, unsigned hap)
{
    return (unsigned)(packed_rows()[r * row_words + 2 * (indiv >> 6) + hap] >> (indiv & 63)) & 1u;
}

/* The reference thins reads with libc rand(), never seeded (src/ibdgem.c:132), so its output
 * depends on glibc's default stream: the TYPE_3 additive-feedback generator of random_r.c
 * (r[i] = r[i-3] + r[i-31], state seeded from 1 with the Lehmer step 16807 mod 2^31-1, first
 * 310 values discarded, result = r >> 1).  The stream is restated here instead of calling
 * rand(): the HIP runtime inside this process draws from libc's global generator too, which
 * would shift the sequence.  tests/test_host_cli.py checks it against libc. */
static uint32_t grand_state[34];
static int grand_f = 3, grand_r = 0, grand_ready = 0;

static void grand_seed(uint32_t seed)
{
    int32_t w = (int32_t)(seed ? seed : 1);
    grand_state[0] = (uint32_t)w;
    for (int i = 1; i < 31; ++i) {
        const long hi = w / 127773, lo = w % 127773;
        w = (int32_t)(16807 * lo - 2836 * hi);
        if (w < 0)
            w += 2147483647;
        grand_state[i] = (uint32_t)w;
    }
    grand_f = 3;
    grand_r = 0;
    grand_ready = 1;
    for (int i = 0; i < 310; ++i) {
        grand_state[grand_f] += grand_state[grand_r];
        grand_f = (grand_f + 1) % 31;
        grand_r = (grand_r + 1) % 31;
    }
}

static int glibc_rand(void)
{
    if (!grand_ready)
        grand_seed(1);
    const uint32_t v = grand_state[grand_f] += grand_state[grand_r];
    grand_f = (grand_f + 1) % 31;
    grand_r = (grand_r + 1) % 31;
    return (int)(v >> 1);
}

/* cull_dp, reference src/ibdgem.c:126-137 */
static unsigned cull(unsigned count, double cull_p)
{
    if (cull_p == 1.0)
        return count;
    unsigned kept = 0;
    for (unsigned i = 0; i < count; ++i)
        if ((glibc_rand() / (double)2147483647) < cull_p)
            kept++;
    return kept;
}

/* ---- per-site rows of the tab file, formatted by a team of threads ---------------------------
 * The reference prints one fprintf per row (src/ibdgem.c:731-733): 4M rows x 14 columns per
 * comparison individual.  Here the rows are cut into contiguous ranges, every thread formats its
 * range into a buffer of its own with the same conversions (%lf, %e, %u, %lu through snprintf, so
 * the text is what printf would have written), and the buffers are written in order. */
/* ---- the reference's conversions without going through printf --------------------------------
 * 4M rows x (three %e + one %lf + six integers) per comparison individual is where a table-writing run
 * spends its host time.  These produce the very characters printf produces -- printf rounds the exact
 * binary value to nearest, ties to even -- and hand the rare case they cannot decide to sprintf:
 *   %lf of 0 <= f < 2^40: f * 10^6 exactly in 128-bit integer arithmetic (a double is m * 2^e);
 *   %e: a * 10^(6-E) as a 64 x 64 -> 128-bit integer product with a table of powers of ten (64-bit mantissas,
 *       made in long double); when the result lies within 1e-6 of a rounding boundary or 1e-5 of a power of
 *       ten, sprintf decides.  (Until round 3 the product was taken in long double itself: 62 ns per number
 *       in x87 code against 20 now -- frexp, floor and the control-word dance of the conversion were the time.)
 * `ibdgem --fmt-check N` compares them with sprintf on N random doubles (tests/test_host_cli.py). */
/* put_e6's table of powers of ten is made in long double and split into 64-bit mantissas */
_Static_assert(LDBL_MANT_DIG >= 64, "put_e6 needs a long double with a 64-bit mantissa (x86-64); elsewhere use sprintf");
static long double pow10_tab[700];           /* 10^(i-350) */
static uint64_t p10_m[700];                  /* the same as integers: pow10_tab[i] = p10_m[i] * 2^p10_e[i], top bit of p10_m set */
static int16_t p10_e[700];
static void fmt_init(void)
{
    static int done;
    if (done)
        return;
    pow10_tab[350] = 1.0L;
    for (int i = 1; i <= 349; ++i) {
        pow10_tab[350 + i] = pow10_tab[350 + i - 1] * 10.0L;      /* exact up to 10^27, 1 rounding per step after */
        pow10_tab[350 - i] = 1.0L / pow10_tab[350 + i];
    }
    for (int i = 1; i < 700; ++i) {
        int e;
        const long double m = frexpl(pow10_tab[i], &e);           /* [0.5, 1) */
        p10_m[i] = (uint64_t)ldexpl(m, 64);                       /* exact: the mantissa has 64 bits */
        p10_e[i] = (int16_t)(e - 64);
    }
    done = 1;
}

static char *put_str(char *d, const char *s)
{
    const size_t n = strlen(s);
    memcpy(d, s, n);
    return d + n;
}

static char *put_u64(char *d, unsigned long v)
{
    char t[24];
    int n = 0;
    do {
        t[n++] = (char)('0' + v % 10);
        v /= 10;
    } while (v);
    while (n)
        *d++ = t[--n];
    return d;
}

/* "%lf" (six decimals) */
static char *put_lf6(char *d, double f)
{
    if (!(f >= 0.0) || f >= 1099511627776.0 || (f == 0.0 && signbit(f)))
        return d + sprintf(d, "%lf", f);
    uint64_t bits;
    memcpy(&bits, &f, 8);
    const unsigned ex = (unsigned)(bits >> 52);            /* sign bit is clear */
    const int sh = 1075 - (int)ex;                         /* f = mi * 2^-sh exactly; f < 2^40 -> sh >= 13 */
    uint64_t q = 0;
    if (ex != 0 && sh < 127) {                             /* zero, subnormals and everything below 2^-54 * 10^6 round to 0 */
        const uint64_t mi = (bits & (((uint64_t)1 << 52) - 1)) | ((uint64_t)1 << 52);
        const unsigned __int128 p = (unsigned __int128)mi * 1000000u;   /* f * 10^6 = p / 2^sh; p < 2^73 */
        const unsigned __int128 q128 = p >> sh;            /* < 2^60 */
        const unsigned __int128 rem = p - (q128 << sh), half = (unsigned __int128)1 << (sh - 1);
        q = (uint64_t)q128;
        if (rem > half || (rem == half && (q & 1)))
            q += 1;
    }
    const unsigned long whole = (unsigned long)(q / 1000000u);
    unsigned long frac = (unsigned long)(q % 1000000u);
    d = put_u64(d, whole);
    *d++ = '.';
    for (int k = 5; k >= 0; --k) {
        d[k] = (char)('0' + frac % 10);
        frac /= 10;
    }
    return d + 6;
}

/* "%e" (six decimals, exponent of at least two digits).  v = mv * 2^ev with a 64-bit mv; q = v * 10^(6-E) as the
 * 128-bit product of mv and the table's 64-bit mantissa: its integer part is the seven digits, the next 64 bits
 * the fraction that decides the rounding.  The table entry is within 2^-56 of the power of ten (a rounding per
 * step of fmt_init), so q is off by less than 1e7 * 2^-56 < 2e-10: a fraction within 1e-6 of one half, or a q
 * within 1e-5 of a power of ten, goes to sprintf -- except for 1e-21 <= v < 1e7, where the power of ten is exact
 * and so is the product (most likelihoods; and exactly the values the guard band would catch most often: 1.0,
 * 0.5, 0.25 of rows without reads were 37 % of a table's numbers and all went to sprintf). */
static char *put_e6(char *d, double v)
{
    if (!isfinite(v))
        return d + sprintf(d, "%e", v);
    uint64_t bits;
    memcpy(&bits, &v, 8);
    if (bits >> 63) {
        *d++ = '-';
        bits &= ~((uint64_t)1 << 63);
        v = -v;
    }
    if (bits == 0) {
        memcpy(d, "0.000000e+00", 12);
        return d + 12;
    }
    const int ex = (int)(bits >> 52);
    if (ex == 0)                                           /* subnormal: printf's business */
        return d + sprintf(d, "%e", v);
    const uint64_t mv = ((bits & (((uint64_t)1 << 52) - 1)) | ((uint64_t)1 << 52)) << 11;
    const int ev = ex - 1075 - 11;
    const int b2 = ex - 1023;                              /* 2^b2 <= v < 2^(b2+1) */
    int E = (b2 * 78913) / 262144 - (b2 < 0);              /* floor(b2 * log10 2) or one off: settled below */
    uint64_t I = 0;
    unsigned __int128 prod = 0;
    int s = 0, settled = 0;
    for (int tries = 0; tries < 3 && !settled; ++tries) {
        const int k = 350 + 6 - E;
        if (k < 1 || k > 699)
            return d + sprintf(d, "%e", v);
        prod = (unsigned __int128)mv * p10_m[k];           /* q = prod * 2^-s */
        s = -(ev + p10_e[k]);
        if (s < 65 || s > 126)
            return d + sprintf(d, "%e", v);
        I = (uint64_t)(prod >> s);
        if (I >= 10000000u)
            ++E;
        else if (I < 1000000u)
            --E;
        else
            settled = 1;
    }
    if (!settled)
        return d + sprintf(d, "%e", v);
    unsigned long D;
    if (E <= 6 && E >= 6 - 27) {
        /* 10^0 .. 10^27 are exact in the table, so prod is q itself: round to nearest, ties to even, on all its bits
         * (1.0, 0.5, 0.25 -- rows without reads -- are exactly such ties or exact values) */
        const unsigned __int128 rem = prod & (((unsigned __int128)1 << s) - 1), half = (unsigned __int128)1 << (s - 1);
        D = I + ((rem > half || (rem == half && (I & 1))) ? 1u : 0u);
    } else {
        const uint64_t fr = (uint64_t)((prod << (128 - s)) >> 64);            /* the fraction, 0.64 fixed point */
        const uint64_t tol5 = 184467440737096ull, tol6 = 18446744073710ull, half = (uint64_t)1 << 63;   /* 1e-5, 1e-6 * 2^64 */
        if ((I == 1000000u && fr < tol5) || (I == 9999999u && fr > ~tol5))
            return d + sprintf(d, "%e", v);
        if (fr > half - tol6 && fr < half + tol6)
            return d + sprintf(d, "%e", v);
        D = I + (fr > half ? 1u : 0u);
    }
    if (D >= 10000000ul) {
        D /= 10;
        ++E;
    }
    char t[8];
    for (int k = 6; k >= 0; --k) {
        t[k] = (char)('0' + D % 10);
        D /= 10;
    }
    *d++ = t[0];
    *d++ = '.';
    memcpy(d, t + 1, 6);
    d += 6;
    *d++ = 'e';
    *d++ = E < 0 ? '-' : '+';
    unsigned a = (unsigned)(E < 0 ? -E : E);
    if (a >= 100) {
        *d++ = (char)('0' + a / 100);
        a %= 100;
    }
    *d++ = (char)('0' + a / 10);
    *d++ = (char)('0' + a % 10);
    return d;
}

static int fmt_ll(char *dst, double v, char sep)
{
    char *d = isnan(v) ? put_str(dst, "-nan") : put_e6(dst, v);        /* "-nan": x86 printf for 0/0 */
    *d++ = sep;
    return (int)(d - dst);
}

/* --fmt-check N: put_e6 / put_lf6 against sprintf on N random doubles (several distributions); prints the
 * number of differences (0 expected) and how many values took the sprintf way out */
static int fmt_check(long n)
{
    fmt_init();
    uint64_t x = 0x9E3779B97F4A7C15ull;
    long bad = 0, shown = 0;
    char a[400], b[400];
    for (long i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        double v;
        switch (i % 7) {
        case 0: { uint64_t bits = x; memcpy(&v, &bits, 8); break; }                       /* any bit pattern */
        case 1: v = (double)(x >> 11) / 9007199254740992.0; break;                         /* [0,1) */
        case 2: v = ldexp((double)(x >> 11) / 9007199254740992.0, -(int)(x % 1070)); break; /* likelihood-like */
        case 3: v = (double)(x % 20000001) / 1e7 * pow(10.0, (int)(x >> 40) % 40 - 20); break; /* short decimals */
        case 4: v = (double)(x % 2000001) / 128.0 / 15625.0; break;                         /* %lf ties: k/2^7 scaled */
        case 5: v = ((double)(1000000 + x % 9000000) + 0.5) / (double)(1u << ((x >> 32) % 3)); break; /* %e ties and near-ties */
        default: v = (double)((x >> 20) % 5009) / 5008.0; break;                           /* allele frequencies */
        }
        if (isnan(v))
            continue;
        *put_e6(a, v) = 0;
        sprintf(b, "%e", v);
        int diff = strcmp(a, b) != 0;
        *put_lf6(a, v) = 0;
        sprintf(b, "%lf", v);
        diff |= strcmp(a, b) != 0;
        if (diff) {
            ++bad;
            if (shown++ < 5)
                printf("DIFF %a: %s vs %s\n", v, a, b);
        }
    }
    printf("fmt-check: %ld values, %ld differences\n", n, bad);
    return bad != 0;
}

/* a row that passed the target-independent part of the filter chain */
typedef struct { uint32_t row; uint32_t pu; uint8_t n_ref, n_alt; double f; int f_is_override; } cand_t;
typedef struct {
    /* what a row is made of */
    const cand_t *cand;
    const uint32_t *s_cand;
    const pileup_t *pu;
    const double *site_ll;
    const uint8_t *s_nr, *s_na;
    unsigned tgt;
    int plan;
    /* columns 1-9 of every row as text, when somebody made them once for all comparison individuals (row_prefix_build) */
    const char *pre;
    const uint32_t *pre_off;       /* [n + 1] */
    /* this thread's rows and its text */
    size_t a, b;
    char *buf;
    size_t len, cap;
    int failed;
} fmt_job;

/* columns 1-9 of a row, "%s\t%s\t%lu\t%s\t%s\t%lf\t%u\t%u\t%u\t" (src/ibdgem.c:731-733): none of them depends on the
 * comparison individual */
static char *put_row_prefix(char *d, const pileup_t *pu, const cand_t *c, unsigned nr, unsigned na)
{
    const row_t *R = &rows[c->row];
    const pu_line *pl = &pu->lines[c->pu];
    d = put_str(d, pu->chr_names[pl->chr]); *d++ = '\t';
    d = put_str(d, arena + R->id_off); *d++ = '\t';
    d = put_u64(d, R->pos); *d++ = '\t';
    d = put_str(d, arena + R->ref_off); *d++ = '\t';
    d = put_str(d, arena + R->alt_off); *d++ = '\t';
    d = put_lf6(d, c->f); *d++ = '\t';          /* alt count / (2 N) or the -A value: the division the engine makes too */
    d = put_u64(d, pl->cov); *d++ = '\t';
    d = put_u64(d, nr); *d++ = '\t';
    d = put_u64(d, na); *d++ = '\t';
    return d;
}

static size_t row_prefix_room(const pileup_t *pu, const cand_t *c)
{
    const row_t *R = &rows[c->row];
    /* strings + 3 x %lu/%lf (<= 330 chars for the largest double) + 5 x %u + 3 x %e + separators */
    return strlen(pu->chr_names[pu->lines[c->pu].chr]) + strlen(arena + R->id_off) + strlen(arena + R->ref_off) +
           strlen(arena + R->alt_off) + 512;
}

static void *fmt_rows(void *arg)
{
    fmt_job *j = arg;
    j->failed = 0;
    for (size_t i = j->a; i < j->b; ++i) {
        const cand_t *c = &j->cand[j->s_cand[i]];
        const size_t room = j->pre ? (size_t)(j->pre_off[i + 1] - j->pre_off[i]) + 128 : row_prefix_room(j->pu, c);
        if (j->len + room > j->cap) {
            size_t nc = j->cap ? j->cap * 2 : ((size_t)1 << 20);
            while (j->len + room > nc)
                nc *= 2;
            char *nb = realloc(j->buf, nc);
            if (!nb) {
                j->failed = 1;
                return NULL;
            }
            j->buf = nb;
            j->cap = nc;
        }
        char *d = j->buf + j->len;
        /* "%s\t%s\t%lu\t%s\t%s\t%lf\t%u\t%u\t%u\t%u\t%u" (src/ibdgem.c:731-733) */
        if (j->pre) {
            const size_t len = j->pre_off[i + 1] - j->pre_off[i];
            memcpy(d, j->pre + j->pre_off[i], len);
            d += len;
        } else {
            d = put_row_prefix(d, j->pu, c, j->s_nr[i], j->s_na[i]);
        }
        d = put_u64(d, row_allele(c->row, j->tgt, 0)); *d++ = '\t';
        d = put_u64(d, row_allele(c->row, j->tgt, 1));
        if (j->plan) {
            *d++ = '\n';
        } else {
            *d++ = '\t';
            d += fmt_ll(d, j->site_ll[3 * i], '\t');
            d += fmt_ll(d, j->site_ll[3 * i + 1], '\t');
            d += fmt_ll(d, j->site_ll[3 * i + 2], '\n');
        }
        j->len = (size_t)(d - j->buf);
    }
    return NULL;
}

/* Columns 1-9 of all n rows once, for runs whose comparison individuals share the site list: a team formats contiguous
 * ranges into buffers of their own, which are then joined (offsets per row).  ~50 bytes per row; a table-writing thread
 * then copies a row's prefix instead of converting nine columns again (3 of the 6 conversions of a row that are not %e,
 * and all its string handling). */
typedef struct {
    fmt_job j;                     /* a, b, buf, len, cap as in fmt_rows */
    uint32_t *lens;                /* [n]: shared, every thread writes its own range */
    char *dst;                     /* the joined text (second step) */
    size_t dst_off;
} pre_job;

static void *pre_rows(void *arg)
{
    pre_job *p = arg;
    fmt_job *j = &p->j;
    j->failed = 0;
    for (size_t i = j->a; i < j->b; ++i) {
        const cand_t *c = &j->cand[j->s_cand[i]];
        const size_t room = row_prefix_room(j->pu, c);
        if (j->len + room > j->cap) {
            size_t nc = j->cap ? j->cap * 2 : ((size_t)1 << 20);
            while (j->len + room > nc)
                nc *= 2;
            char *nb = realloc(j->buf, nc);
            if (!nb) {
                j->failed = 1;
                return NULL;
            }
            j->buf = nb;
            j->cap = nc;
        }
        char *d = j->buf + j->len;
        char *e = put_row_prefix(d, j->pu, c, j->s_nr[i], j->s_na[i]);
        p->lens[i] = (uint32_t)(e - d);
        j->len += (size_t)(e - d);
    }
    return NULL;
}

static void *pre_join(void *arg)
{
    pre_job *p = arg;
    memcpy(p->dst + p->dst_off, p->j.buf, p->j.len);
    return NULL;
}

/* 0 on success: *pre_out (malloc'd text) and *off_out (malloc'd, n + 1 offsets); on any failure nothing is kept and the
 * tables are formatted row by row as before */
static int row_prefix_build(fmt_job proto, size_t n, int threads, char **pre_out, uint32_t **off_out)
{
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    pre_job jobs[64];
    pthread_t tid[64];
    int started[64] = {0};
    uint32_t *off = malloc((n + 1) * sizeof *off);
    if (!off)
        return 1;
    for (int t = 0; t < threads; ++t) {
        memset(&jobs[t], 0, sizeof jobs[t]);
        jobs[t].j = proto;
        jobs[t].j.buf = NULL;
        jobs[t].j.cap = jobs[t].j.len = 0;
        jobs[t].j.a = n * (size_t)t / (size_t)threads;
        jobs[t].j.b = n * (size_t)(t + 1) / (size_t)threads;
        jobs[t].lens = off;                                       /* lengths first, offsets after the scan below */
    }
    for (int t = 0; t + 1 < threads; ++t)
        started[t] = pthread_create(&tid[t], NULL, pre_rows, &jobs[t]) == 0;
    for (int t = 0; t < threads; ++t)
        if (!started[t])
            pre_rows(&jobs[t]);
    int bad = 0;
    size_t total = 0;
    for (int t = 0; t < threads; ++t) {
        if (started[t])
            pthread_join(tid[t], NULL);
        bad |= jobs[t].j.failed;
        jobs[t].dst_off = total;
        total += jobs[t].j.len;
    }
    char *pre = !bad && total < 0xffffffffu ? malloc(total ? total : 1) : NULL;
    if (pre) {
        for (int t = 0; t < threads; ++t) {
            jobs[t].dst = pre;
            started[t] = t + 1 < threads && pthread_create(&tid[t], NULL, pre_join, &jobs[t]) == 0;
        }
        for (int t = 0; t < threads; ++t)
            if (!started[t])
                pre_join(&jobs[t]);
        for (int t = 0; t < threads; ++t)
            if (started[t])
                pthread_join(tid[t], NULL);
        uint32_t run = 0;                                         /* lengths -> offsets */
        for (size_t i = 0; i < n; ++i) {
            const uint32_t len = off[i];
            off[i] = run;
            run += len;
        }
        off[n] = run;
    }
    for (int t = 0; t < threads; ++t)
        free(jobs[t].j.buf);
    if (!pre) {
        free(off);
        return 1;
    }
    *pre_out = pre;
    *off_out = off;
    return 0;
}

/* Rounds of 2^18 rows (bounds the text held in memory); the team formats round k+1 into a second set of buffers
 * while this thread writes round k -- a third of the phase was the copy into the page cache with the team idle. */
/* The rows in chunks of 2^14 (1.3 MB of text): a team of threads takes chunk numbers from a counter and formats each into
 * one of a ring of buffers; the calling thread writes the chunks in order as they become ready and hands the buffers back.
 * The team is started once per table and the writer never formats: putting ready text into the file (6 GB/s from one
 * thread, tools/write_floor.c) is the longer of the two jobs once sixteen threads format, so nothing may hold it up.
 * (Until round 3 a team was started per 2^18 rows and the caller wrote one round's text while the next was formatted.) */
typedef struct {
    fmt_job proto;
    size_t n, chunk_rows, n_chunks;
    int ring;                      /* buffers (a chunk uses buffer chunk % ring) */
    fmt_job *slot;                 /* [ring] */
    int *state;                    /* [ring]: 0 free, 1 being formatted, 2 ready */
    size_t next_chunk;             /* the next chunk a formatter takes */
    int failed;
    pthread_mutex_t mu;
    pthread_cond_t cv_ready, cv_free;
} row_pipe;

static void *row_pipe_worker(void *arg)
{
    row_pipe *P = arg;
    for (;;) {
        pthread_mutex_lock(&P->mu);
        size_t c;
        int b;
        for (;;) {
            c = P->next_chunk;
            if (c >= P->n_chunks || P->failed) {
                pthread_mutex_unlock(&P->mu);
                return NULL;
            }
            b = (int)(c % (size_t)P->ring);
            if (P->state[b] == 0)
                break;
            pthread_cond_wait(&P->cv_free, &P->mu);      /* the buffer's previous chunk is not written yet */
        }
        P->next_chunk = c + 1;
        P->state[b] = 1;
        pthread_mutex_unlock(&P->mu);
        fmt_job *j = &P->slot[b];
        j->a = c * P->chunk_rows;
        j->b = j->a + P->chunk_rows < P->n ? j->a + P->chunk_rows : P->n;
        j->len = 0;
        fmt_rows(j);
        pthread_mutex_lock(&P->mu);
        if (j->failed)
            P->failed = 1;
        P->state[b] = 2;
        pthread_cond_broadcast(&P->cv_ready);
        if (P->failed)
            pthread_cond_broadcast(&P->cv_free);
        pthread_mutex_unlock(&P->mu);
    }
}

/* format and write in turns (small tables, one thread, no thread to be had) */
static int write_rows_serial(FILE *tab, fmt_job proto, size_t n)
{
    fmt_job j = proto;
    j.buf = NULL;
    j.cap = 0;
    int rc = 0;
    for (size_t a = 0; a < n && !rc; a += 65536) {
        j.a = a;
        j.b = a + 65536 < n ? a + 65536 : n;
        j.len = 0;
        fmt_rows(&j);
        if (j.failed || fwrite(j.buf, 1, j.len, tab) != j.len)
            rc = 1;
    }
    free(j.buf);
    return rc;
}

static int write_rows_parallel(FILE *tab, fmt_job proto, size_t n, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    /* (IBDGEM_MT_MIN_BYTES in the environment, the tests' switch for the threaded readers, also sends small tables through
     * the pipeline: chunks of 37 rows and a ring short enough that formatters wait for the writer) */
    const int tiny = ls_mt_min_bytes() < 4096;
    if (threads == 1 || (n < 8192 && !tiny))
        return write_rows_serial(tab, proto, n);
    row_pipe P;
    memset(&P, 0, sizeof P);
    P.proto = proto;
    P.n = n;
    P.chunk_rows = tiny ? 37 : 16384;
    P.n_chunks = (n + P.chunk_rows - 1) / P.chunk_rows;
    P.ring = tiny ? threads + 1 : 3 * threads;
    P.slot = calloc((size_t)P.ring, sizeof *P.slot);
    P.state = calloc((size_t)P.ring, sizeof *P.state);
    if (!P.slot || !P.state) {
        free(P.slot);
        free(P.state);
        return 1;
    }
    for (int b = 0; b < P.ring; ++b) {
        P.slot[b] = proto;
        P.slot[b].buf = NULL;
        P.slot[b].cap = 0;
    }
    pthread_mutex_init(&P.mu, NULL);
    pthread_cond_init(&P.cv_ready, NULL);
    pthread_cond_init(&P.cv_free, NULL);
    pthread_t tid[64];
    int n_started = 0;
    for (int t = 0; t < threads; ++t)
        if (pthread_create(&tid[n_started], NULL, row_pipe_worker, &P) == 0)
            ++n_started;
    int rc = 0;
    if (n_started == 0) {
        free(P.slot);
        free(P.state);
        pthread_mutex_destroy(&P.mu);
        pthread_cond_destroy(&P.cv_ready);
        pthread_cond_destroy(&P.cv_free);
        return write_rows_serial(tab, proto, n);
    }
    if (fflush(tab) != 0)
        rc = 1;
    const int fd = fileno(tab);
    for (size_t c = 0; c < P.n_chunks && !rc; ++c) {
        const int b = (int)(c % (size_t)P.ring);
        pthread_mutex_lock(&P.mu);
        while (P.state[b] != 2 && !P.failed)
            pthread_cond_wait(&P.cv_ready, &P.mu);
        const int bad = P.failed || P.state[b] != 2;
        pthread_mutex_unlock(&P.mu);
        if (bad) {
            rc = 1;
            break;
        }
        const fmt_job *j = &P.slot[b];
        for (size_t off = 0; off < j->len;) {        /* straight to the descriptor: stdio would copy 1.3 MB through its buffer */
            const ssize_t w = write(fd, j->buf + off, j->len - off);
            if (w < 0) {
                if (errno == EINTR)
                    continue;
                rc = 1;
                break;
            }
            off += (size_t)w;
        }
        pthread_mutex_lock(&P.mu);
        P.state[b] = 0;
        pthread_cond_broadcast(&P.cv_free);
        pthread_mutex_unlock(&P.mu);
    }
    if (rc) {
        pthread_mutex_lock(&P.mu);
        P.failed = 1;
        pthread_cond_broadcast(&P.cv_free);
        pthread_mutex_unlock(&P.mu);
    }
    for (int t = 0; t < n_started; ++t)
        pthread_join(tid[t], NULL);
    for (int b = 0; b < P.ring; ++b)
        free(P.slot[b].buf);
    free(P.slot);
    free(P.state);
    pthread_mutex_destroy(&P.mu);
    pthread_cond_destroy(&P.cv_ready);
    pthread_cond_destroy(&P.cv_free);
    return rc;
}

/* ---- the rows of the summary file (:751-756), formatted by the same team ---------------------- */
typedef struct {
    size_t a, b;                         /* windows [a, b) */
    const uint32_t *s_row, *w_first, *w_last, *w_ncov;
    const unsigned long *pos_first, *pos_last;   /* the windows' first and last positions, or NULL (looked up row by row) */
    const double *win_ll;
    char *buf;
    size_t len;
} sum_job;

static void *fmt_summary(void *arg)
{
    sum_job *j = arg;
    char *q = j->buf;
    for (size_t w = j->a; w < j->b; ++w) {
        q = put_u64(q, w + 1); *q++ = '\t';
        q = put_u64(q, j->pos_first ? j->pos_first[w] : rows[j->s_row[j->w_first[w]]].pos); *q++ = '\t';
        q = put_u64(q, j->pos_last ? j->pos_last[w] : rows[j->s_row[j->w_last[w]]].pos); *q++ = '\t';
        q += fmt_ll(q, j->win_ll[3 * w], '\t');
        q += fmt_ll(q, j->win_ll[3 * w + 1], '\t');
        q += fmt_ll(q, j->win_ll[3 * w + 2], '\t');
        q = put_u64(q, j->w_ncov[w]); *q++ = '\n';
    }
    j->len = (size_t)(q - j->buf);
    return NULL;
}

/* 160 bytes hold any row: three integers of at most 20 digits, three numbers of at most 24 characters.
 * *buf / *cap: the caller's buffer, kept from one individual to the next (5.5 MB at 35 000 windows: allocated anew for every
 * individual it went back to the system each time -- a map, 600 page faults and an unmap per summary file). */
static int write_summary_parallel(FILE *sum, sum_job proto, size_t n_win, int threads, char **buf, size_t *cap)
{
    sum_job jobs[64];
    pthread_t tid[64];
    int started[64] = {0};
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    const int team = n_win < 2048 ? 1 : threads;
    if (*cap < n_win * 160 + 16) {
        free(*buf);
        *buf = malloc(n_win * 160 + 16);
        *cap = *buf ? n_win * 160 + 16 : 0;
    }
    char *all = *buf;
    if (!all)
        return 1;
    for (int t = 0; t < team; ++t) {
        jobs[t] = proto;
        jobs[t].a = n_win * (size_t)t / (size_t)team;
        jobs[t].b = n_win * (size_t)(t + 1) / (size_t)team;
        jobs[t].buf = all + jobs[t].a * 160;
        jobs[t].len = 0;
    }
    for (int t = 0; t + 1 < team; ++t)
        started[t] = pthread_create(&tid[t], NULL, fmt_summary, &jobs[t]) == 0;
    for (int t = 0; t < team; ++t)
        if (!started[t])
            fmt_summary(&jobs[t]);
    for (int t = 0; t + 1 < team; ++t)
        if (started[t])
            pthread_join(tid[t], NULL);
    /* straight to the descriptor (stdio would copy the 2.5 MB through its buffer); what stdio holds goes first */
    int rc = fflush(sum) != 0;
    const int fd = fileno(sum);
    for (int t = 0; t < team && !rc; ++t)
        for (size_t off = 0; off < jobs[t].len;) {
            const ssize_t w = write(fd, jobs[t].buf + off, jobs[t].len - off);
            if (w < 0) {
                if (errno == EINTR)
                    continue;
                rc = 1;
                break;
            }
            off += (size_t)w;
        }
    return rc;
}

/* ---- one comparison spread over several GPUs (window ranges, host-side gather) ------------ */
typedef struct {
    ibdg_ctx *eng;
    /* inputs: a contiguous slice [a,b) of the comparison's site list */
    const uint32_t *row;
    const uint8_t *nr, *na;
    const double *fo;
    size_t a, b;
    unsigned window;
    /* When the site list does not depend on the comparison individual (no -v, no -D), the engine
     * evaluates a batch of them per call (groups of four share a workgroup in the --LD kernel):
     * the first individual of a batch uploads (once) and runs, the others only fetch their slice. */
    const uint32_t *targets;                 /* the batch */
    size_t n_targets, t_local;               /* its size; which of them this call is for */
    int do_upload, do_run;
    int want_sites;                          /* 0: --summary-only, the per-site values stay on the device */
    const uint8_t *bg_count;
    int pu_id, ld;
    /* outputs, written at the slice's offsets of the comparison-wide arrays */
    double *site_ll;
    uint32_t *w_first, *w_last, *w_ncov;     /* slice-local arrays, window indices local */
    double *win_ll;
    size_t n_win;
    int failed;
    int dev_idx;                             /* which of the run's devices (its slot of win_cache) */
    int same_sites;                          /* the site list is the same for every comparison individual of the run */
    const uint32_t *next_targets;            /* the batch after this one (NULL: none), queued on the device while the host goes */
    size_t n_next;                           /* through this batch's tables -- window tables only (--summary-only) */
} shard_job;

/* Per device: the window table of the site list at hand (first row, last row, covered rows per window -- the same for
 * every comparison individual over that list: fetched once, not once per individual) and a page-locked buffer through
 * which an individual's window likelihoods come back at link speed (0.8 MB each at 35 000 windows: through pageable
 * memory the copy took longer than the individual's share of the engine's run). */
typedef struct {
    size_t n_win;
    uint32_t *first, *last, *ncov;
    int valid;
    double *stage;
    size_t stage_cap;
    /* a batch's window tables, taken off the device in one copy, so that the next batch can run while the host goes through
     * them (one table at a time left the device idle for the 4 ms the host needs per batch of 30, beside 6 ms of running) */
    double *batch_ll;
    size_t batch_cap, batch_T;
    const uint32_t *batch_of;                /* the batch (its targets) the tables belong to; NULL: none */
    const uint32_t *ahead_of;                /* the batch that has been queued ahead on the device; NULL: none */
} win_cache;
static win_cache g_wcache[64];

static void *shard_run(void *arg)
{
    shard_job *j = arg;
    const size_t n = j->b - j->a;
    win_cache *wc = &g_wcache[j->dev_idx & 63];
    j->failed = 1;
    if (j->do_upload) {
        wc->valid = 0;
        if (ibdg_upload_sites(j->eng, j->row + j->a, j->nr + j->a, j->na + j->a, j->fo ? j->fo + j->a : NULL, n, j->window))
            return NULL;
    }
    if (j->do_upload || j->do_run)
        wc->batch_of = NULL;
    if (j->do_run) {
        if (wc->ahead_of != j->targets) {        /* not queued ahead (the first batch, or no look-ahead in this run) */
            if (wc->ahead_of && ibdg_sync(j->eng))
                return NULL;
            if (ibdg_run(j->eng, j->targets, j->n_targets, j->bg_count, j->pu_id, j->ld))
                return NULL;
        }
        wc->ahead_of = NULL;
        if (j->next_targets || j->n_targets > 1) {
            const size_t nw = ibdg_num_windows(j->eng), need = j->n_targets * (nw + 1) * 24;
            if (wc->batch_cap < need) {
                if (wc->batch_ll)
                    ibdg_host_free(wc->batch_ll);
                wc->batch_ll = ibdg_host_alloc(need);
                wc->batch_cap = wc->batch_ll ? need : 0;
            }
            if (wc->batch_ll && !j->want_sites) {
                /* (the copy is as large as the LAST RUN's results: it must be the run of this very batch) */
                if (ibdg_num_targets(j->eng) != j->n_targets || ibdg_get_window_ll_all(j->eng, wc->batch_ll))        /* waits for the run */
                    return NULL;
                wc->batch_of = j->targets;
                wc->batch_T = j->n_targets;
                if (j->next_targets) {                                    /* the next batch runs while this one's files are made */
                    if (ibdg_set_option(j->eng, "async", 1) ||
                        ibdg_run(j->eng, j->next_targets, j->n_next, j->bg_count, j->pu_id, j->ld) ||
                        ibdg_set_option(j->eng, "async", 0))
                        return NULL;
                    wc->ahead_of = j->next_targets;
                }
            }
        }
    }
    j->n_win = ibdg_num_windows(j->eng);
    j->w_first = malloc((j->n_win + 1) * 4);
    j->w_last = malloc((j->n_win + 1) * 4);
    j->w_ncov = malloc((j->n_win + 1) * 4);
    j->win_ll = malloc((j->n_win + 1) * 24);
    if (!j->w_first || !j->w_last || !j->w_ncov || !j->win_ll)
        return NULL;
    if (j->same_sites && wc->valid && wc->n_win == j->n_win) {
        memcpy(j->w_first, wc->first, j->n_win * 4);
        memcpy(j->w_last, wc->last, j->n_win * 4);
        memcpy(j->w_ncov, wc->ncov, j->n_win * 4);
    } else {
        if (ibdg_get_windows(j->eng, j->w_first, j->w_last, j->w_ncov))
            return NULL;
        if (j->same_sites) {
            free(wc->first); free(wc->last); free(wc->ncov);
            wc->first = malloc((j->n_win + 1) * 4);
            wc->last = malloc((j->n_win + 1) * 4);
            wc->ncov = malloc((j->n_win + 1) * 4);
            if (wc->first && wc->last && wc->ncov) {
                memcpy(wc->first, j->w_first, j->n_win * 4);
                memcpy(wc->last, j->w_last, j->n_win * 4);
                memcpy(wc->ncov, j->w_ncov, j->n_win * 4);
                wc->n_win = j->n_win;
                wc->valid = 1;
            }
        }
    }
    if (wc->batch_of == j->targets && j->t_local < wc->batch_T) {
        memcpy(j->win_ll, wc->batch_ll + j->t_local * j->n_win * 3, j->n_win * 24);
        j->failed = 0;
        return NULL;                             /* (window tables only: want_sites is off on this path) */
    }
    if (wc->stage_cap < (j->n_win + 1) * 24) {
        if (wc->stage)
            ibdg_host_free(wc->stage);
        wc->stage = ibdg_host_alloc((j->n_win + 1) * 24);
        wc->stage_cap = wc->stage ? (j->n_win + 1) * 24 : 0;
    }
    if (wc->stage) {
        if (ibdg_get_window_ll(j->eng, j->t_local, wc->stage))
            return NULL;
        memcpy(j->win_ll, wc->stage, j->n_win * 24);
    } else if (ibdg_get_window_ll(j->eng, j->t_local, j->win_ll))
        return NULL;
    if (j->want_sites && ibdg_get_site_ll(j->eng, j->t_local, j->site_ll + 3 * j->a))
        return NULL;
    j->failed = 0;
    return NULL;
}

/* cut points of a site list into `parts` contiguous pieces holding whole windows
 * (a window = `window` consecutive covered rows; uncovered rows stay with the open window) */
static void window_cuts(const uint8_t *nr, const uint8_t *na, size_t n, unsigned window, int parts, size_t *cuts)
{
    size_t covered = 0;
    for (size_t i = 0; i < n; ++i)
        covered += (nr[i] + na[i]) > 0;
    const size_t n_win = (covered + window - 1) / window;
    cuts[0] = 0;
    size_t seen = 0, i = 0;
    for (int p = 1; p < parts; ++p) {
        const size_t want = (n_win * (size_t)p / (size_t)parts) * window;    /* covered rows before this cut */
        while (i < n && seen < want)
            seen += (nr[i] + na[i]) > 0, ++i;
        cuts[p] = want >= covered ? n : i;
    }
    cuts[parts] = n;
}

/* IBDGEM_TIMING=1 in the environment: wall-clock seconds per phase on stderr ("## time <phase> <s>") */
static int timing_on = -1;
static double timing_last;
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static void phase(const char *name)
{
    if (timing_on < 0) {
        const char *e = getenv("IBDGEM_TIMING");
        timing_on = e && *e && *e != '0';
        timing_last = now_s();
        if (timing_on) {
            /* what the loader and the libraries' initialisers took before main: now (since boot) minus the start
             * time of the process (field 22 of /proc/self/stat, in clock ticks since boot) */
            FILE *f = fopen("/proc/self/stat", "r");
            char buf[1024];
            if (f && fgets(buf, sizeof buf, f)) {
                const char *p = strrchr(buf, ')');
                unsigned long long ticks = 0;
                int field = 2;
                for (p = p ? p + 1 : buf; *p && field < 22; ++p)
                    if (*p == ' ')
                        ++field;
                if (sscanf(p, "%llu", &ticks) == 1) {
                    struct timespec ts;
                    clock_gettime(CLOCK_BOOTTIME, &ts);
                    const double since = (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec - (double)ticks / (double)sysconf(_SC_CLK_TCK);
                    fprintf(stderr, "## time process start to main %.4f\n", since > 0 ? since : 0.0);
                }
            }
            if (f)
                fclose(f);
        }
        return;
    }
    if (!timing_on)
        return;
    const double t = now_s();
    fprintf(stderr, "## time %s %.4f\n", name, t - timing_last);
    timing_last = t;
}

/* page-locked memory from the engine when asked for and available, plain memory otherwise (never freed:
 * these arrays live as long as the program) */
static void *io_alloc(size_t bytes, int pinned)
{
    void *p = pinned ? ibdg_host_alloc(bytes) : NULL;
    return p ? p : malloc(bytes);
}

/* the row filter chain up to the read counts (src/ibdgem.c:589-626) for the rows [a, b) */
typedef struct {
    size_t a, b;
    const pileup_t *pu;
    const uint32_t *alt_count;
    unsigned n_ids;
    int in_vcf, has_p, has_A;
    cand_t *cand;               /* this range's candidates, from its own start */
    size_t n;
    uint8_t *row_fate;
} filter_job;

static void *filter_rows(void *arg)
{
    filter_job *j = arg;
    for (size_t r = j->a; r < j->b; ++r) {
        const row_t *R = &rows[r];
        j->row_fate[r] = 2;
        if (!R->hap_ok) { j->row_fate[r] = 0; continue; }
        if (!R->legend_ok) continue;
        const char *ref = arena + R->ref_off, *alt = arena + R->alt_off;
        if (!is_snp(ref, alt)) continue;
        if (j->in_vcf && R->qual < opt_min_qual) continue;                /* -q (src/ibdgem.c:297) */
        const pu_line *pl = pileup_find(j->pu, R->pos);
        if (!pl) continue;
        if (j->has_p && !pos_listed(R->pos)) continue;
        double f = (double)j->alt_count[r] / (int)(j->n_ids * 2);
        int ovr = 0;
        if (j->has_A) {
            const af_rec *a = find_af(R->pos);
            if (a) { f = a->f; ovr = 1; }
        }
        if (f > opt_max_af || f < opt_min_af) continue;
        const unsigned nr = pileup_count(pl, ref[0]), na = pileup_count(pl, alt[0]);
        if (nr + na > opt_max_cov) continue;
        cand_t *c = &j->cand[j->n++];
        c->row = (uint32_t)r; c->pu = (uint32_t)(pl - j->pu->lines); c->n_ref = (uint8_t)nr; c->n_alt = (uint8_t)na;
        c->f = f; c->f_is_override = ovr;
        j->row_fate[r] = 1;
    }
    return NULL;
}

/* ---- a non-LD comparison without a device (BASELINE configs[0]) -----------------------------------
 * The per-row columns (src/ibdgem.c:632-651) from the library's host twins of find_pDgG / find_pDgf /
 * find_pDgIBD1 -- the operations and the bits of the device kernel k_site -- and the window products
 * multiplied in row order (:665-667), rows without reads left out (:657-663). */
typedef struct {
    size_t a, b;
    const cand_t *cand;
    const uint32_t *s_cand;
    const uint8_t *s_nr, *s_na;
    const double *pdg;          /* ibdg_pdg_table */
    unsigned tgt;
    double *site_ll;
} host_site_job;

static void *host_site_rows(void *arg)
{
    host_site_job *j = arg;
    const size_t d = (size_t)opt_max_cov + 1;
    for (size_t i = j->a; i < j->b; ++i) {
        const cand_t *c = &j->cand[j->s_cand[i]];
        const double *p = j->pdg + ((size_t)j->s_nr[i] * d + j->s_na[i]) * 3;
        const unsigned a0 = row_allele(c->row, j->tgt, 0), a1 = row_allele(c->row, j->tgt, 1);
        j->site_ll[3 * i] = ibdg_pdg_ibd0(c->f, p[0], p[1], p[2]);
        j->site_ll[3 * i + 1] = ibdg_pdg_ibd1(a0, a1, c->f, p[0], p[1], p[2]);
        j->site_ll[3 * i + 2] = p[a0 + a1];                                  /* :643-651 */
    }
    return NULL;
}

static void host_nonld(const cand_t *cand, const uint32_t *s_cand, const uint8_t *s_nr, const uint8_t *s_na, size_t n,
                       unsigned tgt, const double *pdg, int threads, double *site_ll,
                       const uint32_t *w_first, const uint32_t *w_last, size_t n_win, double *win_ll)
{
    host_site_job jobs[64];
    pthread_t th[64];
    int T = threads < 1 ? 1 : threads > 64 ? 64 : threads;
    if (n < 65536)
        T = 1;
    for (int t = 0; t < T; ++t) {
        host_site_job *j = &jobs[t];
        j->a = n * (size_t)t / (size_t)T; j->b = n * (size_t)(t + 1) / (size_t)T;
        j->cand = cand; j->s_cand = s_cand; j->s_nr = s_nr; j->s_na = s_na; j->pdg = pdg; j->tgt = tgt;
        j->site_ll = site_ll;
        if (T == 1 || pthread_create(&th[t], NULL, host_site_rows, j) != 0) {
            host_site_rows(j);
            th[t] = pthread_self();
        }
    }
    for (int t = 0; t < T; ++t)
        if (!pthread_equal(th[t], pthread_self()))
            pthread_join(th[t], NULL);
    for (size_t w = 0; w < n_win; ++w) {
        double s0 = 1.0, s1 = 1.0, s2 = 1.0;                                  /* :562 */
        for (size_t i = w_first[w]; i <= w_last[w]; ++i) {
            if (s_nr[i] + s_na[i] == 0)
                continue;
            s0 *= site_ll[3 * i]; s1 *= site_ll[3 * i + 1]; s2 *= site_ll[3 * i + 2];
        }
        win_ll[3 * w] = s0; win_ll[3 * w + 1] = s1; win_ll[3 * w + 2] = s2;
    }
}

/* ---- the panel's way to the device(s) ---------------------------------------------------------------
 * One thread per device, all at once.  With one site list for all comparison individuals (no -v, no -D)
 * and several devices, a device receives only the panel rows of its window range (windows are independent,
 * reference src/ibdgem.c:558-570) and its sites are numbered from the slice's first row;
 * otherwise every device receives the whole panel.  The copies overlap the host's filter chain: the -F/-f
 * filter takes its alt-allele counts from the host side (ingest.c), not from a device. */
typedef struct {
    ibdg_ctx *eng;
    size_t r0, n;                /* panel rows [r0, r0 + n) */
    unsigned n_ids;
    const uint32_t *bg_idx;      /* --reference-order: the -B list in file order */
    size_t bg_n;
    int ref_order, has_B;
    int n_uploads;               /* uploads running at the same time */
    size_t share_sites;          /* --LD comparison individuals that will run on ONE site list (0: each its own) */
    int failed;
    pthread_t th;
    int started;
} upload_job;

static void *upload_run(void *arg)
{
    upload_job *j = arg;
    j->failed = 1;
    /* the per-site table needs LIBD0/1/2 of every row (its AF column the host has itself); --summary-only needs
     * nothing per row: the engine then neither keeps nor computes per-row results beyond the IBD2 pick */
    if (ibdg_set_option(j->eng, "site_results", opt_summary_only ? 0 : 1))
        return NULL;
    /* the uploads of all devices run side by side: each staging team gets its share of the host's threads (and locks
     * as much less memory: two 8 MB buffers per thread) */
    if (j->n_uploads > 1 && ibdg_set_option(j->eng, "stage_workers", j->n_uploads >= 8 ? 1 : 8 / j->n_uploads))
        return NULL;
    /* that many individuals over one site list pay for its compacted tiles several times over: have them from the
     * first batch on (left alone, the engine gets there by itself once its runs have added up: nine batches) */
    if (j->share_sites >= 240 && ibdg_set_option(j->eng, "compact_tiles", 1))
        return NULL;
    /* with a panel cache the engine reads the rows from the file itself (no mapping of 2.56 GB on this side) */
    if (packed_fd >= 0 && !packed
            ? ibdg_upload_panel_fd(j->eng, packed_fd, packed_off + (uint64_t)j->r0 * row_words * 8, j->n, j->n_ids)
            : ibdg_upload_panel(j->eng, packed_rows() + j->r0 * row_words, j->n, j->n_ids))
        return NULL;
    if (j->ref_order) {                                   /* background list in the -B file's order (:741) */
        if (ibdg_set_option(j->eng, "ld_variant", 3) || (j->has_B && ibdg_set_background_order(j->eng, j->bg_idx, j->bg_n)))
            return NULL;
    }
    j->failed = 0;
    return NULL;
}

static void uploads_start(upload_job *u, int n)
{
    for (int d = 0; d < n; ++d) {
        u[d].started = pthread_create(&u[d].th, NULL, upload_run, &u[d]) == 0;
        if (!u[d].started)
            upload_run(&u[d]);
    }
}

/* returns the first device whose upload failed, or -1 */
static int uploads_join(upload_job *u, int n)
{
    int bad = -1;
    for (int d = 0; d < n; ++d) {
        if (u[d].started) {
            pthread_join(u[d].th, NULL);
            u[d].started = 0;
        }
        if (u[d].failed && bad < 0)
            bad = d;
    }
    return bad;
}

/* engine contexts created on a thread of their own while the main thread parses the inputs */
typedef struct {
    int dev[64], n;
    double eps;
    unsigned max_cov;
    ibdg_ctx *eng[64];
    char *err[64];
} dev_job_t;

static pthread_t g_dev_thread;
static int g_dev_started;
/* the panel uploads that run beside the filter chain (one thread per device, inside the GPU runtime) */
static upload_job g_ups[64];
static int g_n_ups, g_uploads_pending;
static void outs_settle(int failing);

/* exit() while another thread is inside the GPU runtime -- its start-up, a panel upload -- is asking for trouble (the
 * runtime's exit handlers would run under live GPU threads), and so is leaving while output threads are still writing:
 * wait for every thread this program started first.  On a failing exit the output files that were opened but not
 * yet emptied are emptied, so that no complete-looking table of an earlier run survives a run that failed. */
static void quit(int code)
{
    if (g_dev_started) {
        g_dev_started = 0;
        pthread_join(g_dev_thread, NULL);
    }
    if (g_uploads_pending) {
        g_uploads_pending = 0;
        (void)uploads_join(g_ups, g_n_ups);
    }
    outs_settle(code != 0);
    exit(code);
}

typedef struct {
    dev_job_t *job;
    int d;
} dev_one_t;

static void *dev_start_one(void *arg)
{
    dev_one_t *o = arg;
    dev_job_t *j = o->job;
    j->eng[o->d] = ibdg_create(j->dev[o->d], j->eps, j->max_cov);
    return NULL;
}

/* The first context brings the runtime up; the others (one per further device: each device's own start-up takes
 * about as long again) are created side by side.  A failure is reported with the engine's message, which is
 * per process: with several failing devices it is one of theirs. */
static void *dev_start(void *arg)
{
    dev_job_t *j = arg;
    dev_one_t one[64];
    pthread_t th[64];
    int started[64] = {0};
    if (j->n < 1)
        return NULL;
    one[0].job = j;
    one[0].d = 0;
    dev_start_one(&one[0]);
    if (!j->eng[0]) {
        j->err[0] = strdup(ibdg_last_error(NULL));
        return NULL;
    }
    for (int d = 1; d < j->n; ++d) {
        one[d].job = j;
        one[d].d = d;
        started[d] = pthread_create(&th[d], NULL, dev_start_one, &one[d]) == 0;
        if (!started[d])
            dev_start_one(&one[d]);
    }
    for (int d = 1; d < j->n; ++d)
        if (started[d])
            pthread_join(th[d], NULL);
    for (int d = 1; d < j->n; ++d)
        if (!j->eng[d]) {
            j->err[d] = strdup(ibdg_last_error(NULL));
            break;
        }
    return NULL;
}

#define TARGET_BATCH 30      /* comparison individuals per engine call when their site lists coincide: two groups of k_ld_mfma */
#define DIE(...)                          \
    do {                                  \
        fprintf(stderr, __VA_ARGS__);     \
        quit(1);                          \
    } while (0)


/* ---- the two output files of one comparison individual (src/ibdgem.c:144-152, :547-548, :731-733, :751-768) ---------
 * Everything the files are made of, so that they can be written while the next individual is on the device: with the
 * site list shared by all individuals (no -v, no -D) a default run of many of them is bound by its 330 MB tables, and
 * tables of different individuals go into different files, which take writers side by side (tools/write_floor.c: one
 * file 3-6 GB/s whatever the number of writers, eight files 22-32 GB/s). */
typedef struct {
    /* shared with every other individual of the run (read only while a job runs) */
    const char *out_dir, *user_cmd;
    const unsigned long *in_dist;
    double mean_cov, cull_p;
    const cand_t *cand;
    const uint32_t *s_cand, *s_row;
    const uint8_t *s_nr, *s_na;
    const pileup_t *pu;
    /* this individual */
    const char *tname;
    uint32_t tgt;
    size_t n, n_win;
    unsigned long processed, skipped, final_total, final_dist[128];
    const double *site_ll;                      /* per-row values (NULL with --summary-only) */
    const char *pre;                            /* columns 1-9 of every row as text (row_prefix_build), or NULL */
    const uint32_t *pre_off;
    FILE *tab, *sum;                            /* open (stdout with --plan); closed by the job */
    uint32_t *w_first, *w_last, *w_ncov;        /* owned: freed when the files are closed */
    double *win_ll;
    int threads;
    /* when it runs beside the main thread */
    pthread_t th;
    int running, failed;
    int pending;                                /* the files are open and still hold whatever an earlier run left in them */
    const unsigned long *pos_first, *pos_last;  /* shared: the windows' first / last positions of the common site list, or NULL */
    char *sum_buf;                              /* the slot's buffer for the text of a summary file (kept between individuals) */
    size_t sum_cap;
} out_job;

enum { OUT_SLOTS = 12 };
static out_job g_outs[OUT_SLOTS];

/* see quit(): join the output threads; when the run is failing, empty the files nobody got to */
static void outs_settle(int failing)
{
    for (int k = 0; k < OUT_SLOTS; ++k) {
        out_job *o = &g_outs[k];
        if (o->running) {
            o->running = 0;
            pthread_join(o->th, NULL);
        }
        if (failing && o->pending && !opt_plan) {
            o->pending = 0;
            if (o->tab && !opt_summary_only && ftruncate(fileno(o->tab), 0) != 0)
                fprintf(stderr, "[::] WARNING: could not empty an output file of %s.\n", o->tname);
            if (o->sum && ftruncate(fileno(o->sum), 0) != 0)
                fprintf(stderr, "[::] WARNING: could not empty an output file of %s.\n", o->tname);
        }
    }
}

static void *output_individual(void *arg)
{
    out_job *o = arg;
    o->failed = 1;
    FILE *tab = o->tab, *sum = o->sum;           /* opened by the main thread: a directory that cannot be written stops the run at once */
    if (opt_plan)
        printf("## PLAN %s %s processed=%lu skipped=%lu windows=%zu cull_p=%f\n", opt_sq, o->tname, o->processed, o->skipped,
               o->n_win, o->cull_p);
    else {
        if ((!opt_summary_only && ftruncate(fileno(tab), 0) != 0) || ftruncate(fileno(sum), 0) != 0) {
            fprintf(stderr, "[::] ERROR in compare_impute(): Cannot empty the output files of %s.\n", o->tname);
            return NULL;
        }
        o->pending = 0;
        fprintf(tab, "# Entered command: %s\n\n", o->user_cmd);
    }
    /* header block (:144-152, :547-548) */
    fprintf(tab, "# INPUT COVERAGE DISTRIBUTION:\n# COVERAGE N_SITES\n");
    for (unsigned c = 0; c <= opt_max_cov; ++c)
        fprintf(tab, "# %d %lu\n", c, o->in_dist[c]);
    fprintf(tab, "# MEAN DEPTH = %lf\n# CULL DEPTH RATIO = %lf\n", o->mean_cov, o->cull_p);
    fprintf(tab, "# CHR\trsID\tPOS\tREF\tALT\tAF\tDP\tSQ_NREF\tSQ_NALT\tGT_A0\tGT_A1\tLIBD0\tLIBD1\tLIBD2\n");
    if (!opt_plan)
        fprintf(sum, "# SEGMENT\tSTART\tEND\tLIBD0\tLIBD1\tLIBD2\tNUM_SITES\n");
    if (!opt_summary_only) {
        fmt_job proto;
        memset(&proto, 0, sizeof proto);
        proto.cand = o->cand;
        proto.s_cand = o->s_cand;
        proto.pu = o->pu;
        proto.site_ll = o->site_ll;
        proto.s_nr = o->s_nr;
        proto.s_na = o->s_na;
        proto.tgt = o->tgt;
        proto.plan = opt_plan;
        proto.pre = o->pre;
        proto.pre_off = o->pre_off;
        if (write_rows_parallel(tab, proto, o->n, o->threads)) {
            fprintf(stderr, "[::] ERROR writing the per-site rows of %s.\n", o->tname);
            return NULL;
        }
    }
    if (opt_plan) {
        for (size_t w = 0; w < o->n_win; ++w)
            printf("## WINDOW %zu\t%lu\t%lu\t%u\n", w + 1, rows[o->s_row[o->w_first[w]]].pos, rows[o->s_row[o->w_last[w]]].pos,
                   o->w_ncov[w]);
    } else {
        /* the summary rows (:751-756) through the program's own conversions, by the team of threads: with many
         * comparison individuals per run the seven stdio calls per window were the longest item of an individual
         * (25 ms of 33 at 35 000 windows) */
        sum_job sj;
        memset(&sj, 0, sizeof sj);
        sj.s_row = o->s_row; sj.w_first = o->w_first; sj.w_last = o->w_last; sj.w_ncov = o->w_ncov; sj.win_ll = o->win_ll;
        sj.pos_first = o->pos_first; sj.pos_last = o->pos_last;
        if (write_summary_parallel(sum, sj, o->n_win, o->threads, &o->sum_buf, &o->sum_cap)) {
            fprintf(stderr, "[::] ERROR writing the summary rows of %s.\n", o->tname);
            return NULL;
        }
    }
    /* footer (:761-768) */
    fprintf(tab, "# FINAL COVERAGE DISTRIBUTION:\n# COVERAGE N_SITES\n");
    for (unsigned c = 0; c <= opt_max_cov; ++c)
        fprintf(tab, "# %d %lu\n", c, o->final_dist[c]);
    fprintf(tab, "# FINAL MEAN DEPTH = %lf\n", (double)o->final_total / o->processed);
    fprintf(tab, "## Number of sites processed: %lu\n", o->processed);
    fprintf(tab, "## Number of sites skipped: %lu\n", o->skipped);
    int bad = 0;
    if (!opt_plan) {
        bad |= fclose(tab) != 0;
        bad |= fclose(sum) != 0;
    }
    free(o->w_first); free(o->w_last); free(o->w_ncov); free(o->win_ll);
    o->w_first = o->w_last = o->w_ncov = NULL;
    o->win_ll = NULL;
    o->failed = bad;
    return NULL;
}

int main(int argc, char **argv)
{
    const clock_t t_start = clock();
    phase("start");
#ifdef __GLIBC__
    /* the per-individual arrays (window tables, 1.2 MB) and text buffers come and go once per comparison individual: kept in
     * the heap instead of a map, 300 page faults and an unmap each time (0.3 ms of an individual's 0.7 in a whole-panel run) */
    mallopt(M_MMAP_THRESHOLD, 256 << 20);
    mallopt(M_TRIM_THRESHOLD, 512 << 20);
#endif
    fmt_init();
    const char *hap_fn = NULL, *legend_fn = NULL, *indv_fn = NULL, *pu_fn = NULL, *vcf_fn = NULL;
    const char *sample_fn = NULL, *sample_csv = NULL, *bg_fn = NULL, *af_fn = NULL, *pos_fn = NULL;
    const char *uchr = NULL, *out_dir = NULL, *devices_arg = "0";
    char cwd[PATH_MAX];
    if (argc == 1)
        usage(0);
    out_dir = getcwd(cwd, sizeof cwd) ? cwd : "./";

    int o;
    while ((o = getopt_long(argc, argv, ":V:H:L:I:P:w:N:A:S:s:B:p:q:M:F:f:D:O:c:e:vh", longopts, NULL)) != -1) {
        switch (o) {
        case 0: break;
        case 'V': in_vcf = 1; vcf_fn = optarg; break;
        case 'H': in_impute = 1; hap_fn = optarg; break;
        case 'L': in_impute = 1; legend_fn = optarg; break;
        case 'I': in_impute = 1; indv_fn = optarg; break;
        case 'P': pu_fn = optarg; break;
        case 'w': opt_window = atoi(optarg); break;
        case 'N': opt_sq = optarg; break;
        case 'S': has_S = 1; sample_fn = optarg; break;
        case 's': has_s = 1; sample_csv = optarg; break;
        case 'B': has_B = 1; bg_fn = optarg; break;
        case 'A': has_A = 1; af_fn = optarg; break;
        case 'M': opt_max_cov = (unsigned)atoi(optarg); break;
        case 'F': opt_max_af = atof(optarg); break;
        case 'f': opt_min_af = atof(optarg); break;
        case 'p': has_p = 1; pos_fn = optarg; break;
        case 'q': opt_min_qual = atof(optarg); break;
        case 'c': uchr = optarg; break;
        case 'e': opt_eps = atof(optarg); break;
        case 'O': out_dir = optarg; break;
        case 'D': opt_target_dp = atof(optarg); has_D = 1; break;
        case 'v': has_v = 1; break;
        case 'h': usage(0); break;
        case 1001: devices_arg = optarg; break;
        case 1002: opt_threads = atoi(optarg); break;
        case 1003: cache_fn = optarg; break;
        case 1004: dump_panel_fn = optarg; break;   /* test hook: the packed rows + clean flags as a binary file */
        case 1005: exit(fmt_check(atol(optarg)));   /* test hook: the number conversions against printf */
        case 1000:                                  /* test hook: the first N values of the read-thinning stream */
            for (long i = atol(optarg); i > 0; --i)
                printf("%d\n", glibc_rand());
            exit(0);
        case ':': fprintf(stderr, "Option -%c missing required argument.\n", optopt); exit(0);
        case '?':
            if (isprint(optopt))
                fprintf(stderr, "Invalid option -%c.\n", optopt);
            else
                fprintf(stderr, "Invalid option character.\n");
            break;
        default: fprintf(stderr, "[::] ERROR parsing command-line options.\n"); exit(0);
        }
    }
    for (int i = optind; i < argc; ++i)
        fprintf(stderr, "Given extra argument %s.\n", argv[i]);
    /* range checks, messages and exit codes of reference src/ibdgem.c:966-989 */
    if (has_D && opt_target_dp <= 0) { fprintf(stderr, "[::] ERROR: Invalid down-sample coverage (-D) of %.2f (must be > 0).\n", opt_target_dp); exit(0); }
    if (opt_min_af < 0) { fprintf(stderr, "[::] ERROR: Invalid minimum alternate allele frequency (-f) of %.2f (must be >= 0).\n", opt_min_af); exit(0); }
    if (opt_max_af > 1) { fprintf(stderr, "[::] ERROR: Invalid maximum alternate allele frequency (-F) of %.2f (must be <= 1).\n", opt_max_af); exit(0); }
    if (opt_max_cov < 1) { fprintf(stderr, "[::] ERROR: Invalid maximum estimated coverage (-M) of %u (must be >= 1).\n", opt_max_cov); exit(0); }
    if (opt_min_qual < 0) { fprintf(stderr, "[::] ERROR: Invalid genotype quality minimum (-q) of %.2f (must be >= 0).\n", opt_min_qual); exit(0); }
    if (opt_window < 2) { fprintf(stderr, "[::] ERROR: Invalid window size (-w) of %d (must be >= 2).\n", opt_window); exit(0); }
    if (opt_max_cov > 127) { fprintf(stderr, "[::] ERROR: Invalid maximum estimated coverage (-M) of %u (pileup lines hold at most 127 reads).\n", opt_max_cov); exit(0); }

    /* the echoed command: argv joined by single spaces, with a trailing space (:992-997) */
    size_t cmd_len = 2;
    for (int i = 0; i < argc; ++i)
        cmd_len += strlen(argv[i]) + 1;
    char *user_cmd = malloc(cmd_len);
    user_cmd[0] = 0;
    for (int i = 0; i < argc; ++i) {
        strcat(user_cmd, argv[i]);
        strcat(user_cmd, " ");
    }

    if (!pu_fn)
        DIE("[::] ERROR parsing Pileup data; make sure input is valid.\n");
    /* start the device(s) now: ibdg_create takes ~0.2 s of runtime start-up that needs none of the inputs */
    static dev_job_t dev_job;
    if (!opt_plan) {
        dev_job.eps = opt_eps;
        dev_job.max_cov = opt_max_cov;
        char *dl = strdup(devices_arg);
        char *save = NULL;
        for (char *tok = strtok_r(dl, ",", &save); tok && dev_job.n < 64; tok = strtok_r(NULL, ",", &save))
            dev_job.dev[dev_job.n++] = atoi(tok);
        free(dl);
        if (pthread_create(&g_dev_thread, NULL, dev_start, &dev_job) != 0)
            DIE("[::] ERROR: cannot start a worker thread.\n");
        g_dev_started = 1;
    }
    pileup_t *pu = pileup_read_mt(pu_fn, uchr, opt_threads > 0 ? opt_threads : default_threads());
    if (!pu)
        DIE("[::] ERROR parsing Pileup data; make sure input is valid.\n");
    if (has_A && read_af_file(af_fn, uchr))
        quit(1);
    if (has_p) {
        if (read_pos_file(pos_fn, uchr))
            quit(1);
        qsort(pos_tab, pos_n, sizeof *pos_tab, cmp_ul);
    }
    if (!in_vcf && !in_impute)
        DIE("[::] ERROR: Missing genotype files.\n");
    if (in_vcf && in_impute)
        DIE("[::] ERROR: 2 types of genotype inputs detected. Please choose either IMPUTE or VCF format.\n");
    if (in_impute && (!hap_fn || !legend_fn || !indv_fn))
        DIE("[::] ERROR parsing hap/legend/indv data; make sure inputs are valid.\n");

    names_t ids;
    if (in_vcf) {
        if (read_genotypes_vcf(vcf_fn, &ids))
            quit(1);
    } else {
        if (read_names(indv_fn, &ids))
            quit(1);
        if (ids.n == 0)
            DIE("[::] ERROR: No samples found in .indv file.\n");
    }
    const unsigned n_ids = (unsigned)ids.n;
    idlist_t targets = {NULL, 0}, bg = {NULL, 0};
    if (has_S) {                                            /* -S wins over -s (:1135-1155) */
        if (read_idlist_file(sample_fn, &ids, "Sample %s not found in reference panel.\n", "read_sf", &targets))
            quit(1);
    } else if (has_s) {
        if (read_idlist_csv(sample_csv, &ids, &targets))
            quit(1);
    } else {
        targets.n = n_ids;
        targets.idx = malloc(n_ids * sizeof *targets.idx);
        for (unsigned i = 0; i < n_ids; ++i)
            targets.idx[i] = i;
    }
    uint8_t *bg_count = NULL;
    if (has_B) {
        if (read_idlist_file(bg_fn, &ids, "Reference sample %s not found in input panel.\n", "read_rf", &bg))
            quit(1);
        bg_count = calloc(n_ids, 1);
        for (size_t i = 0; i < bg.n; ++i) {
            /* the engine takes one byte per individual: a list naming somebody 256 times is refused, not truncated */
            if (bg_count[bg.idx[i]] == 255)
                DIE("[::] ERROR: Reference sample %s is listed more than 255 times in %s.\n", ids.names[bg.idx[i]], bg_fn);
            bg_count[bg.idx[i]]++;
        }
    }
    const long pu_id = find_name(&ids, opt_sq);            /* is the pileup's own name in the panel? (:501-506) */
    phase("options, pileup, names");

    if (in_impute && read_genotypes(hap_fn, legend_fn, n_ids))
        quit(1);
    phase("genotypes (hap or cache, legend)");

    /* input coverage distribution and cull ratio: find_cull_p (:83-106) */
    unsigned long in_dist[128] = {0}, in_total = 0;
    for (size_t i = 0; i < pu->n_lines; ++i)
        if (pu->lines[i].cov <= opt_max_cov) {
            in_total += pu->lines[i].cov;
            in_dist[pu->lines[i].cov]++;
        }
    const double mean_cov = (double)in_total / pu->n_lines;
    double cull_p = 1;
    if (has_D) {
        if (opt_target_dp > mean_cov)
            fprintf(stderr, "Observed depth is lower than target depth -D. No culling will be done.\n");
        else
            cull_p = opt_target_dp / mean_cov;
    }

    /* ---- engine: panel upload, alt counts back for the AF filter ---------------------- */
    ibdg_ctx *engs[64];
    int n_eng = 0;
    int host_math = 0;              /* no device and a non-LD run: the library's host twins do the arithmetic */
    double *pdg_tab = NULL;
    if (!opt_plan) {
        /* the contexts were being created (device start-up, ~0.2 s) while the inputs were parsed */
        pthread_join(g_dev_thread, NULL);
        g_dev_started = 0;
        phase("device start (the part not hidden behind parsing)");
        if (dev_job.n > 0 && !dev_job.eng[0] && !opt_ld && ibdg_device_count() == 0) {
            const size_t d = (size_t)opt_max_cov + 1;
            pdg_tab = malloc(d * d * 3 * sizeof *pdg_tab);
            if (!pdg_tab || ibdg_pdg_table(opt_eps, opt_max_cov, pdg_tab))
                DIE("[::] ERROR: cannot build the P(D|G) table.\n");
            host_math = 1;
            dev_job.n = 0;
            fprintf(stderr, "No HIP device found: per-row and window likelihoods are computed on the host (non-LD runs only).\n");
        }
    }
    const int no_engine = opt_plan || host_math;
    upload_job *const ups = g_ups;
    /* the same rows for every comparison individual unless -v looks at its genotype or -D thins the
     * reads anew for each (src/ibdgem.c:584, :627-628) */
    const int batchable = !no_engine && !has_v && cull_p == 1.0;
    int slice_mode = 0;             /* several devices and one site list: every device holds its window range's rows only */
    if (!no_engine) {
        for (int d = 0; d < dev_job.n; ++d) {
            ibdg_ctx *e = dev_job.eng[d];
            if (!e)
                DIE("%s\n", dev_job.err[d] ? dev_job.err[d] : "[::] ERROR in ibdg_create");
            engs[n_eng++] = e;
        }
        if (n_eng == 0)
            DIE("[::] ERROR: --devices needs at least one device index.\n");
        slice_mode = batchable && n_eng > 1;
        for (int d = 0; d < n_eng; ++d) {
            memset(&ups[d], 0, sizeof ups[d]);
            ups[d].eng = engs[d]; ups[d].r0 = 0; ups[d].n = n_rows; ups[d].n_ids = n_ids;
            ups[d].ref_order = opt_ref_order; ups[d].has_B = has_B; ups[d].bg_idx = bg.idx; ups[d].bg_n = bg.n;
            ups[d].n_uploads = n_eng;
            ups[d].share_sites = batchable && opt_ld && !opt_ref_order ? targets.n : 0;
        }
        if (!slice_mode) {
            /* whole panel to every device, all copies at once, under the filter chain below */
            g_n_ups = n_eng;
            uploads_start(ups, n_eng);
            g_uploads_pending = 1;
        }
    }
    if (!alt_count_h)
        DIE("[::] ERROR: no genotype rows.\n");
    const uint32_t *alt_count = alt_count_h;

    /* ---- target-independent part of the row filter chain (:589-626) -------------------
     * Rows are independent here: a team of threads takes contiguous row ranges, each fills its own part
     * of cand[] from the range's first row on, and the parts are closed up in order afterwards. */
    cand_t *cand = malloc((n_rows ? n_rows : 1) * sizeof *cand);
    uint8_t *row_fate = calloc(n_rows ? n_rows : 1, 1);   /* 0 skip-before-v, 1 candidate, 2 skipped after the -v test */
    size_t n_cand = 0;
    {
        int T = opt_threads > 0 ? opt_threads : default_threads();
        if (T > 64) T = 64;
        if (n_rows * 64 < ls_mt_min_bytes()) T = 1;           /* small inputs (and the tests, unless they ask) on one thread */
        filter_job fj[64];
        pthread_t th[64];
        for (int t = 0; t < T; ++t) {
            filter_job *j = &fj[t];
            j->a = n_rows * (size_t)t / (size_t)T;
            j->b = n_rows * (size_t)(t + 1) / (size_t)T;
            j->pu = pu; j->alt_count = alt_count; j->n_ids = n_ids; j->in_vcf = in_vcf; j->has_p = has_p; j->has_A = has_A;
            j->cand = cand + j->a; j->row_fate = row_fate; j->n = 0;
            if (T == 1 || pthread_create(&th[t], NULL, filter_rows, j) != 0) {
                filter_rows(j);
                th[t] = pthread_self();
            }
        }
        for (int t = 0; t < T; ++t) {
            if (!pthread_equal(th[t], pthread_self()))
                pthread_join(th[t], NULL);
            if (fj[t].cand != cand + n_cand)
                memmove(cand + n_cand, fj[t].cand, fj[t].n * sizeof *cand);
            n_cand += fj[t].n;
        }
    }
    phase("row filter chain");

    /* ---- per comparison individual (:522-773) ------------------------------------------ */
    /* the arrays that cross the engine's boundary live in page-locked memory when a device is in use
     * (ibdg_host_alloc: copies at link speed, 45+ GB/s instead of ~17 through a staging buffer) */
    /* Round 2 page-locked these arrays (ibdg_host_alloc).  Measured since: locking 128 MB costs 0.1 s, giving it
     * back at exit 0.2 s, and the copies it was meant to speed up (24 MB in, 96 MB out per comparison) run at the same
     * 56 GB/s from ordinary memory (bench.py results_to_host) -- so they are ordinary memory now. */
    const int pin = 0;
    uint32_t *s_row = io_alloc((n_cand ? n_cand : 1) * 4, pin), *s_cand = malloc((n_cand ? n_cand : 1) * 4);
    uint8_t *s_nr = io_alloc(n_cand ? n_cand : 1, pin), *s_na = io_alloc(n_cand ? n_cand : 1, pin);
    double *s_fo = has_A ? malloc((n_cand ? n_cand : 1) * 8) : NULL;
    /* the per-site values are only fetched for the per-site table: no 128 MB of page-locked memory for --summary-only */
    const size_t n_site_out = opt_summary_only && !no_engine ? 1 : (n_cand ? n_cand : 1);
    double *site_ll = io_alloc(n_site_out * 24, pin);
    if (!s_row || !s_cand || !s_nr || !s_na || !site_ll)
        DIE("[::] ERROR: out of memory for %zu rows.\n", n_cand);
    phase("result arrays");
    uint32_t *s_row_dev = NULL;     /* slice_mode: the site list's rows counted from each device's first row */
    /* The files of up to out_slots individuals are written beside the main thread's work on the ones after them, each from
     * a per-row array of its own -- when the site list is the same for all of them (it is read by the writers), the rows go
     * to files (stdout keeps its order) and there is a table to write at all. */
    out_job *const outs = g_outs;
    /* (--summary-only: the summary files alone, 2.5 MB each at 35 000 windows -- 1.5 ms per individual when written one
     * after the other, most of a whole-panel job whose engine time is 0.2 ms per individual) */
    const int overlap = !has_v && cull_p == 1.0 && !opt_plan && targets.n > 1;
    double *site_slot[OUT_SLOTS] = {site_ll};
    /* (--summary-only: 2.5 MB per individual instead of 330: twelve individuals at a time with four formatter threads each --
     * 0.52 ms per individual in a run of 960 against 0.65 with six and eight, 0.90 with four, tools/many_summaries.py) */
    int out_slots = opt_summary_only ? 12 : 4;
    char *row_pre = NULL;                       /* columns 1-9 of every row as text, shared by all individuals' tables */
    uint32_t *row_pre_off = NULL;                  /* (IBDGEM_OUT_SLOTS=1..12, default 4, 12 with --summary-only: for the tests and for measurements) */
    if (getenv("IBDGEM_OUT_SLOTS") && atoi(getenv("IBDGEM_OUT_SLOTS")) >= 1 && atoi(getenv("IBDGEM_OUT_SLOTS")) <= OUT_SLOTS)
        out_slots = atoi(getenv("IBDGEM_OUT_SLOTS"));
    const int out_threads_env = getenv("IBDGEM_OUT_THREADS") ? atoi(getenv("IBDGEM_OUT_THREADS")) : 0;   /* (measurement switch) */
    for (size_t ti = 0; ti < targets.n; ++ti) {
        const uint32_t tgt = targets.idx[ti];
        if (overlap) {
            out_job *prev = &outs[ti % (size_t)out_slots]; /* the slot's previous individual: its files must be closed */
            if (prev->running) {
                pthread_join(prev->th, NULL);
                prev->running = 0;
                if (prev->failed)
                    quit(1);
                phase("per individual: waiting for the output files of an earlier individual");
            }
            if (!site_slot[ti % (size_t)out_slots]) {
                site_slot[ti % (size_t)out_slots] = io_alloc(n_site_out * 24, pin);
                if (!site_slot[ti % (size_t)out_slots])
                    DIE("[::] ERROR: out of memory for %zu rows.\n", n_cand);
            }
            site_ll = site_slot[ti % (size_t)out_slots];
        }
        const char *tname = ids.names[tgt];
        fprintf(stderr, "Running %s-vs-%s comparison...\n", opt_sq, tname);
        /* Without -v and -D the site list does not depend on the comparison individual (:584, :627-628): it is
         * built for the first one and kept -- 9 ms per individual at 4M rows, more than its engine time.  The
         * reference's message for rows whose genotypes did not parse is repeated per individual as it prints it. */
        static unsigned long skipped, final_total, final_dist[128];
        static size_t n;
        const int same_sites = !has_v && cull_p == 1.0 && ti > 0;
        static size_t n_gt_failed;
        if (same_sites) {
            for (size_t r = 0; r < n_rows && n_gt_failed; ++r)
                if (row_fate[r] == 0 && rows[r].gt_failed)
                    fprintf(stderr, "Failed to parse genotype fields at %lu. Skipping to next site.\n", rows[r].pos);
        } else {
            n_gt_failed = 0;
            skipped = final_total = 0;
            memset(final_dist, 0, sizeof final_dist);
            n = 0;
        }
        for (size_t r = 0, ci = 0; r < n_rows && !same_sites; ++r) {
            if (row_fate[r] == 0) {
                if (rows[r].gt_failed) {
                    fprintf(stderr, "Failed to parse genotype fields at %lu. Skipping to next site.\n", rows[r].pos);
                    n_gt_failed++;
                }
                skipped++;
                continue;
            }
            const int is_cand = row_fate[r] == 1;
            const size_t my = ci;
            if (is_cand) ci++;
            if (has_v && row_allele(r, tgt, 0) == 0 && row_allele(r, tgt, 1) == 0) { skipped++; continue; }   /* :584 */
            if (!is_cand) { skipped++; continue; }
            const cand_t *c = &cand[my];
            const unsigned nr = cull(c->n_ref, cull_p), na = cull(c->n_alt, cull_p);                          /* :627-628 */
            final_total += nr + na;
            final_dist[nr + na]++;
            s_row[n] = c->row; s_cand[n] = (uint32_t)my; s_nr[n] = (uint8_t)nr; s_na[n] = (uint8_t)na;
            if (s_fo) s_fo[n] = c->f_is_override ? c->f : NAN;
            n++;
        }
        const unsigned long processed = n;
        phase("per individual: site list");
        if (ti == 0 && overlap && !opt_summary_only && targets.n >= 3 && n > 0) {
            /* nine of a row's fourteen columns are the same for every comparison individual: their text is made once */
            fmt_job pp;
            memset(&pp, 0, sizeof pp);
            pp.cand = cand; pp.s_cand = s_cand; pp.pu = pu; pp.s_nr = s_nr; pp.s_na = s_na;
            if (row_prefix_build(pp, n, opt_threads > 0 ? opt_threads : default_threads(), &row_pre, &row_pre_off)) {
                row_pre = NULL;
                row_pre_off = NULL;
            }
            phase("columns 1-9 of every row as text, once for all individuals");
        }

        /* windows: runs of opt_window covered rows (:572, :657-663, :723-730) */
        size_t n_win = 0;
        uint32_t *w_first = NULL, *w_last = NULL, *w_ncov = NULL;
        double *win_ll = NULL;
        if (no_engine) {
            size_t covered = 0;
            for (size_t i = 0; i < n; ++i)
                covered += (s_nr[i] + s_na[i]) > 0;
            n_win = (covered + opt_window - 1) / opt_window;
            w_first = malloc((n_win + 1) * 4); w_last = malloc((n_win + 1) * 4); w_ncov = calloc(n_win + 1, 4);
            size_t k = 0;
            for (size_t i = 0; i < n; ++i) {
                if (s_nr[i] + s_na[i] == 0) continue;
                const size_t w = k / opt_window;
                if (k % opt_window == 0) w_first[w] = (uint32_t)i;
                w_last[w] = (uint32_t)i;
                w_ncov[w]++;
                k++;
            }
            if (host_math) {
                win_ll = malloc((n_win + 1) * 24);
                host_nonld(cand, s_cand, s_nr, s_na, n, tgt, pdg_tab, opt_threads > 0 ? opt_threads : default_threads(),
                           site_ll, w_first, w_last, n_win, win_ll);
            }
        } else {
            /* one contiguous window range per GPU, evaluated concurrently, gathered in order */
            static size_t cuts[65];                    /* kept with the site list */
            shard_job jobs[64];
            pthread_t th[64];
            if (!same_sites)
                window_cuts(s_nr, s_na, n, (unsigned)opt_window, n_eng, cuts);
            if (slice_mode && ti == 0) {
                /* every device gets the panel rows from its first site's row to its last site's row, and its
                 * sites are numbered within that slice */
                s_row_dev = io_alloc((n ? n : 1) * 4, pin);
                if (!s_row_dev)
                    DIE("[::] ERROR: out of memory for %zu rows.\n", n);
                for (int d = 0; d < n_eng; ++d) {
                    const size_t a = cuts[d], b = cuts[d + 1];
                    ups[d].r0 = a < b ? s_row[a] : 0;
                    ups[d].n = a < b ? (size_t)s_row[b - 1] + 1 - ups[d].r0 : 0;
                    for (size_t i = a; i < b; ++i)
                        s_row_dev[i] = s_row[i] - (uint32_t)ups[d].r0;
                    if (timing_on > 0)
                        fprintf(stderr, "## panel slice of device %d: rows %zu + %zu of %zu\n", d, ups[d].r0, ups[d].n, n_rows);
                }
                g_n_ups = n_eng;
                uploads_start(ups, n_eng);
                g_uploads_pending = 1;
            }
            if (g_uploads_pending) {
                const int bad = uploads_join(ups, n_eng);
                g_uploads_pending = 0;
                if (bad >= 0)
                    DIE("%s\n", ibdg_last_error(engs[bad]));
                phase("panel upload (copy, alt counts, transposition; the part not hidden behind the filter chain)");
            }
            int th_started[64] = {0};
            for (int d = 0; d < n_eng; ++d) {
                shard_job *j = &jobs[d];
                memset(j, 0, sizeof *j);
                j->eng = engs[d]; j->row = slice_mode ? s_row_dev : s_row; j->nr = s_nr; j->na = s_na; j->fo = s_fo;
                j->a = cuts[d]; j->b = cuts[d + 1]; j->window = (unsigned)opt_window;
                j->want_sites = !opt_summary_only;
                if (batchable) {
                    const size_t b0 = ti - ti % TARGET_BATCH;
                    j->targets = targets.idx + b0;
                    j->n_targets = targets.n - b0 < TARGET_BATCH ? targets.n - b0 : TARGET_BATCH;
                    j->t_local = ti - b0;
                    j->do_upload = ti == 0;
                    j->do_run = ti == b0;
                    if (b0 + TARGET_BATCH < targets.n && opt_summary_only) {
                        j->next_targets = targets.idx + b0 + TARGET_BATCH;
                        j->n_next = targets.n - (b0 + TARGET_BATCH) < TARGET_BATCH ? targets.n - (b0 + TARGET_BATCH) : TARGET_BATCH;
                    }
                } else {
                    j->targets = &targets.idx[ti];
                    j->n_targets = 1;
                    j->t_local = 0;
                    j->do_upload = j->do_run = 1;
                }
                j->bg_count = bg_count; j->pu_id = (int)pu_id; j->ld = opt_ld;
                j->dev_idx = d; j->same_sites = batchable;
                j->site_ll = site_ll;
                /* (no thread to be had: the shard runs here -- never exit() while other shard threads are
                 * inside the GPU runtime) */
                th_started[d] = n_eng > 1 && pthread_create(&th[d], NULL, shard_run, j) == 0;
                if (!th_started[d])
                    shard_run(j);
            }
            n_win = 0;
            int shard_failed = -1;
            for (int d = 0; d < n_eng; ++d) {
                if (th_started[d])
                    pthread_join(th[d], NULL);
                if (jobs[d].failed && shard_failed < 0)
                    shard_failed = d;
                n_win += jobs[d].n_win;
            }
            if (shard_failed >= 0)
                DIE("%s\n", ibdg_last_error(jobs[shard_failed].eng));
            if (n_eng == 1 && jobs[0].a == 0) {
                /* one device, the whole site list: its arrays as they are (a copy of 1.2 MB per individual otherwise) */
                w_first = jobs[0].w_first; w_last = jobs[0].w_last; w_ncov = jobs[0].w_ncov; win_ll = jobs[0].win_ll;
            } else {
                w_first = malloc((n_win + 1) * 4); w_last = malloc((n_win + 1) * 4); w_ncov = malloc((n_win + 1) * 4);
                win_ll = malloc((n_win + 1) * 24);
                size_t wo = 0;
                for (int d = 0; d < n_eng; ++d) {
                    shard_job *j = &jobs[d];
                    const uint32_t a0 = (uint32_t)j->a;
                    for (size_t w = 0; w < j->n_win; ++w) {
                        w_first[wo + w] = j->w_first[w] + a0;
                        w_last[wo + w] = j->w_last[w] + a0;
                    }
                    memcpy(w_ncov + wo, j->w_ncov, j->n_win * 4);
                    memcpy(win_ll + 3 * wo, j->win_ll, j->n_win * 24);
                    wo += j->n_win;
                    free(j->w_first); free(j->w_last); free(j->w_ncov); free(j->win_ll);
                }
            }
        }

        /* the positions a summary row names (:751-756) are the same for every individual over a common site list: looked up
         * once -- row by row they are two dependent loads into 160 MB of row records per window and individual */
        static unsigned long *sum_pos_first, *sum_pos_last;
        static size_t sum_pos_n;
        if (overlap && !no_engine && (ti == 0 || sum_pos_n != n_win)) {
            free(sum_pos_first); free(sum_pos_last);
            sum_pos_first = malloc((n_win + 1) * sizeof *sum_pos_first);
            sum_pos_last = malloc((n_win + 1) * sizeof *sum_pos_last);
            sum_pos_n = n_win;
            for (size_t w = 0; w < n_win && sum_pos_first && sum_pos_last; ++w) {
                sum_pos_first[w] = rows[s_row[w_first[w]]].pos;
                sum_pos_last[w] = rows[s_row[w_last[w]]].pos;
            }
        }
        phase("per individual: engine (upload, run, results)");
        out_job *o = &outs[overlap ? ti % (size_t)out_slots : 0];
        o->pos_first = overlap && !no_engine && sum_pos_first && sum_pos_last ? sum_pos_first : NULL;
        o->pos_last = o->pos_first ? sum_pos_last : NULL;
        o->out_dir = out_dir; o->user_cmd = user_cmd; o->in_dist = in_dist; o->mean_cov = mean_cov; o->cull_p = cull_p;
        o->cand = cand; o->s_cand = s_cand; o->s_row = s_row; o->s_nr = s_nr; o->s_na = s_na; o->pu = pu;
        o->tname = tname; o->tgt = tgt; o->n = n; o->n_win = n_win;
        o->processed = processed; o->skipped = skipped; o->final_total = final_total;
        memcpy(o->final_dist, final_dist, sizeof o->final_dist);
        o->site_ll = site_ll;
        o->pre = row_pre; o->pre_off = row_pre_off;
        o->w_first = w_first; o->w_last = w_last; o->w_ncov = w_ncov; o->win_ll = win_ll;
        if (opt_plan) {
            o->tab = o->sum = stdout;
        } else {
            char *tab_fn, *sum_fn;
            if (asprintf(&tab_fn, "%s/%s.%s.tab.txt", out_dir, opt_sq, tname) < 0 ||
                asprintf(&sum_fn, "%s/%s.%s.summary.txt", out_dir, opt_sq, tname) < 0)
                quit(1);
            /* opened here, emptied by whoever writes them: giving back the pages of an earlier run's 330 MB table takes
             * tens of milliseconds, which belong to the individual's output job, not between two engine calls */
            const int tab_fd = open(opt_summary_only ? "/dev/null" : tab_fn, O_WRONLY | O_CREAT, 0666);
            const int sum_fd = open(sum_fn, O_WRONLY | O_CREAT, 0666);
            o->tab = tab_fd >= 0 ? fdopen(tab_fd, "w") : NULL;
            o->sum = sum_fd >= 0 ? fdopen(sum_fd, "w") : NULL;
            o->pending = 1;
            if (!o->tab || !o->sum) {
                fprintf(stderr, "[::] ERROR in compare_impute(): Cannot open '%s' and/or '%s' for writing.\n", tab_fn, sum_fn);
                quit(1);
            }
            free(tab_fn);
            free(sum_fn);
        }
        const int all_threads = opt_threads > 0 ? opt_threads : default_threads();
        /* (tools/many_tables.py: files of 1 / 2 / 3 / 4 / 6 individuals at once with 8 threads each 73* / 66 / 57 / 47 / 54 ms per
         * individual, *16 threads; 4 x 4 threads 58, 3 x 16 threads 57) */
        o->threads = overlap && out_slots > 1 && all_threads > 3 ? (opt_summary_only ? (all_threads + 3) / 4 : all_threads / 2) : all_threads;
        if (overlap && out_threads_env > 0)
            o->threads = out_threads_env;
        o->running = overlap && pthread_create(&o->th, NULL, output_individual, o) == 0;
        if (!o->running) {
            output_individual(o);
            if (o->failed)
                quit(1);
        }
        phase("per individual: output files");
    }
    for (int k = 0; k < OUT_SLOTS; ++k) {
        if (outs[k].running) {
            pthread_join(outs[k].th, NULL);
            outs[k].running = 0;
            if (outs[k].failed)
                quit(1);
        }
    }
    if (overlap)
        phase("output files of the last individuals (written beside the engine's work on the ones after them)");
#if !defined(__SANITIZE_ADDRESS__) && !defined(__SANITIZE_THREAD__)
    if (getenv("IBDGEM_EXIT_PROBE")) {       /* measurement only (tools/warm_phases.py): what of the process's end is the mapping */
        if (packed_mapped_bytes)
            munmap(packed, packed_mapped_bytes);
        phase("probe: munmap of the panel cache");
    }
    if (!getenv("IBDGEM_KEEP_TEARDOWN")) {
        /* every output file is closed: leave without tearing the device contexts and the runtime down -- the driver
         * reclaims the memory of a process that ends, and freeing 5 GB of it buffer by buffer plus the runtime's
         * own exit handlers cost ~0.06 s of a 0.6 s run (the sanitizer builds keep the orderly way out) */
        fprintf(stderr, "Run time: %f minutes.\n", ((double)(clock() - t_start) / CLOCKS_PER_SEC) / 60);
        fflush(NULL);
        _exit(EXIT_SUCCESS);
    }
#endif
    for (int d = 0; d < n_eng; ++d)
        ibdg_destroy(engs[d]);
    phase("engine shutdown");
    free(row_pre);
    free(row_pre_off);
    pileup_free(pu);
    fprintf(stderr, "Run time: %f minutes.\n", ((double)(clock() - t_start) / CLOCKS_PER_SEC) / 60);
    return EXIT_SUCCESS;
}
