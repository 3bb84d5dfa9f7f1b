/* ingest.c -- IMPUTE .hap text -> bit-packed site-major panel, many rows at a time.
 *
 * The reference reads one 4*N-character row per SNP with gzgets/fgets and walks its characters
 * for every comparison individual again (src/ibdgem.c:573-574, :771-772; src/file-io.c:20-27;
 * src/ibd-parse.c:91-99): ~5e4 rows/s at N=2504.  Here the file is read once in large blocks,
 * split at newlines, and the rows of a block are packed by a team of threads, 32 characters per
 * step (AVX2 compare + BMI2 bit extract; a scalar loop where the CPU lacks them).  The result is
 * the layout of include/ibdgem_hip.h (word[2*chunk+plane], bit n%64) plus one "row is clean"
 * flag per row with the meaning of ibdg_pack_hap_text's return value: the row has at least
 * 4*N-1 characters and only '0'/'1' at the allele offsets 4n and 4n+2.  Rows that are not clean
 * are stored as zeros.  An optional cache file keeps the packed panel for the next run. */
#include "ingest.h"

#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#define BLOCK_BYTES ((size_t)64 << 20)
#define PAD 64 /* readable bytes past the data of a block (vector loads may run over a row's end) */

static size_t row_words_of(unsigned n_ids) { return 2 * (((size_t)n_ids + 63) / 64); }

/* ---- one row ---------------------------------------------------------------------------- */
static int pack_row_scalar(const char *line, size_t len, unsigned n0, unsigned n_ids, uint64_t *row)
{
    int bad = 0;
    (void)len;
    for (unsigned n = n0; n < n_ids; ++n) {
        const char c0 = line[4 * (size_t)n], c1 = line[4 * (size_t)n + 2];
        const size_t w = 2 * (size_t)(n >> 6);
        const uint64_t bit = 1ull << (n & 63);
        if (c0 == '1') row[w] |= bit; else if (c0 != '0') bad = 1;
        if (c1 == '1') row[w + 1] |= bit; else if (c1 != '0') bad = 1;
    }
    return bad;
}

#if defined(__x86_64__)
/* 8 individuals (32 characters "a b a b ...") per step: byte k of the block is an allele when
 * k%4 is 0 (first haplotype) or 2 (second).  Returns the number of individuals done (a multiple
 * of 8); *bad is set if an allele offset held something other than '0'/'1'. */
__attribute__((target("avx2,bmi2"))) static unsigned pack_row_avx2(const char *line, unsigned n_ids, uint64_t *row,
                                                                   int *bad)
{
    const __m256i one = _mm256_set1_epi8('1'), zero = _mm256_set1_epi8('0');
    const unsigned n_vec = n_ids & ~7u;
    unsigned invalid = 0;
    for (unsigned n = 0; n < n_vec; n += 64) {
        uint64_t h0 = 0, h1 = 0;
        const unsigned lim = n_vec - n < 64 ? n_vec - n : 64;
        for (unsigned j = 0; j < lim; j += 8) {
            const __m256i v = _mm256_loadu_si256((const __m256i *)(line + 4 * (size_t)(n + j)));
            const unsigned m1 = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, one));
            const unsigned m0 = (unsigned)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, zero));
            invalid |= ~(m0 | m1) & 0x55555555u;
            h0 |= (uint64_t)_pext_u32(m1, 0x11111111u) << j;
            h1 |= (uint64_t)_pext_u32(m1, 0x44444444u) << j;
        }
        row[2 * (size_t)(n >> 6)] = h0;
        row[2 * (size_t)(n >> 6) + 1] = h1;
    }
    if (invalid)
        *bad = 1;
    return n_vec;
}
#endif

static int have_avx2_bmi2(void)
{
#if defined(__x86_64__)
    return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
#else
    return 0;
#endif
}

/* line[0..len) without its newline; line has PAD readable bytes after len */
static int pack_row(const char *line, size_t len, unsigned n_ids, uint64_t *row, int simd)
{
    const size_t words = row_words_of(n_ids);
    memset(row, 0, words * 8);
    const size_t need = 4 * (size_t)n_ids - 1;
    if (len < need || memchr(line, 0, need))   /* fgets/strnlen semantics: a NUL ends the row */
        return 1;
    int bad = 0;
    unsigned done = 0;
#if defined(__x86_64__)
    if (simd)
        done = pack_row_avx2(line, n_ids, row, &bad);
#else
    (void)simd;
#endif
    bad |= pack_row_scalar(line, len, done, n_ids, row);
    if (bad)
        memset(row, 0, words * 8);
    return bad;
}

/* ---- a block of rows, in parallel ------------------------------------------------------- */
typedef struct {
    const char *base;
    const size_t *start, *len;   /* per line of the block */
    size_t first, last;          /* this worker's lines */
    unsigned n_ids;
    uint64_t *out;               /* first row of the block */
    uint8_t *ok;
    size_t words;
    int simd;
} job_t;

static void *worker(void *arg)
{
    job_t *j = arg;
    for (size_t i = j->first; i < j->last; ++i)
        j->ok[i] = pack_row(j->base + j->start[i], j->len[i], j->n_ids, j->out + i * j->words, j->simd) == 0;
    return NULL;
}

typedef struct {
    FILE *f;
    gzFile gz;
} src_t;

static long src_read(src_t *s, char *dst, size_t n)
{
    if (s->gz) {
        size_t got = 0;
        while (got < n) {
            const unsigned ask = (unsigned)((n - got) < (1u << 30) ? (n - got) : (1u << 30));
            const int r = gzread(s->gz, dst + got, ask);
            if (r < 0)
                return -1;
            if (r == 0)
                break;
            got += (size_t)r;
        }
        return (long)got;
    }
    const size_t r = fread(dst, 1, n, s->f);
    if (r < n && ferror(s->f))
        return -1;
    return (long)r;
}

static int name_is_gz(const char *fn)
{
    const size_t n = strlen(fn);
    return n >= 3 && strcmp(fn + n - 3, ".gz") == 0;
}

/* ---- streamed input (gzip, pipes): blocks of text, the rows of a block packed by the team ---- */
static int ingest_stream(const char *fn, unsigned n_ids, int threads, uint64_t **packed_out, uint8_t **ok_out,
                         size_t *n_rows_out)
{
    const size_t max_rows = (size_t)-1;
    src_t src = {NULL, NULL};
    if (name_is_gz(fn)) {
        src.gz = gzopen(fn, "r");
        if (src.gz)
            gzbuffer(src.gz, 1 << 20);
    } else {
        src.f = fopen(fn, "r");
        if (!src.f) {
            fprintf(stderr, "Failed to open %s.\n", fn);
            perror("Error");
        }
    }
    if (!src.f && !src.gz)
        return 1;
    const int simd = have_avx2_bmi2();
    const size_t words = row_words_of(n_ids);
    size_t cap_rows = 0, n_rows = 0;
    uint64_t *packed = NULL;
    uint8_t *ok = NULL;
    size_t buf_cap = BLOCK_BYTES;
    char *buf = malloc(buf_cap + PAD);
    size_t *start = NULL, *len = NULL, line_cap = 0;
    size_t carry = 0;            /* bytes of an unfinished line at the front of buf */
    int eof = 0, rc = 0;
    if (!buf)
        rc = 1;
    while (!rc && !(eof && carry == 0) && n_rows < max_rows) {
        if (carry == buf_cap) {  /* one line longer than the block: grow */
            buf_cap *= 2;
            char *nb = realloc(buf, buf_cap + PAD);
            if (!nb) { rc = 1; break; }
            buf = nb;
        }
        size_t have = carry;
        if (!eof) {
            const long got = src_read(&src, buf + carry, buf_cap - carry);
            if (got < 0) { rc = 1; break; }
            have += (size_t)got;
            if ((size_t)got < buf_cap - carry)
                eof = 1;
        }
        memset(buf + have, 0, PAD);
        /* split into lines; an unfinished last line is carried over unless the file ended */
        size_t n_lines = 0, pos = 0;
        while (pos < have) {
            const char *nl = memchr(buf + pos, '\n', have - pos);
            if (!nl && !eof)
                break;
            const size_t end = nl ? (size_t)(nl - buf) : have;
            if (n_lines == line_cap) {
                line_cap = line_cap ? line_cap * 2 : 8192;
                start = realloc(start, line_cap * sizeof *start);
                len = realloc(len, line_cap * sizeof *len);
                if (!start || !len) { rc = 1; break; }
            }
            start[n_lines] = pos;
            len[n_lines] = end - pos;
            n_lines++;
            pos = nl ? end + 1 : have;
        }
        if (rc)
            break;
        if (n_lines > max_rows - n_rows)
            n_lines = max_rows - n_rows;
        if (n_rows + n_lines > cap_rows) {
            cap_rows = cap_rows ? cap_rows * 2 : (1 << 16);
            while (cap_rows < n_rows + n_lines)
                cap_rows *= 2;
            uint64_t *np = realloc(packed, cap_rows * words * 8);
            uint8_t *no = realloc(ok, cap_rows);
            if (!np || !no) { free(np ? np : packed); free(no ? no : ok); packed = NULL; ok = NULL; rc = 1; break; }
            packed = np;
            ok = no;
        }
        /* the block's rows, split evenly over the team; the caller takes the last share */
        pthread_t tid[64];
        int started[64] = {0};
        job_t jobs[64];
        const int team = n_lines < (size_t)threads * 4 ? 1 : threads;
        for (int t = 0; t < team; ++t)
            jobs[t] = (job_t){buf, start, len, n_lines * (size_t)t / team, n_lines * (size_t)(t + 1) / team, n_ids,
                              packed + n_rows * words, ok + n_rows, words, simd};
        for (int t = 0; t + 1 < team; ++t)
            started[t] = pthread_create(&tid[t], NULL, worker, &jobs[t]) == 0;
        for (int t = 0; t < team; ++t)
            if (!started[t])
                worker(&jobs[t]);               /* the last share, and any whose thread did not start */
        for (int t = 0; t + 1 < team; ++t)
            if (started[t])
                pthread_join(tid[t], NULL);
        n_rows += n_lines;
        carry = have - pos;
        if (carry)
            memmove(buf, buf + pos, carry);
    }
    free(buf);
    free(start);
    free(len);
    if (src.gz)
        gzclose(src.gz);
    if (src.f)
        fclose(src.f);
    if (rc) {
        free(packed);
        free(ok);
        return 1;
    }
    *packed_out = packed;
    *ok_out = ok;
    *n_rows_out = n_rows;
    return 0;
}

/* ---- plain files: the page cache is mapped and every thread works on a byte range of its own ----
 * Pass 1 counts the lines that START in each range (a line starts at offset 0 or after a '\n'),
 * a prefix sum gives every range its first row, pass 2 packs.  No copy of the text is made. */
typedef struct {
    const char *base;
    size_t size, a, b;           /* the whole file; this thread's byte range [a, b) */
    size_t starts;               /* pass 1: lines starting in [a, b) */
    size_t first_row;            /* pass 2 */
    unsigned n_ids;
    uint64_t *out;
    uint8_t *ok;
    size_t words;
    int simd;
} mjob_t;

static size_t count_newlines(const char *p, size_t n)
{
    size_t c = 0;
    const char *end = p + n;
    while (p < end) {
        const char *q = memchr(p, '\n', (size_t)(end - p));
        if (!q)
            break;
        c++;
        p = q + 1;
    }
    return c;
}

static void *mmap_count(void *arg)
{
    mjob_t *j = arg;
    /* a start at p in [a, b), p > 0, is a newline at p-1 in [a-1, b-1) */
    const size_t lo = j->a ? j->a - 1 : 0, hi = j->b ? j->b - 1 : 0;
    j->starts = (hi > lo ? count_newlines(j->base + lo, hi - lo) : 0) + (j->a == 0 && j->size > 0 ? 1 : 0);
    return NULL;
}

static void *mmap_pack(void *arg)
{
    mjob_t *j = arg;
    size_t p = j->a;
    if (p > 0) {                 /* first line start at or after a */
        const char *q = memchr(j->base + p - 1, '\n', j->size - (p - 1));
        if (!q)
            return NULL;
        p = (size_t)(q - j->base) + 1;
    }
    const size_t need = 4 * (size_t)j->n_ids - 1;
    char *tail = NULL;           /* padded copy for rows too close to the end of the mapping */
    size_t row = j->first_row;
    while (p < j->b && p < j->size) {
        const char *q = memchr(j->base + p, '\n', j->size - p);
        const size_t end = q ? (size_t)(q - j->base) : j->size, len = end - p;
        const char *line = j->base + p;
        int simd = j->simd;
        if (len >= need && p + need + PAD > j->size) {    /* vector loads could leave the mapping */
            if (!tail)
                tail = calloc(1, need + 1 + PAD);
            if (tail) {
                memcpy(tail, line, need);
                line = tail;
            } else {
                simd = 0;                                   /* the scalar loop stays inside the row */
            }
        }
        j->ok[row] = pack_row(line, len, j->n_ids, j->out + row * j->words, simd) == 0;
        row++;
        p = q ? end + 1 : j->size;
    }
    free(tail);
    return NULL;
}

static void run_team(void *(*fn)(void *), mjob_t *jobs, int team)
{
    pthread_t tid[64];
    int started[64] = {0};
    for (int t = 0; t + 1 < team; ++t)
        started[t] = pthread_create(&tid[t], NULL, fn, &jobs[t]) == 0;
    for (int t = 0; t < team; ++t)
        if (!started[t])
            fn(&jobs[t]);
    for (int t = 0; t + 1 < team; ++t)
        if (started[t])
            pthread_join(tid[t], NULL);
}

/* returns 0 ok, 1 error, 2 not mappable (caller falls back to the streamed reader) */
static int ingest_mmap(const char *fn, unsigned n_ids, int threads, uint64_t **packed_out, uint8_t **ok_out,
                       size_t *n_rows_out)
{
    const int fd = open(fn, O_RDONLY);
    if (fd < 0) {
        fprintf(stderr, "Failed to open %s.\n", fn);
        perror("Error");
        return 1;
    }
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        close(fd);
        return 2;
    }
    const size_t size = (size_t)st.st_size;
    const size_t words = row_words_of(n_ids);
    if (size == 0) {
        close(fd);
        *packed_out = malloc(8);
        *ok_out = malloc(1);
        *n_rows_out = 0;
        return !(*packed_out && *ok_out);
    }
    const char *base = mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (base == MAP_FAILED)
        return 2;
    (void)madvise((void *)base, size, MADV_SEQUENTIAL);
    int team = threads;
    if (size < ((size_t)1 << 20))
        team = 1;
    mjob_t jobs[64];
    for (int t = 0; t < team; ++t) {
        memset(&jobs[t], 0, sizeof jobs[t]);
        jobs[t].base = base;
        jobs[t].size = size;
        jobs[t].a = size / (size_t)team * (size_t)t;
        jobs[t].b = t + 1 == team ? size : size / (size_t)team * (size_t)(t + 1);
        jobs[t].n_ids = n_ids;
        jobs[t].words = words;
        jobs[t].simd = have_avx2_bmi2();
    }
    run_team(mmap_count, jobs, team);
    size_t n_rows = 0;
    for (int t = 0; t < team; ++t) {
        jobs[t].first_row = n_rows;
        n_rows += jobs[t].starts;
    }
    uint64_t *packed = malloc(n_rows * words * 8 + 8);
    uint8_t *ok = malloc(n_rows + 1);
    int rc = !(packed && ok);
    if (!rc) {
        for (int t = 0; t < team; ++t) {
            jobs[t].out = packed;
            jobs[t].ok = ok;
        }
        run_team(mmap_pack, jobs, team);
    }
    munmap((void *)base, size);
    if (rc) {
        free(packed);
        free(ok);
        return 1;
    }
    *packed_out = packed;
    *ok_out = ok;
    *n_rows_out = n_rows;
    return 0;
}

int ingest_hap(const char *fn, unsigned n_ids, int threads, uint64_t **packed, uint8_t **ok, size_t *n_rows)
{
    if (threads < 1)
        threads = 1;
    if (threads > 64)
        threads = 64;
    if (!name_is_gz(fn)) {
        const int rc = ingest_mmap(fn, n_ids, threads, packed, ok, n_rows);
        if (rc != 2)
            return rc;
    }
    return ingest_stream(fn, n_ids, threads, packed, ok, n_rows);
}

/* ---- alternate-allele counts of the packed rows (find_f_impute / find_f_vcf, reference
 * src/ibd-parse.c:91-110: the number of '1' alleles of a row over ALL individuals) -------------
 * The host needs them for its -F/-f filter before any device has seen the panel (a device gets only
 * the rows of its window range, which are known after the filter); the engine recounts its own rows
 * (k_alt_count) for the per-row arithmetic. */
typedef struct {
    const uint64_t *packed;
    size_t a, b, words;
    uint32_t *out;
} cnt_job;

static void *count_rows(void *arg)
{
    cnt_job *j = arg;
    for (size_t r = j->a; r < j->b; ++r) {
        const uint64_t *row = j->packed + r * j->words;
        unsigned c = 0;
        for (size_t w = 0; w < j->words; ++w)
            c += (unsigned)__builtin_popcountll(row[w]);
        j->out[r] = c;
    }
    return NULL;
}

void ingest_alt_counts(const uint64_t *packed, size_t n_rows, unsigned n_ids, int threads, uint32_t *out)
{
    cnt_job jobs[64];
    pthread_t th[64];
    int started[64] = {0};
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (n_rows < 4096) threads = 1;
    for (int t = 0; t < threads; ++t) {
        jobs[t].packed = packed;
        jobs[t].words = row_words_of(n_ids);
        jobs[t].a = n_rows * (size_t)t / (size_t)threads;
        jobs[t].b = n_rows * (size_t)(t + 1) / (size_t)threads;
        jobs[t].out = out;
    }
    for (int t = 0; t + 1 < threads; ++t)
        started[t] = pthread_create(&th[t], NULL, count_rows, &jobs[t]) == 0;
    for (int t = 0; t < threads; ++t)
        if (!started[t])
            count_rows(&jobs[t]);
    for (int t = 0; t + 1 < threads; ++t)
        if (started[t])
            pthread_join(th[t], NULL);
}

/* ---- cache file --------------------------------------------------------------------------
 * header | ok flags | zero padding to a 4096-byte boundary | alt-allele counts (u32 per row) | padding |
 * packed rows.  The rows are not read but mapped: loading the cache costs nothing up front, and the
 * engine's host-to-device copy (a team of threads staging through page-locked buffers) reads the page
 * cache directly.  The counts are what the host's -F/-f filter needs of the panel, so a warm run
 * starts filtering without touching the 2.56 GB of rows. */
#define CACHE_ALIGN 4096u
#define CACHE_MAGIC "IBDGPNL3"
typedef struct {
    char magic[8];               /* CACHE_MAGIC */
    uint32_t n_ids, reserved;
    uint64_t n_rows, row_words;
    uint64_t src_size;
    int64_t src_mtime_s, src_mtime_ns;
} cache_hdr;

static int stat_src(const char *fn, cache_hdr *h)
{
    struct stat st;
    if (stat(fn, &st) != 0)
        return 1;
    h->src_size = (uint64_t)st.st_size;
    h->src_mtime_s = (int64_t)st.st_mtim.tv_sec;
    h->src_mtime_ns = (int64_t)st.st_mtim.tv_nsec;
    return 0;
}

static size_t cache_counts_offset(size_t n_rows)
{
    const size_t raw = sizeof(cache_hdr) + n_rows;
    return (raw + CACHE_ALIGN - 1) / CACHE_ALIGN * CACHE_ALIGN;
}

static size_t cache_rows_offset(size_t n_rows)
{
    const size_t raw = cache_counts_offset(n_rows) + n_rows * 4;
    return (raw + CACHE_ALIGN - 1) / CACHE_ALIGN * CACHE_ALIGN;
}

uint64_t *ingest_cache_map(int fd, uint64_t rows_off, size_t bytes)
{
    if (bytes == 0)
        return malloc(8);
    /* not MAP_POPULATE: whoever reads rows touches the pages it needs */
    void *m = mmap(NULL, bytes, PROT_READ, MAP_PRIVATE, fd, (off_t)rows_off);
    return m == MAP_FAILED ? NULL : m;
}

static int cache_load(const char *cache_fn, const char *hap_fn, unsigned n_ids, uint64_t **packed_out, int *fd_out,
                      uint64_t *off_out, uint8_t **ok_out, uint32_t **alt_out, size_t *n_rows_out);

int ingest_cache_load(const char *cache_fn, const char *hap_fn, unsigned n_ids, uint64_t **packed_out, uint8_t **ok_out,
                      uint32_t **alt_out, size_t *n_rows_out)
{
    return cache_load(cache_fn, hap_fn, n_ids, packed_out, NULL, NULL, ok_out, alt_out, n_rows_out);
}

int ingest_cache_open(const char *cache_fn, const char *hap_fn, unsigned n_ids, int *fd_out, uint64_t *off_out, uint8_t **ok_out,
                      uint32_t **alt_out, size_t *n_rows_out)
{
    return cache_load(cache_fn, hap_fn, n_ids, NULL, fd_out, off_out, ok_out, alt_out, n_rows_out);
}

static int cache_load(const char *cache_fn, const char *hap_fn, unsigned n_ids, uint64_t **packed_out, int *fd_out,
                      uint64_t *off_out, uint8_t **ok_out, uint32_t **alt_out, size_t *n_rows_out)
{
    const int fd = open(cache_fn, O_RDONLY);
    if (fd < 0)
        return 1;
    cache_hdr h, want;
    memset(&want, 0, sizeof want);
    int rc = 1;
    uint8_t *ok = NULL;
    uint32_t *alt = NULL;
    struct stat st;
    if (fstat(fd, &st) == 0 && pread(fd, &h, sizeof h, 0) == (ssize_t)sizeof h && memcmp(h.magic, CACHE_MAGIC, 8) == 0 &&
        stat_src(hap_fn, &want) == 0 && h.n_ids == n_ids && h.row_words == row_words_of(n_ids) &&
        h.src_size == want.src_size && h.src_mtime_s == want.src_mtime_s && h.src_mtime_ns == want.src_mtime_ns) {
        const size_t n = (size_t)h.n_rows, bytes = n * (size_t)h.row_words * 8, off = cache_rows_offset(n);
        ok = malloc(n ? n : 1);
        alt = malloc((n ? n : 1) * sizeof *alt);
        if (ok && alt && (size_t)st.st_size >= off + bytes && pread(fd, ok, n, sizeof h) == (ssize_t)n &&
            pread(fd, alt, n * 4, (off_t)cache_counts_offset(n)) == (ssize_t)(n * 4)) {
            if (fd_out) {                      /* no mapping: the caller gets the open file and where its rows start */
                *fd_out = fd;
                *off_out = (uint64_t)off;
                rc = 0;
            } else if (bytes == 0) {
                *packed_out = malloc(8);
                rc = *packed_out ? 0 : 1;
            } else {
                /* not MAP_POPULATE: the engine's staging team touches the pages from eight threads while it copies,
                 * which costs nothing extra there, where populating up front took 50-60 ms on this thread */
                void *m = mmap(NULL, bytes, PROT_READ, MAP_PRIVATE, fd, (off_t)off);
                if (m != MAP_FAILED) {
                    *packed_out = m;           /* lives as long as the program; never freed by the caller */
                    rc = 0;
                }
            }
            if (!rc) {
                *ok_out = ok;
                *alt_out = alt;
                *n_rows_out = n;
            }
        }
    }
    if (rc || !fd_out)
        close(fd);
    if (rc) {
        free(ok);
        free(alt);
    }
    return rc;
}

int ingest_cache_store(const char *cache_fn, const char *hap_fn, unsigned n_ids, const uint64_t *packed,
                       const uint8_t *ok, const uint32_t *alt, size_t n_rows)
{
    cache_hdr h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, CACHE_MAGIC, 8);
    h.n_ids = n_ids;
    h.n_rows = n_rows;
    h.row_words = row_words_of(n_ids);
    if (stat_src(hap_fn, &h))
        return 1;
    char tmp[4096];
    if (snprintf(tmp, sizeof tmp, "%s.tmp%ld", cache_fn, (long)getpid()) >= (int)sizeof tmp)
        return 1;
    FILE *f = fopen(tmp, "wb");
    if (!f)
        return 1;
    const size_t bytes = n_rows * (size_t)h.row_words * 8;
    const size_t pad = cache_counts_offset(n_rows) - (sizeof h + n_rows);
    const size_t pad2 = cache_rows_offset(n_rows) - (cache_counts_offset(n_rows) + n_rows * 4);
    static const char zeros[CACHE_ALIGN];
    int rc = !(fwrite(&h, sizeof h, 1, f) == 1 && fwrite(ok, 1, n_rows, f) == n_rows &&
               fwrite(zeros, 1, pad, f) == pad && fwrite(alt, 4, n_rows, f) == n_rows &&
               fwrite(zeros, 1, pad2, f) == pad2 && fwrite(packed, 1, bytes, f) == bytes);
    if (fclose(f) != 0)
        rc = 1;
    if (!rc && rename(tmp, cache_fn) != 0)
        rc = 1;
    if (rc)
        remove(tmp);
    return rc;
}
