#include "pileup.h"
#include "lineio.h"

#include <ctype.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int base_slot(char b)
{
    switch (b) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return -1;
    }
}

/* one whitespace-delimited field, like sscanf("%s") */
static const char *next_field(const char *p, const char **end)
{
    while (*p && isspace((unsigned char)*p))
        ++p;
    if (!*p)
        return NULL;
    const char *q = p;
    while (*q && !isspace((unsigned char)*q))
        ++q;
    *end = q;
    return p;
}

int pileup_parse_line(const char *line, pu_line *out, char *chr_buf)
{
    return pileup_parse_line_to(line, out, chr_buf, stderr);
}

int pileup_parse_line_to(const char *line, pu_line *out, char *chr_buf, FILE *err)
{
    const char *e;
    /* chr, pos, ref, cov  -- sscanf("%s\t%u\t%c\t%u\t") of src/pileup.c:216-220 */
    const char *f_chr = next_field(line, &e);
    if (!f_chr)
        return 2;
    size_t n = (size_t)(e - f_chr);
    if (n > 255)
        n = 255;
    memcpy(chr_buf, f_chr, n);
    chr_buf[n] = 0;
    char *stop;
    const char *f_pos = next_field(e, &e);
    if (!f_pos)
        return 2;
    unsigned long pos = strtoul(f_pos, &stop, 10);
    if (stop == f_pos)
        return 2;
    const char *p = stop;
    while (*p && isspace((unsigned char)*p))
        ++p;
    if (!*p)
        return 2;
    const char ref = *p++;                         /* %c: one character */
    const char *f_cov = next_field(p, &e);
    if (!f_cov)
        return 2;
    unsigned long cov = strtoul(f_cov, &stop, 10);
    if (stop == f_cov)
        return 2;
    if (cov >= 128)                                /* src/pileup.c:223: more than we can handle */
        return 1;
    /* three more fields are required (bases, base quals, map quals), src/pileup.c:232-241 */
    const char *after_cov = stop;
    const char *f_b = next_field(after_cov, &e), *e_b = e;
    if (!f_b)
        return 1;
    const char *f_q = next_field(e_b, &e), *e_q = e;
    if (!f_q)
        return 1;
    const char *f_m = next_field(e_q, &e), *e_m = e;
    if (!f_m)
        return 1;
    memset(out, 0, sizeof *out);
    out->pos = (uint32_t)pos;
    out->cov = (uint8_t)cov;
    if (cov == 0)
        return 0;                                  /* :243-246 */
    unsigned got = 0;
    const size_t len = (size_t)(e_b - f_b);
    size_t i = 0;
    unsigned cnt[4] = {0, 0, 0, 0};
    while (i < len) {
        const char c = f_b[i];
        char b = 0;
        switch (c) {
        case '.': case ',': b = ref; break;        /* the pileup's own REF character, verbatim */
        case 'A': case 'a': b = 'A'; break;
        case 'C': case 'c': b = 'C'; break;
        case 'G': case 'g': b = 'G'; break;
        case 'T': case 't': b = 'T'; break;
        case 'N': case 'n': b = 'N'; break;
        case '*': b = '*'; break;
        case '-': case '+': {                      /* indel: skip the length and that many characters */
            ++i;
            size_t il = 0;
            while (i < len && isdigit((unsigned char)f_b[i]))
                il = il * 10 + (size_t)(f_b[i++] - '0');
            i += il;
            continue;
        }
        case '$': ++i; continue;
        case '^': i += 2; continue;                /* start marker + mapping-quality character */
        default:
            fprintf(err, "Cannot parse %c in reads field\n", c);
            return 1;
        }
        if (got >= 128)
            return 1;
        const int sl = base_slot(b);
        if (sl >= 0)
            cnt[sl]++;
        ++got;
        ++i;
    }
    if (got != cov) {
        fprintf(err, "Incorrect number of bases read in: %s\n", line);
        return 1;
    }
    if ((size_t)(e_q - f_q) != got && (size_t)(e_m - f_m) != got) {
        fprintf(err, "Incorrect number of base or map quals in: %s\n", line);
        return 1;
    }
    for (int k = 0; k < 4; ++k)
        out->n[k] = (uint8_t)cnt[k];
    return 0;
}

/* one line into a growing table: what the loop of init_Pu_chr does with it (src/pileup.c:487-559) */
typedef struct {
    pu_line *lines;
    size_t n, cap;
    char **chr_names;
    size_t n_chr;
} pu_part;

static void part_add_line(pu_part *t, const char *line, const char *chr, FILE *err)
{
    char cb[256];
    if (t->n == t->cap) {
        t->cap = t->cap ? t->cap * 2 : (1 << 16);
        t->lines = ls_xrealloc(t->lines, t->cap * sizeof *t->lines);
    }
    pu_line *l = &t->lines[t->n];
    const int st = pileup_parse_line_to(line, l, cb, err);
    if (st == 2) {
        fprintf(err, "Problem parsing %s\n", line);
        return;
    }
    if (st)
        return;
    if (chr && strcmp(cb, chr) != 0)
        return;
    size_t ci = t->n_chr;
    if (t->n_chr && strcmp(t->chr_names[t->n_chr - 1], cb) == 0)
        ci = t->n_chr - 1;
    else
        for (ci = 0; ci < t->n_chr; ++ci)
            if (strcmp(t->chr_names[ci], cb) == 0)
                break;
    if (ci == t->n_chr) {
        t->chr_names = ls_xrealloc(t->chr_names, (t->n_chr + 1) * sizeof *t->chr_names);
        t->chr_names[t->n_chr++] = strdup(cb);
    }
    l->chr = (uint32_t)ci;
    t->n++;
}

static pileup_t *pileup_finish(pileup_t *pu, const char *fn)
{
    if (pu->n_lines == 0) {
        fprintf(stderr, "[::] ERROR in init_Pu_chr(): Cannot parse mpileup lines from %s.\n", fn);
        pileup_free(pu);
        return NULL;
    }
    for (size_t i = 0; i + 1 < pu->n_lines; ++i)
        if (pu->lines[i].pos > pu->lines[i + 1].pos) {
            fprintf(stderr, "mpileup lines not sorted!\n");
            pileup_free(pu);
            return NULL;
        }
    return pu;
}

pileup_t *pileup_read(const char *fn, const char *chr)
{
    line_src *ls = ls_open(fn);
    if (!ls)
        return NULL;
    pu_part t;
    memset(&t, 0, sizeof t);
    char *line;
    while ((line = ls_next(ls, NULL)))
        part_add_line(&t, line, chr, stderr);
    ls_close(ls);
    pileup_t *pu = calloc(1, sizeof *pu);
    pu->lines = t.lines;
    pu->n_lines = t.n;
    pu->chr_names = t.chr_names;
    pu->n_chr = t.n_chr;
    return pileup_finish(pu, fn);
}

typedef struct {
    const char *base;
    size_t a, b;
    const char *chr;
    pu_part part;
    char *msg;              /* what this range would have written to stderr */
    size_t msg_len;
} pu_job;

static void *pu_worker(void *arg)
{
    pu_job *j = arg;
    FILE *err = open_memstream(&j->msg, &j->msg_len);
    char *buf = NULL;
    size_t cap = 0;
    for (size_t p = j->a; p < j->b;) {
        const char *nl = memchr(j->base + p, '\n', j->b - p);
        const size_t len = nl ? (size_t)(nl - (j->base + p)) + 1 : j->b - p;    /* with its '\n', like ls_next */
        if (len + 1 > cap) {
            cap = (len + 1) * 2;
            buf = ls_xrealloc(buf, cap);
        }
        memcpy(buf, j->base + p, len);
        buf[len] = 0;
        part_add_line(&j->part, buf, j->chr, err ? err : stderr);
        p += len;
    }
    free(buf);
    if (err)
        fclose(err);
    return NULL;
}

pileup_t *pileup_read_mt(const char *fn, const char *chr, int threads)
{
    size_t size = 0;
    const char *base = threads > 1 ? ls_map(fn, &size) : NULL;
    if (!base || size < ls_mt_min_bytes()) {                            /* gzip, small or unmappable: line by line */
        ls_unmap(base, size);
        return pileup_read(fn, chr);
    }
    if (threads > 64)
        threads = 64;
    size_t cut[65];
    pu_job jobs[64];
    pthread_t th[64];
    ls_split_lines(base, size, 0, threads, cut);
    for (int t = 0; t < threads; ++t) {
        memset(&jobs[t], 0, sizeof jobs[t]);
        jobs[t].base = base;
        jobs[t].a = cut[t];
        jobs[t].b = cut[t + 1];
        jobs[t].chr = chr;
        if (pthread_create(&th[t], NULL, pu_worker, &jobs[t]) != 0) {
            pu_worker(&jobs[t]);
            th[t] = pthread_self();
        }
    }
    pileup_t *pu = calloc(1, sizeof *pu);
    size_t total = 0;
    for (int t = 0; t < threads; ++t) {
        if (!pthread_equal(th[t], pthread_self()))
            pthread_join(th[t], NULL);
        total += jobs[t].part.n;
    }
    pu->lines = ls_xmalloc((total ? total : 1) * sizeof *pu->lines);
    for (int t = 0; t < threads; ++t) {
        pu_job *j = &jobs[t];
        if (j->msg_len)
            fwrite(j->msg, 1, j->msg_len, stderr);
        free(j->msg);
        /* chromosome names in order of first appearance over the whole file; lines renumbered to them */
        uint32_t map[256];
        uint32_t *mp = j->part.n_chr <= 256 ? map : ls_xmalloc(j->part.n_chr * sizeof *mp);
        for (size_t c = 0; c < j->part.n_chr; ++c) {
            size_t g = 0;
            for (; g < pu->n_chr; ++g)
                if (strcmp(pu->chr_names[g], j->part.chr_names[c]) == 0)
                    break;
            if (g == pu->n_chr) {
                pu->chr_names = ls_xrealloc(pu->chr_names, (pu->n_chr + 1) * sizeof *pu->chr_names);
                pu->chr_names[pu->n_chr++] = j->part.chr_names[c];
            } else {
                free(j->part.chr_names[c]);
            }
            mp[c] = (uint32_t)g;
        }
        for (size_t i = 0; i < j->part.n; ++i) {
            pu->lines[pu->n_lines] = j->part.lines[i];
            pu->lines[pu->n_lines].chr = mp[j->part.lines[i].chr];
            pu->n_lines++;
        }
        if (mp != map)
            free(mp);
        free(j->part.chr_names);
        free(j->part.lines);
    }
    ls_unmap(base, size);
    return pileup_finish(pu, fn);
}

const pu_line *pileup_find(const pileup_t *pu, unsigned long pos)
{
    size_t lo = 0, hi = pu->n_lines;
    while (lo < hi) {
        const size_t mid = lo + (hi - lo) / 2;
        const unsigned long p = pu->lines[mid].pos;
        if (p == pos)
            return &pu->lines[mid];
        if (p < pos)
            lo = mid + 1;
        else
            hi = mid;
    }
    return NULL;
}

unsigned pileup_count(const pu_line *l, char base)
{
    const int sl = base_slot(base);
    return sl < 0 ? 0u : l->n[sl];
}

void pileup_free(pileup_t *pu)
{
    if (!pu)
        return;
    for (size_t i = 0; i < pu->n_chr; ++i)
        free(pu->chr_names[i]);
    free(pu->chr_names);
    free(pu->lines);
    free(pu);
}
