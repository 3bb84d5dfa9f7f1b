#include "pileup.h"
#include "lineio.h"

#include <ctype.h>
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
            fprintf(stderr, "Cannot parse %c in reads field\n", c);
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
        fprintf(stderr, "Incorrect number of bases read in: %s\n", line);
        return 1;
    }
    if ((size_t)(e_q - f_q) != got && (size_t)(e_m - f_m) != got) {
        fprintf(stderr, "Incorrect number of base or map quals in: %s\n", line);
        return 1;
    }
    for (int k = 0; k < 4; ++k)
        out->n[k] = (uint8_t)cnt[k];
    return 0;
}

pileup_t *pileup_read(const char *fn, const char *chr)
{
    line_src *ls = ls_open(fn);
    if (!ls)
        return NULL;
    pileup_t *pu = calloc(1, sizeof *pu);
    size_t cap = 0;
    char cb[256];
    char *line;
    while ((line = ls_next(ls, NULL))) {
        if (pu->n_lines == cap) {
            cap = cap ? cap * 2 : (1 << 16);
            pu->lines = realloc(pu->lines, cap * sizeof *pu->lines);
        }
        pu_line *l = &pu->lines[pu->n_lines];
        const int st = pileup_parse_line(line, l, cb);
        if (st == 2) {
            fprintf(stderr, "Problem parsing %s\n", line);
            continue;
        }
        if (st)
            continue;
        if (chr && strcmp(cb, chr) != 0)
            continue;
        size_t ci = pu->n_chr;
        if (pu->n_chr && strcmp(pu->chr_names[pu->n_chr - 1], cb) == 0)
            ci = pu->n_chr - 1;
        else
            for (ci = 0; ci < pu->n_chr; ++ci)
                if (strcmp(pu->chr_names[ci], cb) == 0)
                    break;
        if (ci == pu->n_chr) {
            pu->chr_names = realloc(pu->chr_names, (pu->n_chr + 1) * sizeof *pu->chr_names);
            pu->chr_names[pu->n_chr++] = strdup(cb);
        }
        l->chr = (uint32_t)ci;
        pu->n_lines++;
    }
    ls_close(ls);
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
