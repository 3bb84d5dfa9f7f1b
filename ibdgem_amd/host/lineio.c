#include "lineio.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

struct line_src {
    FILE *f;
    gzFile gz;
    char *buf;
    size_t cap;
};

static int name_is_gz(const char *fn)
{
    size_t n = strlen(fn);
    return n >= 3 && strcmp(fn + n - 3, ".gz") == 0;
}

line_src *ls_open(const char *fn)
{
    if (!fn)
        return NULL;
    line_src *ls = calloc(1, sizeof *ls);
    if (!ls)
        return NULL;
    if (name_is_gz(fn)) {
        ls->gz = gzopen(fn, "r");
        if (ls->gz)
            gzbuffer(ls->gz, 1 << 20);
    } else {
        ls->f = fopen(fn, "r");
        if (!ls->f) {
            fprintf(stderr, "Failed to open %s.\n", fn);
            perror("Error");
        }
    }
    if (!ls->f && !ls->gz) {
        free(ls);
        return NULL;
    }
    ls->cap = 1 << 16;
    ls->buf = malloc(ls->cap);
    return ls;
}

char *ls_next(line_src *ls, size_t *len)
{
    size_t have = 0;
    for (;;) {
        char *got = ls->gz ? gzgets(ls->gz, ls->buf + have, (int)(ls->cap - have))
                           : fgets(ls->buf + have, (int)(ls->cap - have), ls->f);
        if (!got)
            break;
        have += strlen(ls->buf + have);
        if (have && ls->buf[have - 1] == '\n')
            break;
        if (have + 1 < ls->cap)
            break;                      /* EOF without newline */
        ls->cap *= 2;
        ls->buf = realloc(ls->buf, ls->cap);
    }
    if (have == 0)
        return NULL;
    if (len)
        *len = have;
    return ls->buf;
}

int ls_rewind(line_src *ls)
{
    return ls->gz ? gzrewind(ls->gz) : fseek(ls->f, 0, SEEK_SET);
}

void ls_close(line_src *ls)
{
    if (!ls)
        return;
    if (ls->gz)
        gzclose(ls->gz);
    if (ls->f)
        fclose(ls->f);
    free(ls->buf);
    free(ls);
}

const char *ls_map(const char *fn, size_t *size)
{
    if (!fn || name_is_gz(fn))
        return NULL;
    const int fd = open(fn, O_RDONLY);
    if (fd < 0)
        return NULL;
    struct stat st;
    const char *base = NULL;
    if (fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
        void *m = mmap(NULL, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m != MAP_FAILED) {
            base = m;
            *size = (size_t)st.st_size;
        }
    }
    close(fd);
    return base;
}

void ls_unmap(const char *base, size_t size)
{
    if (base)
        munmap((void *)base, size);
}

void ls_split_lines(const char *base, size_t size, size_t from, int parts, size_t *cut)
{
    cut[0] = from;
    for (int p = 1; p < parts; ++p) {
        size_t c = from + (size - from) / (size_t)parts * (size_t)p;
        if (c < cut[p - 1])
            c = cut[p - 1];
        if (c > from && c < size && base[c - 1] != '\n') {          /* move to the start of the next line */
            const char *q = memchr(base + c, '\n', size - c);
            c = q ? (size_t)(q - base) + 1 : size;
        }
        cut[p] = c;
    }
    cut[parts] = size;
}

size_t ls_mt_min_bytes(void)
{
    const char *e = getenv("IBDGEM_MT_MIN_BYTES");
    return e && *e ? (size_t)strtoull(e, NULL, 10) : (size_t)1 << 20;
}

void *ls_xrealloc(void *p, size_t bytes)
{
    void *q = realloc(p, bytes ? bytes : 1);
    if (!q) {
        static const char msg[] = "[::] ERROR: out of memory while reading the input files.\n";
        if (write(2, msg, sizeof msg - 1) < 0) { /* nothing left to do about it */ }
        _exit(1);
    }
    return q;
}
