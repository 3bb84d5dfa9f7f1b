/* write_floor.c -- how fast can this machine put N bytes that are already text in memory into one file?
 * The floor under any way of producing the per-site table (329.8 MB for 4M rows): T threads pwrite() disjoint
 * ranges of one file, the way ibdgem_amd/host/ibdgem.c's write_rows_parallel does after formatting.
 *   gcc -O2 -pthread tools/write_floor.c -o /tmp/write_floor && /tmp/write_floor 16 /dev/shm/x [bytes]
 * THREADS < 0: -THREADS writers, each with a file of its own (FILE.0, FILE.1, ...) of BYTES each -- do several tables
 * written at the same time share the one-writer rate or each get it?
 * Not part of the product; evidence for DESIGN.md s9 (device-side formatting). */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static char *buf;
static size_t n_bytes = 329779046;
static int fd, n_thr;

static char path_base[4000];
static void *file_writer(void *arg)
{
    long i = (long)arg;
    char fn[4100];
    snprintf(fn, sizeof fn, "%s.%ld", path_base, i);
    int f = open(fn, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (f < 0) { perror(fn); _exit(1); }
    for (size_t lo = 0; lo < n_bytes;) {
        size_t c = n_bytes - lo;
        if (c > ((size_t)4 << 20)) c = (size_t)4 << 20;
        ssize_t r = write(f, buf + lo, c);
        if (r <= 0) { perror("write"); _exit(1); }
        lo += (size_t)r;
    }
    close(f);
    return NULL;
}

static void *writer(void *arg)
{
    long i = (long)arg;
    size_t lo = n_bytes / n_thr * i, hi = (i == n_thr - 1) ? n_bytes : n_bytes / n_thr * (i + 1);
    while (lo < hi) {
        size_t c = hi - lo;
        if (c > ((size_t)4 << 20)) c = (size_t)4 << 20;
        ssize_t r = pwrite(fd, buf + lo, c, (off_t)lo);
        if (r <= 0) { perror("pwrite"); _exit(1); }
        lo += (size_t)r;
    }
    return NULL;
}

static double now(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec + t.tv_nsec * 1e-9;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: write_floor THREADS FILE [BYTES]\n"); return 2; }
    n_thr = atoi(argv[1]);
    const int own_files = n_thr < 0;
    if (own_files) n_thr = -n_thr;
    if (n_thr < 1 || n_thr > 64) return 2;
    snprintf(path_base, sizeof path_base, "%s", argv[2]);
    if (argc > 3) n_bytes = strtoull(argv[3], NULL, 10);
    buf = malloc(n_bytes);
    if (!buf) return 1;
    memset(buf, 'x', n_bytes);
    for (int rep = 0; own_files && rep < 3; rep++) {
        double t0 = now();
        pthread_t th[64];
        for (long i = 0; i < n_thr; i++) pthread_create(&th[i], NULL, file_writer, (void *)i);
        for (int i = 0; i < n_thr; i++) pthread_join(th[i], NULL);
        double t1 = now();
        printf("%d writers, a file of %zu bytes each -> %s.N: %.3f s  %.2f GB/s in all, %.2f per file%s\n", n_thr, n_bytes,
               argv[2], t1 - t0, n_thr * (double)n_bytes / (t1 - t0) / 1e9, n_bytes / (t1 - t0) / 1e9, rep ? "" : "  (first: new pages)");
    }
    for (long i = 0; own_files && i < n_thr; i++) {
        char fn[4100];
        snprintf(fn, sizeof fn, "%s.%ld", path_base, i);
        unlink(fn);
    }
    if (own_files)
        return 0;
    for (int rep = 0; rep < 4; rep++) {
        fd = open(argv[2], O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) { perror(argv[2]); return 1; }
        double t0 = now();
        pthread_t th[64];
        for (long i = 0; i < n_thr; i++) pthread_create(&th[i], NULL, writer, (void *)i);
        for (int i = 0; i < n_thr; i++) pthread_join(th[i], NULL);
        double t1 = now();
        close(fd);
        printf("%d threads, %zu bytes -> %s: %.3f s  %.2f GB/s%s\n", n_thr, n_bytes, argv[2], t1 - t0,
               n_bytes / (t1 - t0) / 1e9, rep ? "" : "  (first: new pages)");
    }
    unlink(argv[2]);
    return 0;
}
