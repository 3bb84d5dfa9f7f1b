/* lineio.h -- line-at-a-time reader for plain or gzip text.
 * Restates the role of the reference's File_Src (src/file-io.h:17-47): the
 * file is treated as gzip when its name ends in ".gz" (src/file-io.c:9-18).
 * Unlike the reference there is no 30720-byte line limit (src/file-io.h:10):
 * the buffer grows, so panels of any width parse. */
#ifndef IBDG_LINEIO_H
#define IBDG_LINEIO_H
#include <stddef.h>

typedef struct line_src line_src;

line_src *ls_open(const char *fn);          /* NULL (and a message on stderr) on failure */
/* next line including its '\n' (if any), NUL-terminated; NULL at EOF.  The
 * pointer is valid until the next call. */
char *ls_next(line_src *ls, size_t *len);
int ls_rewind(line_src *ls);
void ls_close(line_src *ls);

/* Plain (not .gz) files for readers that split the text among threads: the whole file mapped
 * read-only, or NULL (gzip name, empty file, any failure -- the caller then reads line by line). */
const char *ls_map(const char *fn, size_t *size);
void ls_unmap(const char *base, size_t size);
/* cut[0..parts]: byte offsets from `from` to `size` that all sit at the start of a line (cut[0] = from,
 * cut[parts] = size), about equal in bytes. */
void ls_split_lines(const char *base, size_t size, size_t from, int parts, size_t *cut);
/* files smaller than this are read line by line (1 MiB; IBDGEM_MT_MIN_BYTES in the environment overrides it,
 * which is how the tests send their small files through the threaded readers) */
size_t ls_mt_min_bytes(void);

/* realloc / malloc for the readers' growing tables, which run on worker threads with nobody to return an error to:
 * on failure a message and _exit(1) -- not exit(): other threads may be inside the GPU runtime, whose exit handlers
 * must not run under them. */
void *ls_xrealloc(void *p, size_t bytes);
#define ls_xmalloc(bytes) ls_xrealloc(NULL, (bytes))
#endif
