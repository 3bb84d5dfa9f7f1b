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
#endif
