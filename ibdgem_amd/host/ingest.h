/* ingest.h -- block-wise, multi-threaded IMPUTE .hap reader and the packed-panel cache file.
 * Replaces, for the whole file at once, what the reference does per row and per comparison
 * individual: get_line_FS on the .hap file (src/ibdgem.c:573-574, src/file-io.c:20-27) and the
 * character walk of find_f_impute (src/ibd-parse.c:91-99). */
#ifndef IBDG_INGEST_H
#define IBDG_INGEST_H
#include <stddef.h>
#include <stdint.h>

/* Read the rows of `fn` (plain text: mapped, every thread packs a byte range of its own; gzip if
 * the name ends in ".gz": streamed in blocks whose rows the team packs) into the
 * packed layout of include/ibdgem_hip.h.  ok[r] = 1 when row r has >= 4*n_ids-1 characters and
 * only '0'/'1' at its allele offsets (ibdg_pack_hap_text returning 0); other rows are all zero.
 * The arrays are malloc'ed; returns 0, or 1 on I/O / memory failure. */
int ingest_hap(const char *fn, unsigned n_ids, int threads, uint64_t **packed, uint8_t **ok, size_t *n_rows);

/* Alternate-allele count of every packed row (the number of set bits: find_f_impute / find_f_vcf,
 * reference src/ibd-parse.c:91-110), by a team of threads. */
void ingest_alt_counts(const uint64_t *packed, size_t n_rows, unsigned n_ids, int threads, uint32_t *out);

/* Cache file = header (panel width, row count, size and mtime of the .hap file it was made from),
 * ok flags, alt-allele counts, packed rows.  load returns 1 when the file is missing or was made from
 * another input.  *packed of a successful load is a read-only MAPPING of the file that lives as long as
 * the program: the caller must not free() it, and a file truncated or rewritten in place by somebody
 * else during the run raises SIGBUS in whoever reads the rows (ingest_cache_store itself replaces the
 * file by rename, which leaves an existing mapping intact).  ok and alt are malloc'ed. */
int ingest_cache_load(const char *cache_fn, const char *hap_fn, unsigned n_ids, uint64_t **packed, uint8_t **ok,
                      uint32_t **alt, size_t *n_rows);
/* The same without the mapping: the file stays open (*fd) and the packed rows start at byte *rows_off of it -- for a
 * caller that hands the rows to the engine as a file (ibdg_upload_panel_fd) and maps them only if something on the host
 * asks for a row (ingest_cache_map).  The same caveat about a file rewritten in place applies to whoever reads it. */
int ingest_cache_open(const char *cache_fn, const char *hap_fn, unsigned n_ids, int *fd, uint64_t *rows_off, uint8_t **ok,
                      uint32_t **alt, size_t *n_rows);
/* read-only mapping of `bytes` bytes of the open cache file from rows_off on (NULL on failure) */
uint64_t *ingest_cache_map(int fd, uint64_t rows_off, size_t bytes);
int ingest_cache_store(const char *cache_fn, const char *hap_fn, unsigned n_ids, const uint64_t *packed,
                       const uint8_t *ok, const uint32_t *alt, size_t n_rows);
#endif
