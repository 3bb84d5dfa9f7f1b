/* pileup.h -- samtools pileup -> compact per-position base counts.
 *
 * Restates what ibdgem needs from the reference's pileup model
 * (src/pileup.c:206-415 line2pul, :442-450 count_base_from_pul, :472-485
 * fetch_Pul, :487-559 init_Pu_chr) in 16 bytes per line instead of the
 * reference's 1944-byte Pul: ibdgem only ever asks "how many of the cov base
 * characters equal this letter" for the letters A,C,G,T
 * (src/ibdgem.c:620-621, is_snp :113-119). */
#ifndef IBDG_PILEUP_H
#define IBDG_PILEUP_H
#include <stddef.h>
#include <stdint.h>

typedef struct {
    uint32_t pos;       /* column 2 (unsigned int, like Pul.pos) */
    uint32_t chr;       /* index into pileup_t.chr_names */
    uint8_t cov;        /* column 4, < 128 */
    uint8_t n[4];       /* bases equal to 'A','C','G','T' after '.'/',' -> ref char */
    uint8_t pad[3];
} pu_line;

typedef struct {
    pu_line *lines;
    size_t n_lines;
    char **chr_names;
    size_t n_chr;
} pileup_t;

/* Whole file -> table; only lines whose chromosome equals `chr` when chr != NULL.
 * NULL on error (message on stderr), including unsorted input. */
pileup_t *pileup_read(const char *fn, const char *chr);
/* the same with the lines of a plain file parsed by `threads` threads (byte ranges cut at line starts,
 * tables and messages joined in file order: same table, same stderr text as pileup_read) */
pileup_t *pileup_read_mt(const char *fn, const char *chr, int threads);
/* line with exactly this position, or NULL (binary search, like fetch_Pul) */
const pu_line *pileup_find(const pileup_t *pu, unsigned long pos);
/* count_base_from_pul: bytes equal to `base`; 0 for anything but A,C,G,T */
unsigned pileup_count(const pu_line *l, char base);
void pileup_free(pileup_t *pu);

/* Parse one pileup line into *out / chr_buf (size >= 256).  0 = keep the line,
 * 1 = drop it silently or with the reference's message, 2 = unparsable start
 * (src/pileup.c:206-415). */
int pileup_parse_line(const char *line, pu_line *out, char *chr_buf);
/* the same with the reference's messages written to `err` instead of stderr */
#include <stdio.h>
int pileup_parse_line_to(const char *line, pu_line *out, char *chr_buf, FILE *err);
#endif
