/*
 * oracle/ibd_oracle.c -- CPU restatement of the IBDGem hot path (see
 * ibd_oracle.h for status and usage rules: TEST INFRASTRUCTURE ONLY).
 *
 * Written from the behaviour of the reference, keeping its operation order so
 * that doubles come out bit-identical on x86-64/glibc:
 *   - every product/sum is evaluated left to right exactly as the reference's
 *     expressions associate (no FMA: build with -ffp-contract=off);
 *   - pow() is the real libm pow (build with -fno-builtin so the compiler does
 *     not turn pow(x,2.0) into x*x, which differs in the last bit for ~0.1% of
 *     inputs under glibc 2.35).
 */
#include "ibd_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* src/ibd-math.c:5-23: C(i,j) by the recurrence (i*C(i-1,j-1))/j in unsigned
 * long, C(i,0)=1, and 0 whenever j>i (the recursion hits a factor n==0). */
unsigned long *orc_nck_table(unsigned n)
{
    size_t dim = (size_t)n + 1;
    unsigned long *t = malloc(dim * dim * sizeof *t);
    if (!t) return NULL;
    for (size_t i = 0; i < dim; i++) {
        for (size_t j = 0; j < dim; j++) {
            unsigned long v;
            if (j == 0)      v = 1;
            else if (i == 0) v = 0;
            else             v = ((unsigned long)(unsigned)i * t[(i - 1) * dim + (j - 1)]) / (unsigned)j;
            t[i * dim + j] = v;
        }
    }
    return t;
}

/* src/ibd-math.c:46-81 */
double orc_pDgG(const unsigned long *nck, unsigned max_cov, double eps,
                unsigned g, unsigned n_ref, unsigned n_alt)
{
    if (n_ref == 0 && n_alt == 0)
        return 1.0;                                   /* :51-53 */
    size_t dim = (size_t)max_cov + 1;
    unsigned long c = nck[(size_t)(n_ref + n_alt) * dim + n_ref];   /* :55 */
    double p;
    if (g == 0)
        p = c * pow(1 - eps, n_ref) * pow(eps, n_alt);              /* :58 */
    else if (g == 2)
        p = c * pow(1 - eps, n_alt) * pow(eps, n_ref);              /* :61 */
    else
        p = c * pow(0.5, n_ref) * pow(0.5, n_alt);                  /* :69 */
    if (p == 0.0)
        p = DBL_MIN;                                                /* :77-79 */
    return p;
}

/* src/ibd-math.c:84-101 */
double orc_pDgf(double f, double p00, double p01, double p11)
{
    if (p00 == 1 || p01 == 1 || p11 == 1)
        return 1.0;                                                 /* :88-90 */
    double p = pow(1 - f, 2.0) * p00 + 2 * (1 - f) * f * p01 + pow(f, 2.0) * p11;
    if (p == 0.0)
        p = DBL_MIN;
    return p;
}

/* src/ibd-math.c:104-142 */
double orc_pDgIBD1(unsigned A0, unsigned A1, double f,
                   double p00, double p01, double p11)
{
    double p = 1.0;
    unsigned g = A0 + A1;
    if (A0 > 1 || A1 > 1)
        return p;                       /* no branch taken in the reference */
    if (g == 0)
        p = (f * p01) + ((1 - f) * p00);                            /* :119 */
    else if (g == 1)
        p = (0.5 * p01) + (0.5 * (1 - f) * p00) + (0.5 * f * p11);  /* :126-128 */
    else
        p = ((1 - f) * p01) + (f * p11);                            /* :135 */
    if (p == 0.0)
        p = DBL_MIN;
    return p;
}

/* src/ibd-parse.c:91-99: count of '1' alleles over all individuals, as a
 * double, divided by the int 2*n_ids. */
double orc_alt_fraction(const uint8_t *row, unsigned n_ids)
{
    double n_alt = 0;
    for (unsigned i = 0; i < 2 * n_ids; i++)
        if (row[i] == 1)
            n_alt++;
    return n_alt / (int)(n_ids * 2);
}

/* genotype-class pick used all over src/ibdgem.c:643-651 and :678-711 */
static inline double pick(unsigned x, unsigned y, double p00, double p01, double p11)
{
    unsigned g = x + y;
    return g == 0 ? p00 : (g == 1 ? p01 : p11);
}

size_t orc_compare(const orc_input *in, unsigned target,
                   const int32_t *refids, size_t n_refids, int pu_id,
                   int ld_mode,
                   double *site_af, double *site_ll,
                   double *win_ll, uint32_t *win_first, uint32_t *win_last,
                   uint32_t *win_nsites)
{
    const unsigned N = in->n_ids;
    const size_t row_len = 2 * (size_t)N;
    unsigned long *nck = orc_nck_table(in->max_cov);           /* ibdgem.c:1168 */

    int32_t *all = NULL;
    if (!refids) {                                             /* ibdgem.c:517-520 */
        all = malloc((N ? N : 1) * sizeof *all);
        for (unsigned n = 0; n < N; n++) all[n] = (int32_t)n;
        refids = all;
        n_refids = N;
    }
    double *bg2 = malloc((n_refids ? n_refids : 1) * sizeof *bg2);        /* sum_ibd2_ref */
    double *bg1 = malloc((n_refids ? n_refids : 1) * 4 * sizeof *bg1);    /* sum_ibd1_ref */

    size_t n_win = 0;
    size_t s = 0;
    int more = 1;
    while (more) {                                             /* ibdgem.c:558 */
        unsigned snp_count = 0;
        uint32_t first = 0, last = 0;
        double S0 = 1, S1 = 1, S2 = 1;                         /* :562 */
        for (size_t n = 0; n < n_refids; n++) {                /* :564-570 */
            bg2[n] = 1;
            bg1[4 * n] = bg1[4 * n + 1] = bg1[4 * n + 2] = bg1[4 * n + 3] = 1;
        }
        while (snp_count < in->window) {                       /* :572 */
            if (s >= in->n_sites) { more = 0; break; }         /* :575-578 */
            const uint8_t *row = in->alleles + s * row_len;
            unsigned r = in->n_ref[s], a = in->n_alt[s];
            double f = orc_alt_fraction(row, N);               /* :608 */
            if (in->f_override && !isnan(in->f_override[s]))
                f = in->f_override[s];                         /* :609-614 */
            double p00 = orc_pDgG(nck, in->max_cov, in->eps, 0, r, a);   /* :632-634 */
            double p01 = orc_pDgG(nck, in->max_cov, in->eps, 1, r, a);
            double p11 = orc_pDgG(nck, in->max_cov, in->eps, 2, r, a);
            unsigned A0 = row[2 * target], A1 = row[2 * target + 1];     /* :638-639 */
            double ibd0 = orc_pDgf(f, p00, p01, p11);          /* :641 */
            double ibd1 = orc_pDgIBD1(A0, A1, f, p00, p01, p11);/* :642 */
            double ibd2 = pick(A0, A1, p00, p01, p11);         /* :643-651 */
            site_af[s] = f;
            site_ll[3 * s] = ibd0;
            site_ll[3 * s + 1] = ibd1;
            site_ll[3 * s + 2] = ibd2;
            size_t cur = s++;
            if (r + a < 1)                                     /* :657-663 */
                continue;
            S0 *= ibd0;                                        /* :665-667 */
            S1 *= ibd1;
            S2 *= ibd2;
            if (ld_mode) {                                     /* :669-722 */
                for (size_t n = 0; n < n_refids; n++) {
                    int k = refids[n];
                    unsigned h0 = row[2 * k], h1 = row[2 * k + 1];
                    if (k != pu_id && k != (int)target) {      /* :714 */
                        bg2[n] *= pick(h0, h1, p00, p01, p11);
                        bg1[4 * n] *= pick(A0, h0, p00, p01, p11);
                        bg1[4 * n + 1] *= pick(A0, h1, p00, p01, p11);
                        bg1[4 * n + 2] *= pick(A1, h0, p00, p01, p11);
                        bg1[4 * n + 3] *= pick(A1, h1, p00, p01, p11);
                    }
                }
            }
            snp_count++;                                       /* :723-730 */
            if (snp_count == 1) first = (uint32_t)cur;
            last = (uint32_t)cur;
        }
        if (snp_count > 0) {                                   /* :736-759 */
            double l0 = S0, l1 = S1;
            if (ld_mode) {
                int n_refpanel = (int)n_refids;
                double t0 = 0, t1 = 0;
                for (size_t n = 0; n < n_refids; n++) {
                    int k = refids[n];
                    if (k != pu_id && k != (int)target) {
                        t0 += bg2[n];
                        t1 += (bg1[4 * n] + bg1[4 * n + 1] + bg1[4 * n + 2] + bg1[4 * n + 3]);
                    } else {
                        n_refpanel--;
                    }
                }
                l0 = t0 / n_refpanel;
                l1 = t1 / (n_refpanel * 4);
            }
            win_ll[3 * n_win] = l0;
            win_ll[3 * n_win + 1] = l1;
            win_ll[3 * n_win + 2] = S2;
            win_first[n_win] = first;
            win_last[n_win] = last;
            win_nsites[n_win] = snp_count;
            n_win++;
        }
    }
    free(bg1);
    free(bg2);
    free(all);
    free(nck);
    return n_win;
}
