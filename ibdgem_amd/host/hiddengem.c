/* hiddengem.c -- most probable path of IBD0/IBD1/IBD2 states over the windows of a summary file.
 *
 * Own implementation with the command line, messages and output format of the reference's second
 * program (reference src/hiddengem.c: options :17-24/:192-237, summary reader :51-85, recurrence
 * :103-147, traceback and output :246-283).  Downstream of the accelerated path: it consumes the
 * *.summary.txt written by `ibdgem`; a three-state max-product recurrence over at most a few
 * ten thousand windows is host work.
 *
 * Arithmetic follows the reference so that the printed scores are the same text: per window the
 * three likelihoods are normalised in double (l / ((l0+l1)+l2)); scores are products kept in
 * long double; a transition multiplies (previous score * window probability) * penalty, in that
 * order; ties keep the lowest state (strict >).  Differences, on purpose: any number of windows
 * (the reference holds 12288 in fixed arrays, src/hiddengem.c:8,27-33), an empty table prints the
 * header and the three percentage lines instead of reading out of bounds, and a missing -s is an
 * error message instead of an uninitialised file name. */
#include <ctype.h>
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lineio.h"

static double pen01 = 0.001, pen02 = 0.000001, pen12 = 0.001;

static struct option longopts[] = {{"summary", required_argument, 0, 's'}, {"p01", required_argument, 0, 1},
                                   {"p02", required_argument, 0, 2},       {"p12", required_argument, 0, 3},
                                   {"help", no_argument, 0, 'h'},          {0, 0, 0, 0}};

static void usage(int code)
{
    fputs("HIDDENGEM: Finds most probable path of IBD states across genomic segments.\n\n"
          "Usage: ./hiddengem -s [summary-file] [other options...] >[out-file]\n"
          "--summary, -s  FILE      Summary file from IBDGem likelihood calculation (*.summary.txt) (required)\n"
          "--p01  FLOAT             Penalty for switching between states IBD0 and IBD1 (default: 1e-3)\n"
          "--p02  FLOAT             Penalty for switching between states IBD0 and IBD2 (default: 1e-6)\n"
          "--p12  FLOAT             Penalty for switching between states IBD1 and IBD2 (default: 1e-3)\n"
          "--help                   Show this help message and exit\n\n"
          "Format of output table is tab-delimited with columns:\n"
          "Segment, IBD0_Score, IBD1_Score, IBD2_Score, Inferred_State\n",
          stderr);
    exit(code);
}

static int argmax3(const long double v[3])
{
    int m = 0;
    for (int i = 0; i < 3; ++i)
        if (v[i] > v[m])
            m = i;
    return m;
}

int main(int argc, char **argv)
{
    const char *summary_fn = NULL;
    if (argc == 1)
        usage(0);
    int o;
    while ((o = getopt_long(argc, argv, ":s:h", longopts, NULL)) != -1) {
        switch (o) {
        case 's': summary_fn = optarg; break;
        case 1: pen01 = atof(optarg); break;
        case 2: pen02 = atof(optarg); break;
        case 3: pen12 = atof(optarg); break;
        case 'h': usage(0); break;
        case ':': fprintf(stderr, "Option -%c missing required argument.\n", optopt); exit(0);
        case '?':
            if (isprint(optopt))
                fprintf(stderr, "Invalid option -%c.\n", optopt);
            else
                fprintf(stderr, "Invalid option character.\n");
            break;
        default: fprintf(stderr, "[::] ERROR parsing command-line options.\n"); exit(0);
        }
    }
    for (int i = optind; i < argc; ++i)
        fprintf(stderr, "Given extra argument %s.\n", argv[i]);
    if (!summary_fn) {
        fprintf(stderr, "[::] ERROR parsing likelihood data; make sure input is valid.\n");
        return 1;
    }
    line_src *ls = ls_open(summary_fn);
    if (!ls)
        return 1;

    /* rows: SEGMENT START END LIBD0 LIBD1 LIBD2 NUM_SITES; leading '#' lines are the header, later
     * lines that do not parse are passed over (src/hiddengem.c:62-80) */
    size_t n = 0, cap = 0;
    double (*p)[3] = NULL;
    int in_header = 1;
    for (char *line; (line = ls_next(ls, NULL));) {
        if (in_header && line[0] == '#')
            continue;
        in_header = 0;
        size_t start, end;
        double l0, l1, l2;
        int nsites;
        if (sscanf(line, "%*s\t%zu\t%zu\t%lf\t%lf\t%lf\t%d", &start, &end, &l0, &l1, &l2, &nsites) != 6)
            continue;
        if (n == cap) {
            cap = cap ? cap * 2 : 4096;
            p = realloc(p, cap * sizeof *p);
            if (!p)
                return 1;
        }
        p[n][0] = l0 / (l0 + l1 + l2);
        p[n][1] = l1 / (l0 + l1 + l2);
        p[n][2] = l2 / (l0 + l1 + l2);
        n++;
    }
    ls_close(ls);

    /* score[i][s] = best product of probabilities and switch penalties over paths ending in state
     * s at window i; from[i][s] = the state at i-1 on that path */
    long double (*score)[3] = malloc((n ? n : 1) * sizeof *score);
    unsigned char (*from)[3] = malloc((n ? n : 1) * sizeof *from);
    int *path = malloc((n ? n : 1) * sizeof *path);
    if (!score || !from || !path)
        return 1;
    const double pen[3][3] = {{1, pen01, pen02}, {pen01, 1, pen12}, {pen02, pen12, 1}};
    for (size_t i = 0; i < n; ++i) {
        for (int s = 0; s < 3; ++s) {
            if (i == 0) {
                score[0][s] = p[0][s];
                from[0][s] = (unsigned char)s;
                continue;
            }
            long double cand[3];
            for (int q = 0; q < 3; ++q) {
                cand[q] = score[i - 1][q] * p[i][s];
                if (q != s)
                    cand[q] = cand[q] * pen[q][s];
            }
            const int m = argmax3(cand);
            score[i][s] = cand[m];
            from[i][s] = (unsigned char)m;
        }
    }
    if (n) {
        path[n - 1] = argmax3(score[n - 1]);
        for (size_t i = n - 1; i > 0; --i)
            path[i - 1] = from[i][path[i]];
    }

    printf("Segment\tIBD0_Score\tIBD1_Score\tIBD2_Score\tInferred_State\n");
    double count[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i) {
        count[path[i]]++;
        printf("%d\t%.5Le\t%.5Le\t%.5Le\t%d\n", (int)(i + 1), score[i][0], score[i][1], score[i][2], path[i]);
    }
    for (int s = 0; s < 3; ++s)
        printf("#%% IBD%d (n = %.0f): %.2f\n", s, count[s], (count[s] / (int)n) * 100);
    free(p);
    free(score);
    free(from);
    free(path);
    return 0;
}
