// ibdg_kernels.h -- kernel argument blocks and launch wrappers (internal).
#pragma once
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#include <stddef.h>
#include <stdint.h>

namespace ibdg {

struct SiteArgs {
    const uint64_t *panel;      // [n_rows][stride]
    uint32_t stride;            // u64 words per device row
    uint32_t n_ids;
    const uint2 *rec_all;       // [n_sites] {row, lut byte offset}
    size_t n_sites;
    const double *lut;          // [(M+1)^2][3]
    const uint32_t *alt_count;  // [n_rows]
    const double *pow_tab;      // [2*n_ids+1][2] = pow(1-f,2), pow(f,2) at f=k/(2N)
    const double *fo;           // NULL or [n_sites][3] = f, pow(1-f,2), pow(f,2) (-A)
    const uint32_t *targets;    // [T]
    double *af;                 // [n_sites]
    double *site_ll;            // [T][n_sites][3]
};

struct WinArgs {
    const double *site_ll;      // [T][n_sites][3]
    size_t n_sites;
    const uint32_t *cov_site;   // [n_cov] site index of each covered row
    uint32_t n_cov;
    uint32_t window;
    uint32_t n_win;
    int ld_mode;
    double *win_ll;             // [T][n_win][3]
};

struct LdArgs {
    const uint64_t *panel;
    uint32_t stride;
    const uint2 *rec_cov;       // [n_cov] {row, lut byte offset} of covered rows
    uint32_t n_cov;
    uint32_t window;
    uint32_t n_win;
    uint32_t n_groups;          // chunk groups per row = stride / (2*CPW)
    const double *lut;
    const uint32_t *targets;    // [T]
    const double *weight;       // [T][n_groups*CPW*64] background multiplicity (0 = excluded)
    const int *n_refpanel;      // [T]
    double *win_ll;             // [T][n_win][3]
};

void launch_alt_count(const uint64_t *panel, uint32_t stride, size_t n_rows, uint32_t *alt_count,
                      hipStream_t st);
void launch_site(const SiteArgs &a, unsigned n_targets, hipStream_t st);
void launch_window_prod(const WinArgs &a, unsigned n_targets, hipStream_t st);
int launch_ld(const LdArgs &a, unsigned n_targets, int cpw, unsigned waves, hipStream_t st);

}  // namespace ibdg
