#!/usr/bin/env python3
"""bench.py -- SNP-sites/s of the --LD hot path on synthetic chr1-scale data.

Workload (BASELINE.json configs[3], the config the metric is quoted on):
  --LD, ~4M SNP rows, 2504-individual phased panel, window 100, 1 comparison
  individual, epsilon 0.02, max-cov 20; Poisson(2) read depth so ~13.5% of the
  rows have no informative read (printed, not windowed -- src/ibdgem.c:657-663).
A "step" = one full pass over all rows: the --LD background loop + window averages
(k_target_weights, k_win_target_mx, k_ld_popcount, the finalising step) and, beside them on a
second stream, per-site LIBD0/1/2 and the window products (k_rows_windows); inputs resident in HBM (the packed
panel in its two layouts and the per-row alt-allele counts, all produced once
by the panel upload -- they depend on the panel only, like the reference's -A
file), results left in HBM.  value = windowed sites processed by all ranks /
max-over-ranks wall time per step.  The JSON also carries the cost of
recomputing the alt counts inside the step ("alt_count_ms", "value_with_recount").

With --gpus N the SAME chromosome is cut into N contiguous window ranges, one
per rank (strong scaling; no data-path collective -- windows are independent,
src/ibdgem.c:558-570; torch.distributed is used only for the barrier and the
max-over-ranks reduction of the timing).

Beside `value` (inputs resident in HBM, results left there) the line carries the other clocks of
SURVEY.md s8(d), each labelled: "upload_sites" (what one more comparison costs before its first run:
copying its row list and read counts and preparing windows / segments on the device), "engine_clock"
(site arrays resident -> per-window results in host memory: preparation, alt-allele counts, all
kernels, the copy back), "results_to_host" (per-site results), "warm_e2e" / "cold_e2e" (the host
program ibdgem_amd/host/ibdgem from files to files), and "cpu_baseline" (the unmodified reference
binary on a slice of the same data).

The timed steps take two comparison individuals in turn (a NEW individual per step, as in the reference's loop over
the individuals of the panel, src/ibdgem.c:522): nothing a step needs is left over from the step before it -- the
individual's window / segment images (k_win_target_mx), its background weights and the --LD kernel all run in every
step; a step's finalising arithmetic (k_ld_finalize's, 5 us) rides in the next step's --LD launch and the last step's
in a launch of its own before the closing barrier.  `--same-target` times the round-4 form (one comparison run again).

Rank 0 prints ONE compact JSON line (<= 4096 bytes: the contract's keys, `roofline`, `cpu_baseline`, per-rank numbers);
everything else -- the other clocks, notes, the host-program legs -- goes to the file named in its `detail_file`.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BLOCK_ROWS = 1 << 16           # generation granule; data depend on the global row only


def algorithmic_bytes_per_site(n_ids, n_targets):
    """packed panel row N/4 + n_ref,n_alt 2 B + alt count 2 B, and per
    target 24 B of per-site output + 24 B/window."""
    return n_ids / 4.0 + 4.0 + n_targets * 24.24


# ----------------------------------------------------------------------------- synthetic data (GPU, torch = plumbing)
def _block_generators(torch, dev, block, seed):
    """Two streams per block: `g` for the panel's bits, `g2` for what the read counts need (the allele
    frequencies, the comparison individual's own two bits per row, depth, reads).  Keeping them apart lets a
    rank learn the read counts of ALL rows (for the window cut points) without generating the panel."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed * 1000003 + block)
    g2 = torch.Generator(device=dev)
    g2.manual_seed((seed * 1000003 + block) ^ 0x5BD1E995)
    return g, g2


def gen_counts(torch, dev, block, seed):
    """Rows [block*BLOCK_ROWS, +BLOCK_ROWS): allele frequency, the comparison individual's two alleles,
    n_ref, n_alt -- a few bytes per row, no panel."""
    _, g2 = _block_generators(torch, dev, block, seed)
    R = BLOCK_ROWS
    u = torch.rand(R, generator=g2, device=dev, dtype=torch.float64)
    f = torch.clamp(u ** (1.0 / 0.3), 1e-3, 0.999).to(torch.float32)      # Beta(0.3, 1)
    tb = torch.rand(R, 2, generator=g2, device=dev) < f[:, None]
    a0 = tb[:, 0].to(torch.float32)
    a1 = tb[:, 1].to(torch.float32)
    cov = torch.poisson(torch.full((R,), 2.0, device=dev), generator=g2).clamp_(max=20)
    p_alt = (a0 + a1) * 0.5 * (1 - 2 * 0.02) + 0.02          # g=0: eps, g=1: 0.5, g=2: 1-eps
    n_alt = torch.binomial(cov, p_alt, generator=g2)
    n_ref = cov - n_alt
    return f, tb, n_ref.to(torch.uint8), n_alt.to(torch.uint8)


def gen_block(torch, dev, block, n_ids, target, seed):
    """Rows [block*BLOCK_ROWS, +BLOCK_ROWS): packed panel words, n_ref, n_alt.  Deterministic in
    (seed, block), independent of how blocks are spread over ranks."""
    g, _ = _block_generators(torch, dev, block, seed)
    f, tb, n_ref, n_alt = gen_counts(torch, dev, block, seed)
    R = BLOCK_ROWS
    chunks = (n_ids + 63) // 64
    bits = torch.rand(R, chunks * 64, 2, generator=g, device=dev) < f[:, None, None]
    bits[:, n_ids:, :] = False
    bits[:, target, :] = tb                                  # the comparison individual is a panel member
    # [R][chunk][bit][plane] -> word[2*chunk+plane] bit
    b = bits.view(R, chunks, 64, 2).permute(0, 1, 3, 2).to(torch.int64)
    weights = (torch.ones(64, dtype=torch.int64, device=dev) << torch.arange(64, device=dev))
    words = (b * weights).sum(dim=-1).reshape(R, chunks * 2).contiguous()
    return words, n_ref, n_alt


def build_shard(torch, dev, row0, row1, n_ids, target, seed):
    """Panel words (device tensor) and read counts (host arrays) for global rows [row0,row1)."""
    words, nrs, nas = [], [], []
    b0, b1 = row0 // BLOCK_ROWS, (row1 + BLOCK_ROWS - 1) // BLOCK_ROWS
    for blk in range(b0, b1):
        w, nr, na = gen_block(torch, dev, blk, n_ids, target, seed)
        lo = max(row0, blk * BLOCK_ROWS) - blk * BLOCK_ROWS
        hi = min(row1, (blk + 1) * BLOCK_ROWS) - blk * BLOCK_ROWS
        words.append(w[lo:hi])
        nrs.append(nr[lo:hi].cpu().numpy())
        nas.append(na[lo:hi].cpu().numpy())
    panel = torch.cat(words, dim=0).contiguous()
    return panel, np.concatenate(nrs), np.concatenate(nas)


# ----------------------------------------------------------------------------- CPU baseline (reference binary)
def unpack_rows(words, n_ids):
    """packed uint64 [L][2*chunks] -> alleles uint8 [L][2*n_ids] ([2n]=first, [2n+1]=second)."""
    L = words.shape[0]
    chunks = words.shape[1] // 2
    by = words.view(np.uint8).reshape(L, chunks, 2, 8)
    bits = np.unpackbits(by, axis=-1, bitorder="little")            # [L][chunk][plane][64]
    return np.ascontiguousarray(bits.transpose(0, 1, 3, 2).reshape(L, chunks * 64, 2)[:, :n_ids, :]).reshape(L, 2 * n_ids)


def write_reference_inputs(d, alleles, n_ref, n_alt, n_ids):
    L = alleles.shape[0]
    hap = np.full((L, 4 * n_ids), ord(" "), dtype=np.uint8)
    hap[:, 0::2] = alleles + ord("0")
    hap[:, -1] = ord("\n")
    hap.tofile(os.path.join(d, "p.hap"))
    with open(os.path.join(d, "p.legend"), "w") as fh:
        fh.write("ID pos allele0 allele1\n")
        fh.write("".join(f"rs{i} {100 + 10 * i} A C\n" for i in range(L)))
    with open(os.path.join(d, "p.indv"), "w") as fh:
        fh.write("".join(f"ind{n}\n" for n in range(n_ids)))
    with open(os.path.join(d, "p.pileup"), "w") as fh:
        out = []
        for i in range(L):
            r, a = int(n_ref[i]), int(n_alt[i])
            c = r + a
            if c == 0:
                out.append(f"1\t{100 + 10 * i}\tN\t0\t*\t*\t*\n")
            else:
                out.append(f"1\t{100 + 10 * i}\tN\t{c}\t{'A' * r}{'C' * a}\t{'I' * c}\t{'I' * c}\n")
        fh.write("".join(out))


def cpu_baseline(words_host, n_ref, n_alt, n_ids, target, window, gpu_win):
    """Time the reference binary (oracle/_ref/ibdgem, unmodified sources, -O0 as shipped) on the
    first rows of the same workload, 1 thread; also check its summary file against the GPU's
    windows for those rows (7 printed digits)."""
    exe = os.path.join(REPO, "oracle", "_ref", "ibdgem")
    L = words_host.shape[0]
    n_cov = int(((n_ref.astype(np.int32) + n_alt) > 0).sum())
    if not os.path.exists(exe):
        # the reference binary did not travel: time our CPU restatement instead and say so
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import oracle_lib
        so = os.path.join(REPO, "oracle", "liboracle.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "liboracle.so"], check=True,
                           stdout=subprocess.DEVNULL)
        orc = oracle_lib.Oracle(so)
        alle = unpack_rows(words_host, n_ids)
        t0 = time.perf_counter()
        orc.compare(alle, n_ref, n_alt, target, window=window, ld=True)
        dt = time.perf_counter() - t0
        return dict(value=n_cov / dt, unit="sites/s", cores=1, kind="port",
                    sample=f"first {L} rows ({n_cov} windowed) of the workload, LD loop only (no parsing), oracle/liboracle.so")
    alle = unpack_rows(words_host, n_ids)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        write_reference_inputs(d, alle, n_ref, n_alt, n_ids)
        base = [exe, "-H", "p.hap", "-L", "p.legend", "-I", "p.indv", "-P", "p.pileup", "-s", f"ind{target}",
                "-O", d]
        times = {}
        cpu0 = min(os.sched_getaffinity(0))             # one process pinned to one core

        def pin():
            os.sched_setaffinity(0, {cpu0})

        for mode, extra in (("ld", ["--LD"]), ("nonld", [])):
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                subprocess.run(base + extra, cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               preexec_fn=pin)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            times[mode] = best
            if mode == "ld":
                rows = [l.split("\t") for l in open(os.path.join(d, f"UNKWN.ind{target}.summary.txt"))
                        if not l.startswith("#")]
        # the same sources at -O2 (labelled; the north-star ratio uses the build as shipped)
        exe_o2 = os.path.join(REPO, "oracle", "_ref", "ibdgem_O2")
        o2 = o2_ld_stage = None
        if os.path.exists(exe_o2):
            t_o2 = {}
            for mode, extra in (("ld", ["--LD"]), ("nonld", [])):
                t0 = time.perf_counter()
                subprocess.run([exe_o2] + base[1:] + extra, cwd=d, check=True, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, preexec_fn=pin)
                t_o2[mode] = time.perf_counter() - t0
            o2 = n_cov / t_o2["ld"]
            o2_ld_stage = n_cov / max(t_o2["ld"] - t_o2["nonld"], 1e-9)
        parity = None
        if gpu_win is not None:
            ok = len(rows) <= len(gpu_win)
            for w, r in enumerate(rows[:-1]):          # the sample's last window is cut short by the slice
                ok = ok and ["%e" % v for v in gpu_win[w]] == r[3:6]
            parity = bool(ok)
    ld_stage = max(times["ld"] - times["nonld"], 1e-9)
    return dict(value=n_cov / times["ld"], unit="sites/s", cores=1, kind="reference",
                sample=(f"first {L} rows ({n_cov} windowed) of the same workload as reference text inputs, "
                        f"unmodified reference built -O0 as shipped, 1 process pinned to core {cpu0}, best of 3, "
                        f"end-to-end --LD run {times['ld']:.2f}s; non-LD {times['nonld']:.2f}s"),
                ld_stage_only_sites_per_s=n_cov / ld_stage,
                O2_rebuild_sites_per_s=o2,
                O2_rebuild_ld_stage_only_sites_per_s=o2_ld_stage,
                summary_matches_gpu_7digits=parity)



# ----------------------------------------------------------------------------- other clocks
def best_of(n, fn):
    best = None
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        dt = (time.perf_counter() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    return best


def upload_and_engine_clocks(torch, eng, ibdgem_amd, n_ref, n_alt, window, targets, n_cov, eng2=None):
    """What one more comparison costs around its kernels, at the full size of this rank's rows.

    upload_sites: ibdg_upload_sites with the caller's arrays in pageable memory (numpy), in page-locked
    memory (ibdg_host_alloc) and already on the device (ibdg_upload_sites_dev); 6 bytes per row go to the
    device (2 when row_index is NULL = the panel's own rows), everything else is derived there.
    engine_clock: site arrays resident in HBM -> per-window results in host memory =
    preparation + alt-allele counts (k_alt_count, inside the run) + per-site, --LD and window kernels +
    the copy of the window table to page-locked host memory; also with the alt counts left to the panel
    upload ("amortised": they depend on the panel only), and from pageable host arrays."""
    n = len(n_ref)
    idx = np.arange(n, dtype=np.uint32)
    pins = [ibdgem_amd.PinnedArray(n, np.uint32), ibdgem_amd.PinnedArray(n, np.uint8), ibdgem_amd.PinnedArray(n, np.uint8)]
    pins[0].array[:] = idx
    pins[1].array[:] = n_ref
    pins[2].array[:] = n_alt
    d_idx = torch.from_numpy(idx.view(np.int32)).cuda()
    d_nr, d_na = torch.from_numpy(n_ref).cuda(), torch.from_numpy(n_alt).cuda()
    torch.cuda.synchronize()
    ups = {
        "pageable_ms": lambda: eng.upload_sites(idx, n_ref, n_alt, window),
        "pageable_rows_implicit_ms": lambda: eng.upload_sites(None, n_ref, n_alt, window),
        "pinned_ms": lambda: eng.upload_sites(pins[0].array, pins[1].array, pins[2].array, window),
        "device_resident_ms": lambda: eng.upload_sites_dev(d_idx.data_ptr(), d_nr.data_ptr(), d_na.data_ptr(), n, window),
    }
    up = {}
    for name, fn in ups.items():
        fn()
        up[name] = best_of(5, fn)
        c = eng.upload_ms()
        up[name.replace("_ms", "_parts")] = {"h2d_ms": c["h2d"], "device_prep_ms": c["device_prep"]}
    up["bytes_to_device_per_row"] = 6
    up["note"] = ("host wall clock of one ibdg_upload_sites[_dev] call, best of 5, all rows of the workload; parts are the "
                  "engine's own event clocks of the last call")
    n_win = eng.n_windows
    win_pin = ibdgem_amd.PinnedArray((n_win, 3), np.float64)

    def clock(recount, dev):
        eng.set_option("count_in_run", 1 if recount else 0)

        def once():
            if dev:
                eng.upload_sites_dev(d_idx.data_ptr(), d_nr.data_ptr(), d_na.data_ptr(), n, window)
            else:
                eng.upload_sites(idx, n_ref, n_alt, window)
            eng.run(targets, ld=True)
            eng.window_ll(0, out=win_pin.array)
        # (option "async": ibdg_run returns once its kernels are queued and ibdg_get_window_ll is what waits -- one host
        # wait for the comparison instead of two)
        eng.set_option("async", 1)
        for _ in range(10):
            once()
        all_ms = []
        for _ in range(20):
            t0 = time.perf_counter()
            once()
            all_ms.append((time.perf_counter() - t0) * 1e3)
        eng.set_option("async", 0)
        eng.set_option("count_in_run", 0)
        medians.append(float(np.median(all_ms)))
        return min(all_ms)
    medians = []
    full = clock(True, True)
    ec = {
        "definition": "site arrays (row index, n_ref, n_alt) resident in HBM -> per-window LIBD0/1/2 in host memory: "
                      "ibdg_upload_sites_dev + ibdg_run with the alt-allele counts recomputed inside (K0) + "
                      "ibdg_get_window_ll into page-locked memory; host wall clock of one comparison, best of 20 after 10 untimed "
                      "ones (the chip's clock takes that long to settle after the other legs: 1.05 ms for the first few, "
                      "0.97-0.99 from then on, profiles/r03_engine_clock_ab.txt)",
        "ms": full, "sites_per_s": n_cov / (full * 1e-3),
        "alt_counts_amortised_ms": None, "alt_counts_amortised_sites_per_s": None,
        "from_pageable_host_arrays_ms": None,
    }
    am = clock(False, True)
    ec["alt_counts_amortised_ms"] = am
    ec["alt_counts_amortised_sites_per_s"] = n_cov / (am * 1e-3)
    ec["from_pageable_host_arrays_ms"] = clock(True, False)
    ec["median_of_20_ms"] = {"ms": medians[0], "alt_counts_amortised_ms": medians[1], "from_pageable_host_arrays_ms": medians[2]}
    if eng2 is not None:
        # comparisons in a stream: two contexts on this GPU (each with the panel) take turns, so the preparation of
        # comparison i+1 is on the device under the --LD kernel of comparison i and the host waits once per comparison
        # (for the window table of the one before).  Same calls as above, alt counts amortised; per comparison.
        win_pin2 = ibdgem_amd.PinnedArray((n_win, 3), np.float64)
        clock(False, True)
        single = win_pin.array.copy()
        engs, outs = [eng, eng2], [win_pin, win_pin2]
        for e in engs:
            e.set_option("async", 1)
            e.set_option("dev_inputs_ready", 1)     # the arrays were complete long ago: no device-wide wait per upload

        def submit(e):
            e.upload_sites_dev(d_idx.data_ptr(), d_nr.data_ptr(), d_na.data_ptr(), n, window)
            e.run(targets, ld=True)

        def stream_of(k):
            submit(engs[0])
            t0 = time.perf_counter()
            for i in range(1, k + 1):
                submit(engs[i & 1])
                engs[(i - 1) & 1].window_ll(0, out=outs[(i - 1) & 1].array)
            dt = (time.perf_counter() - t0) / k
            engs[k & 1].window_ll(0, out=outs[k & 1].array)
            return dt * 1e3
        stream_of(6)
        alt = min(stream_of(40) for _ in range(3))
        for e in engs:
            e.set_option("async", 0)
            e.set_option("dev_inputs_ready", 0)
        ec["two_contexts_alternating_ms"] = alt
        ec["two_contexts_alternating_sites_per_s"] = n_cov / (alt * 1e-3)
        ec["two_contexts_alternating_bits_equal_single"] = bool(
            np.array_equal(single, win_pin.array) and np.array_equal(single, win_pin2.array))
        ec["two_contexts_alternating_note"] = (
            "40 comparisons through two ibdg contexts on one GPU taking turns, option dev_inputs_ready (upload_sites_dev + run of i+1 queued before the "
            "window table of i is fetched); host wall clock per comparison, best of 3; a throughput, where `ms` above is the "
            "latency of one comparison")
        win_pin2.close()
    # per-site results (96 MB at 4M rows): pageable vs page-locked destination
    eng.run(targets, ld=True)
    site_pin = ibdgem_amd.PinnedArray((n, 3), np.float64)
    page = np.empty((n, 3))
    eng.site_ll(0, out=page)
    t_page = best_of(3, lambda: eng.site_ll(0, out=page))
    eng.site_ll(0, out=site_pin.array)
    t_pin = best_of(3, lambda: eng.site_ll(0, out=site_pin.array))
    d2h = {"per_site_results_ms": t_pin, "per_site_results_GBps": page.nbytes / (t_pin * 1e-3) / 1e9,
           "per_site_results_pageable_ms": t_page, "per_site_results_pageable_GBps": page.nbytes / (t_page * 1e-3) / 1e9,
           "note": "ibdg_get_site_ll of one comparison individual into page-locked (ibdg_host_alloc) / pageable memory"}
    for a in pins + [win_pin, site_pin]:
        a.close()
    del d_idx, d_nr, d_na
    return up, ec, d2h


def sparse_pileup_clocks(torch, eng, ibdgem_amd, n_ref, n_alt, window, target, many_t, seed):
    """The reference's real inputs are thin pileups (low-coverage ancient DNA): a panel row without a pileup line
    never reaches the window loop (src/ibdgem.c:596-601), and every row that does costs the same (:669-722).
    Engine clock -- site arrays resident in HBM -> window tables in host memory: ibdg_upload_sites_dev (site
    preparation AND the gather + transposition of the compacted tiles inside) + ibdg_run + ibdg_get_window_ll_all (every
    comparison individual's window table in one copy) -- with a pileup on 10 % and on 2 % of this rank's panel rows, one comparison
    individual and many; beside it the same with the compacted tiles forbidden (option compact_tiles -1: what
    round 3 did with such a pileup -- the strict fp64 kernel), and the largest relative difference between the two
    window tables."""
    n = len(n_ref)
    rng = np.random.default_rng(seed + 77)
    out = {}
    for share in (0.10, 0.02):
        rows = np.sort(rng.choice(n, size=int(n * share), replace=False)).astype(np.uint32)
        nr, na = n_ref[rows].copy(), n_alt[rows].copy()
        nr[(nr.astype(np.int32) + na) == 0] = 1            # a pileup line means a read
        k = len(rows)
        d_idx = torch.from_numpy(rows.view(np.int32)).cuda()
        d_nr, d_na = torch.from_numpy(nr).cuda(), torch.from_numpy(na).cuda()
        torch.cuda.synchronize()
        leg = {"panel_rows": int(n), "pileup_rows": int(k), "windowed_sites": int(k)}
        for name, tg in (("one_individual", [target]), ("many_individuals", many_t)):
            T = len(tg)
            res = {}
            tables = {}
            for tiles in (0, -1):
                eng.set_option("compact_tiles", tiles)
                eng.set_option("async", 1)
                eng.upload_sites_dev(d_idx.data_ptr(), d_nr.data_ptr(), d_na.data_ptr(), k, window)
                n_win = eng.n_windows
                pin = ibdgem_amd.PinnedArray((T, n_win, 3), np.float64)

                def once():
                    eng.upload_sites_dev(d_idx.data_ptr(), d_nr.data_ptr(), d_na.data_ptr(), k, window)
                    eng.run(tg, ld=True)
                    eng.window_ll_all(T, out=pin.array)       # every individual's window table in one copy (ibdg_get_window_ll_all)
                for _ in range(3):
                    once()
                ms = []
                for _ in range(5 if tiles < 0 and T > 1 else 10):
                    t0 = time.perf_counter()
                    once()
                    ms.append((time.perf_counter() - t0) * 1e3)
                eng.set_option("async", 0)
                c = eng.upload_ms()
                r = eng.last_run_ms()
                tables[tiles] = pin.array.copy()
                pin.close()
                res["compacted_tiles" if tiles == 0 else "compacted_tiles_forbidden"] = {
                    "ms": min(ms), "median_ms": float(np.median(ms)),
                    "site_individual_pairs_per_s": k * T / (min(ms) * 1e-3),
                    "ld_variant": eng.last_ld_variant(), "ld_layout": eng.ld_layout(), "windows": int(n_win),
                    "upload_device_prep_ms": c["device_prep"], "ld_launches_ms": r["ld"], "rows_kernel_ms": r["rows"]}
            a, b = tables[0][..., :2], tables[-1][..., :2]
            big = np.abs(b) >= 1e-290
            res["max_rel_diff_of_the_two"] = float((np.abs(a[big] - b[big]) / np.abs(b[big])).max()) if big.any() else 0.0
            res["libd2_bits_equal"] = bool(np.array_equal(tables[0][..., 2], tables[-1][..., 2]))
            res["speedup"] = res["compacted_tiles_forbidden"]["ms"] / res["compacted_tiles"]["ms"]
            res["comparison_individuals"] = T
            leg[name] = res
        out[f"pileup_on_{int(share * 100)}pct_of_rows"] = leg
        del d_idx, d_nr, d_na
    eng.set_option("compact_tiles", 0)
    out["note"] = ("engine clock (host wall, best of 10 after 3 untimed; upload_sites_dev + run + all window tables to page-locked "
                   "memory in one copy, option async), the re-layout inside it; ld_variant 2 = exponent counting / matrix cores, 1 = strict fp64 "
                   "products; ld_layout 2 = compacted tiles of the site list, 1 = the panel's own tiles")
    return out



def valu_roofline(launch_ms, n_win, n_chunks):
    """The second roofline of the dominant kernel: VALU issue.
    Instructions per launch come from the committed PMC passes of this very workload (profiles/*_ld_pmc.json,
    tools/pmc_ld.sh), cycles per instruction from the micro-benchmark of the kernel's own instruction mix
    (profiles/*_dep_distance.txt); issue cycles per SIMD = instructions x cycles / 1024 SIMDs.  Reported
    against the kernel's own cycle count (GRBM_GUI_ACTIVE / 8 XCDs, same passes) and, as a time, at the
    nominal 2.4 GHz -- under this dense integer load the chip sustains about 2.0 GHz (kernel cycles /
    kernel time).  None when no PMC file matches the workload."""
    pdir = os.path.join(REPO, "profiles")
    best = None
    for fn in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if fn.endswith("_ld_pmc.json"):
            with open(os.path.join(pdir, fn)) as fh:
                best = (fn, json.load(fh))
    if best is None:
        return None
    fn, p = best
    c = p.get("config", {})
    if (c.get("n_win"), c.get("n_chunks")) != (n_win, n_chunks):
        return None
    k, d = p["per_launch"], p["derived"]
    cyc = p["cycles_per_valu_instruction"]["value"]
    mfma = k.get("SQ_INSTS_MFMA", 0.0) if p.get("cycles_per_mfma") else 0.0
    issue = ((k["SQ_INSTS_VALU"] - mfma) * cyc + mfma * (p.get("cycles_per_mfma") or 0.0)) / p["simds"]
    bound_ms = issue / p["nominal_clock_hz"] * 1e3
    busy = {n: k[n] for n in ("VALUBusy", "LdsUtil", "SALUBusy", "MeanOccupancyPerCU") if n in k}
    return {"bound": "valu-issue", "profile": fn, "instruction_counts_replayed_from_profile": True, "valu_instructions_per_launch": k["SQ_INSTS_VALU"],
            "mfma_instructions_per_launch": mfma or None, "cycles_per_mfma": p.get("cycles_per_mfma"),
            "busy_shares_of_the_profile_pct": busy or None,
            "valu_instructions_per_window_and_chunk": d["valu_per_window_chunk"],
            "cycles_per_instruction": cyc, "simds": p["simds"], "issue_cycles_per_simd": issue,
            "kernel_cycles_profiled": d["kernel_cycles"], "frac_of_kernel_cycles": issue / d["kernel_cycles"],
            "bound_ms_at_2.4GHz": bound_ms, "achieved_ms": launch_ms, "frac_at_2.4GHz": bound_ms / launch_ms,
            "sustained_clock_GHz": d["kernel_cycles"] / (launch_ms * 1e-3) / 1e9,
            "note": "frac_of_kernel_cycles: share of the kernel's cycles in which the SIMDs must be issuing its VALU "
                    "instructions (~1 = at the VALU-issue roofline); frac_at_2.4GHz: the same bound as a time at the "
                    "nominal clock over the measured kernel time (the rest is the clock the chip sustains)"}


def write_pileup_and_legend(d, n_ref, n_alt, n_ids, rows):
    """Reference text inputs for `rows` rows (legend, indv, pileup; the .hap comes from the caller)."""
    with open(os.path.join(d, "p.legend"), "w") as fh:
        fh.write("ID pos allele0 allele1\n")
        for a in range(0, rows, 500_000):
            fh.write("".join([f"rs{i} {100 + 10 * i} A C\n" for i in range(a, min(rows, a + 500_000))]))
    with open(os.path.join(d, "p.indv"), "w") as fh:
        fh.write("".join(f"ind{n}\n" for n in range(n_ids)))
    tails = {}
    for r in range(21):
        for a in range(21 - r):
            c = r + a
            tails[r * 32 + a] = ("N\t0\t*\t*\t*\n" if c == 0 else
                                 f"N\t{c}\t{'A' * r}{'C' * a}\t{'I' * c}\t{'I' * c}\n")
    key = (n_ref[:rows].astype(np.int32) * 32 + n_alt[:rows]).tolist()
    with open(os.path.join(d, "p.pileup"), "w") as fh:
        for a in range(0, rows, 500_000):
            fh.write("".join([f"1\t{100 + 10 * i}\t{tails[key[i]]}" for i in range(a, min(rows, a + 500_000))]))


def write_panel_cache(path, words_all, n_ids, st):
    """The host program's packed-panel cache (ibdgem_amd/host/ingest.c): header naming the .hap file's size and
    mtime | row-is-clean flags | alt-allele counts | packed rows, the last two on 4096-byte boundaries."""
    import struct
    rows = words_all.shape[0]
    with open(path, "wb") as fh:
        hdr = struct.pack("<8sIIQQQqq", b"IBDGPNL3", n_ids, 0, rows, words_all.shape[1], st.st_size,
                          st.st_mtime_ns // 1_000_000_000, st.st_mtime_ns % 1_000_000_000)
        fh.write(hdr)
        fh.write(np.ones(rows, dtype=np.uint8).tobytes())
        fh.write(b"\0" * (-(len(hdr) + rows) % 4096))
        for a in range(0, rows, 500_000):
            fh.write(np.bitwise_count(words_all[a:a + 500_000]).sum(axis=1, dtype=np.uint32).tobytes())
        fh.write(b"\0" * (-(rows * 4) % 4096))
        for a in range(0, rows, 500_000):
            fh.write(np.ascontiguousarray(words_all[a:a + 500_000]).tobytes())


ALL_RUNS = {}            # every wall clock behind a best-of figure of the host-program legs, by output directory


def timed_run(cmd, cwd, repeat=3, env=None):
    """Best of `repeat` runs (the first run of the host program after this process's own GPU work is regularly
    0.3 s slower than the ones after it; all times are kept in ALL_RUNS)."""
    times = []
    for _ in range(repeat):
        t0 = time.perf_counter()
        subprocess.run(cmd, cwd=cwd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       env=dict(os.environ, **env) if env else None)
        times.append(time.perf_counter() - t0)
    ALL_RUNS[cmd[cmd.index("-O") + 1] if "-O" in cmd else str(len(ALL_RUNS))] = times
    return min(times)


def run_phases(cmd, cwd, env=None):
    """One more run with IBDGEM_TIMING=1: the host program's own wall clock per phase (summed per name), the wall
    clock of that run as its parent sees it, and what the phases do not cover (the end of the process: the driver
    taking the device memory and the mappings back)."""
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=cwd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, IBDGEM_TIMING="1", **(env or {})))
    wall = time.perf_counter() - t0
    out = {}
    for line in r.stderr.splitlines():
        if line.startswith("## time "):
            name, sec = line[8:].rsplit(" ", 1)
            out[name] = out.get(name, 0.0) + float(sec)
    out["(wall clock of this run)"] = wall
    out["(not covered by the phases: process exit)"] = max(0.0, wall - sum(v for k, v in out.items() if not k.startswith("(")))
    return out


def end_to_end_clocks(words_all, n_ref, n_alt, n_ids, target, window, cold_rows, gpu_win):
    """warm_e2e: the host program from the packed-panel cache + legend + pileup text of ALL rows to its
    output files (summary only, and with the per-site table).  cold_e2e: from .hap text, on the first
    cold_rows rows (the whole chromosome would be 40 GB of text), with the fixed cost of a run (process
    start, device initialisation) measured on 1000 rows so the linear extrapolation is explicit.
    Reference: src/ibdgem.c:522-773 end to end (it re-reads and re-parses the text per individual)."""
    import struct
    exe = os.path.join(REPO, "ibdgem_amd", "host", "ibdgem")
    if not os.path.exists(exe):
        return {"error": "ibdgem_amd/host/ibdgem not built"}, None
    rows = words_all.shape[0]
    threads = str(max(1, min(16, len(os.sched_getaffinity(0)))))
    base_dir = "/dev/shm" if os.path.isdir("/dev/shm") else os.environ.get("TMPDIR", "/tmp")
    warm = cold = None
    with tempfile.TemporaryDirectory(dir=base_dir) as d:
        # ---- warm: every row.  The 40 GB of .hap text behind the cache are not written: the cache header
        # names the size and mtime of a placeholder p.hap, which is all the host program checks.
        write_pileup_and_legend(d, n_ref, n_alt, n_ids, rows)
        with open(os.path.join(d, "p.hap"), "w") as fh:
            fh.write("placeholder: the packed-panel cache p.cache stands for the .hap text\n")
        st = os.stat(os.path.join(d, "p.hap"))
        write_panel_cache(os.path.join(d, "p.cache"), words_all, n_ids, st)
        base = [exe, "-H", "p.hap", "-L", "p.legend", "-I", "p.indv", "-P", "p.pileup", "-s", f"ind{target}", "--LD",
                "-w", str(window), "--threads", threads, "--panel-cache", "p.cache"]
        os.makedirs(os.path.join(d, "o1"))
        os.makedirs(os.path.join(d, "o2"))
        t_sum = timed_run(base + ["-O", "o1", "--summary-only"], d)
        t_tab = timed_run(base + ["-O", "o2"], d)
        got = [l.split("\t")[3:6] for l in open(os.path.join(d, "o1", f"UNKWN.ind{target}.summary.txt")) if not l.startswith("#")]
        same = gpu_win is not None and len(got) == len(gpu_win) and all(
            ["%e" % v for v in gpu_win[w]] == got[w] for w in range(len(got)))
        warm = {"rows": rows, "summary_only_s": t_sum, "with_per_site_table_s": t_tab,
                "phases_summary_only_s": run_phases(base + ["-O", "o1", "--summary-only"], d),
                "phases_with_per_site_table_s": run_phases(base + ["-O", "o2"], d),
                "rows_per_s_summary_only": rows / t_sum, "rows_per_s_with_per_site_table": rows / t_tab,
                "per_site_table_bytes": os.path.getsize(os.path.join(d, "o2", f"UNKWN.ind{target}.tab.txt")),
                "summary_equals_engine_windows_7digits": bool(same), "host_threads": int(threads), "all_runs_s": ALL_RUNS,
                "note": "ibdgem_amd/host/ibdgem --LD, packed-panel cache (2.56 GB) + legend + pileup text -> output files, "
                        "process start and device initialisation included, files in page cache, best of 3"}
        # the same comparison spread over four engine contexts (--devices 0,0,0,0: all on THIS box's one GPU, so no
        # speed-up is to be had; what it shows is the cost of the decomposition -- every context receives only the
        # panel rows of its window range, all uploads at once -- and that the summary file does not change by a byte)
        os.makedirs(os.path.join(d, "o5"))
        t_four = timed_run(base + ["-O", "o5", "--summary-only", "--devices", "0,0,0,0"], d)
        same4 = open(os.path.join(d, "o5", f"UNKWN.ind{target}.summary.txt"), "rb").read() == \
            open(os.path.join(d, "o1", f"UNKWN.ind{target}.summary.txt"), "rb").read()
        warm["four_contexts_one_gpu"] = {"summary_only_s": t_four, "summary_identical_to_one_context": bool(same4),
                                         "phases_s": run_phases(base + ["-O", "o5", "--summary-only", "--devices", "0,0,0,0"], d)}
        # the same with 30 comparison individuals (one batch of the host program, two groups of k_ld_mfma): what a
        # further individual costs end to end once the panel is on the device
        many_names = ",".join(f"ind{(target + 5 * i) % n_ids}" for i in range(30))
        mbase = [a if a != f"ind{target}" else many_names for a in base]
        os.makedirs(os.path.join(d, "o4"))
        t_many = timed_run(mbase + ["-O", "o4", "--summary-only"], d)
        ph_many = run_phases(mbase + ["-O", "o4", "--summary-only"], d)
        warm["many_individuals"] = {"individuals": 30, "summary_only_s": t_many,
                                    # from the program's own phase clocks: engine + output files of the 30, the site list being
                                    # built once.  (The difference of two wall clocks, which this field was until round 3, mostly
                                    # measures whether the two runs met the driver's 0.2 s at process end and 0.13 s at device start.)
                                    "s_per_further_individual": sum(v for k, v in ph_many.items()
                                                                    if k.startswith("per individual: engine") or k.startswith("per individual: output")
                                                                    or k.startswith("per individual: waiting") or k.startswith("output files of the last")) / 30,
                                    "wall_clock_difference_per_individual_s": (t_many - t_sum) / 29,
                                    "phases_s": ph_many}
        # ... and with 600 of them (a quarter of the whole-panel job the reference's loop is for, src/ibdgem.c:522: one pileup
        # against every individual of the panel; batches of 30, the next batch queued on the device while the host writes
        # this one's summary files, twelve files at a time)
        n_600 = min(600, n_ids)
        names600 = ",".join(f"ind{(target + 5 * i) % n_ids}" for i in range(n_600))
        qbase = [a if a != f"ind{target}" else names600 for a in base]
        os.makedirs(os.path.join(d, "o7"))
        t_600 = timed_run(qbase + ["-O", "o7", "--summary-only"], d, repeat=2)
        ph_600 = run_phases(qbase + ["-O", "o7", "--summary-only"], d)
        n_files = len([f for f in os.listdir(os.path.join(d, "o7")) if f.endswith(".summary.txt")])
        warm["six_hundred_individuals"] = {"individuals": n_600, "summary_only_s": t_600, "summary_files_written": n_files,
                                           "s_per_individual_from_phases": sum(v for k, v in ph_600.items()
                                                                               if k.startswith("per individual") or k.startswith("output files of the last")) / n_600,
                                           "phases_s": ph_600}
        for fn in os.listdir(os.path.join(d, "o7")):
            os.remove(os.path.join(d, "o7", fn))
        # eight individuals WITH their per-site tables (the default run): 8 x 330 MB of text.  The files of up to four
        # individuals are written beside the main thread's work on the ones after them; IBDGEM_OUT_SLOTS=1 is one at a time.
        eight = ",".join(f"ind{(target + 5 * i) % n_ids}" for i in range(8))
        ebase = [a if a != f"ind{target}" else eight for a in base]
        os.makedirs(os.path.join(d, "o6"))
        per = {}
        for label, env in (("side_by_side", None), ("one_at_a_time", {"IBDGEM_OUT_SLOTS": "1"})):
            t8 = timed_run(ebase + ["-O", "o6"], d, repeat=2, env=env)
            ph8 = run_phases(ebase + ["-O", "o6"], d, env=env)
            own = sum(v for k, v in ph8.items() if k.startswith("per individual: engine") or k.startswith("per individual: output")
                      or k.startswith("per individual: waiting") or k.startswith("output files of the last"))
            per[label] = {"s": t8, "s_per_individual_from_phases": own / 8, "phases_s": ph8}
        warm["eight_individuals_with_tables"] = dict(per, table_bytes_each=warm["per_site_table_bytes"],
                                                     note="default run (tables written) of 8 comparison individuals in one batch")
        for fn in os.listdir(os.path.join(d, "o6")):
            os.remove(os.path.join(d, "o6", fn))
        for fn in ("p.cache",):
            os.remove(os.path.join(d, fn))
        # ---- cold: text .hap of the first cold_rows rows
        r = min(cold_rows, rows)
        alle = None
        with open(os.path.join(d, "p.hap"), "wb") as fh:
            for a in range(0, r, 50_000):
                alle = unpack_rows(np.ascontiguousarray(words_all[a:min(r, a + 50_000)]), n_ids)
                hap = np.full((alle.shape[0], 4 * n_ids), ord(" "), dtype=np.uint8)
                hap[:, 0::2] = alle + ord("0")
                hap[:, -1] = ord("\n")
                fh.write(hap.tobytes())
        del alle, hap
        write_pileup_and_legend(d, n_ref, n_alt, n_ids, r)
        cbase = [exe, "-H", "p.hap", "-L", "p.legend", "-I", "p.indv", "-P", "p.pileup", "-s", f"ind{target}", "--LD",
                 "-w", str(window), "--threads", threads]
        os.makedirs(os.path.join(d, "o3"))
        t_cold = timed_run(cbase + ["-O", "o3"], d)
        # fixed cost: the same program on the first 1000 rows
        os.makedirs(os.path.join(d, "tiny"))
        with open(os.path.join(d, "p.hap"), "rb") as fi, open(os.path.join(d, "tiny", "p.hap"), "wb") as fo:
            fo.write(fi.read(1000 * 4 * n_ids))
        write_pileup_and_legend(os.path.join(d, "tiny"), n_ref, n_alt, n_ids, 1000)
        t_fixed = timed_run(cbase + ["-O", "."], os.path.join(d, "tiny"))
        ph = run_phases(cbase + ["-O", "o3"], d)
        # what does not grow with the rows, from the program's own phase clocks of this very run: device start (the part
        # parsing did not hide), engine shutdown, and whatever the phases do not cover (process start and exit)
        fixed_part = sum(v for k, v in ph.items() if k.startswith("device start") or k.startswith("engine shutdown") or
                         k.startswith("process start"))
        fixed_part += max(0.0, t_cold - sum(v for k, v in ph.items() if not k.startswith("(")))
        fixed_part = min(fixed_part, t_cold)
        cold = {"rows": r, "s": t_cold, "rows_per_s": r / t_cold, "fixed_cost_s": t_fixed, "fixed_part_s": fixed_part,
                "phases_s": ph,
                "hap_text_bytes": r * 4 * n_ids,
                "extrapolated_to_all_rows_s": fixed_part + (t_cold - fixed_part) * rows / r,
                "note": f"ibdgem_amd/host/ibdgem --LD, .hap/.legend/.pileup text -> both output files on the first {r} rows "
                        f"(the whole chromosome is {rows * 4 * n_ids / 1e9:.0f} GB of text); fixed_cost_s = the same program on "
                        "1000 rows (process start + device initialisation, nothing to hide it behind); fixed_part_s = what does not "
                        "grow with the rows in THIS run (device start not hidden by parsing, shutdown, process start, from the "
                        "program's phase clocks); extrapolation linear in the rows for the rest"}
    return warm, cold


def traffic_bytes(args, world):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/*_ld_traffic.json, tools/pmc_traffic.sh) when they were taken on this very
    workload; None otherwise (counters cannot be read from inside the process)."""
    best, src = None, None
    pdir = os.path.join(REPO, "profiles")
    for fn in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not fn.endswith("_ld_traffic.json"):
            continue
        with open(os.path.join(pdir, fn)) as fh:
            t = json.load(fh)
        c = t.get("config", {})
        if world == 1 and (c.get("sites"), c.get("n_ids"), c.get("window")) == (args.sites, args.ids, args.window):
            best, src = t.get("dominant_kernel_hbm_bytes_per_launch"), fn
    return best, src


# ----------------------------------------------------------------------------- the line the driver reads
LINE_LIMIT = 4096


def _sig(x, n=6):
    """Numbers of the compact line: n significant digits (the full values are in the detail file)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        return float(f"{x:.{n}g}") if x == x and abs(x) != float("inf") else None
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    return x


def _pick(d, keys):
    return {k: d.get(k) for k in keys} if isinstance(d, dict) else None


TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "clock", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data")
CONFIG_KEYS = ("workload", "rows", "windowed_sites", "n_ids", "window", "targets", "sharding")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "bytes_per_site", "sites_per_launch", "kernel_ms",
                 "dominant_kernel_only_ms", "traffic", "traffic_from_profile")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "ld_stage_only_sites_per_s")
RANK_KEYS = ("rank", "rows", "windowed_sites", "windows", "ms_per_step", "ld_launch_ms", "step_device_ms", "many_device_ms",
             "many_wall_ms")


def compact_line(out, detail_file):
    """The ONE line rank 0 prints last: the contract's keys, `roofline`, `cpu_baseline`, numbers per rank -- at most
    LINE_LIMIT bytes whatever the number of ranks (round 4's line had grown to 22 KB and the driver could not parse it)."""
    line = {k: out.get(k) for k in TOP_KEYS}
    line["config"] = _pick(out.get("config"), CONFIG_KEYS)
    line["roofline"] = _pick(out.get("roofline"), ROOFLINE_KEYS)
    cb = _pick(out.get("cpu_baseline"), CPU_KEYS)
    if cb and isinstance(cb.get("sample"), str) and len(cb["sample"]) > 300:
        cb["sample"] = cb["sample"][:297] + "..."
    line["cpu_baseline"] = cb
    # (the parity gate that goes with the number: the reference's summary file of the sample against the GPU's windows, 7 digits)
    line["parity"] = {"reference_summary_equals_gpu_windows_7digits": (out.get("cpu_baseline") or {}).get("summary_matches_gpu_7digits"),
                      "host_program_summary_equals_engine_windows_7digits": (out.get("warm_e2e") or {}).get("summary_equals_engine_windows_7digits")
                      if isinstance(out.get("warm_e2e"), dict) else None}
    ranks = []
    for p in out.get("per_rank") or []:
        r = _pick(p, RANK_KEYS[:7])
        m = p.get("many_comparison_individuals") if isinstance(p, dict) else None
        if m:
            r["many_device_ms"], r["many_wall_ms"] = m.get("device_ms"), m.get("wall_ms")
        ranks.append(r)
    line["per_rank"] = ranks
    line["barrier_ms"] = out.get("barrier_ms")
    ec = out.get("engine_clock") or {}
    line["engine_sites_per_s"] = ec.get("sites_per_s")
    line["engine_clock_ms"] = ec.get("ms")
    line["new_individual_per_step"] = out.get("new_individual_per_step")
    # (the step without the engine's once-per-site-list pass for the IBD0 terms, same queue and tiles: beside `value`'s step)
    at = out.get("all_terms_in_every_step") or {}
    line["all_terms_in_every_step_ms"] = at.get("ms_per_step")
    many = out.get("many_comparison_individuals") or {}
    line["many_individuals"] = _pick(many, ("comparison_individuals", "ms_per_individual", "site_individual_pairs_per_s")) if many else None
    # (like `value`: the cost on a site list in use -- its rows re-laid out back to back, which runs of many individuals bring about
    #  by themselves after thirteen groups of 15 -- beside the first runs' on the panel's own tiles)
    fc = many.get("from_compacted_tiles") if isinstance(many, dict) else None
    if line["many_individuals"] and isinstance(fc, dict) and fc.get("ms_per_individual"):
        line["many_individuals"]["ms_per_individual_first_runs"] = line["many_individuals"]["ms_per_individual"]
        line["many_individuals"]["ms_per_individual"] = fc["ms_per_individual"]
        line["many_individuals"]["tiles"] = "compacted"
        if many.get("site_individual_pairs_per_s") and many.get("ms_per_individual"):
            line["many_individuals"]["site_individual_pairs_per_s"] = (many["site_individual_pairs_per_s"] * many["ms_per_individual"]
                                                                     / fc["ms_per_individual"])
    for k in ("step_vs_reference_end_to_end", "engine_clock_vs_reference_end_to_end", "ld_kernels_vs_reference_ld_stage"):
        line[k] = out.get(k)
    line["detail_file"] = detail_file
    line = _sig(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:                       # (cannot happen at <= 8 ranks; drop the optional parts rather than the contract's)
        for k in ("parity", "many_individuals", "ld_kernels_vs_reference_ld_stage", "engine_clock_vs_reference_end_to_end",
                  "step_vs_reference_end_to_end", "engine_clock_ms", "new_individual_per_step", "all_terms_in_every_step_ms"):
            line.pop(k, None)
        line["per_rank"] = [_pick(r, ("rank", "windowed_sites", "ms_per_step", "ld_launch_ms")) for r in line["per_rank"]]
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= LINE_LIMIT, f"the result line is {len(text)} bytes (limit {LINE_LIMIT})"
    return text


def write_detail(out, world):
    """Everything measured, notes included, as indented JSON under gpurun_out/ (merged back from a GPU box); the path is
    relative to the repository root."""
    rel = os.path.join("gpurun_out", "bench_detail.json" if world == 1 else f"bench_detail_{world}gpu.json")
    try:
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, rel), "w") as fh:
            json.dump(out, fh, indent=1)
            fh.write("\n")
    except OSError as e:                              # a read-only checkout: the line still goes out
        rel = f"(not written: {e.strerror})"
    return rel


# ----------------------------------------------------------------------------- main
def launcher_command(n_ranks, argv):
    """The command `python bench.py --gpus N` runs when nobody launched it under torch.distributed.run: one rank
    per GPU of this node, rendezvous on 127.0.0.1 (the container's hostname may not resolve) at a free port."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)       # ~0.9 ms each; the clocks settle after ~30 steps
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--sites", type=int, default=4_000_000, help="SNP rows of the synthetic chromosome")
    ap.add_argument("--ids", type=int, default=2504)
    ap.add_argument("--window", type=int, default=100)
    ap.add_argument("--target", type=int, default=7)
    ap.add_argument("--seed", type=int, default=20241008)
    ap.add_argument("--cpu-sample-rows", type=int, default=200_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-program clocks (warm_e2e, cold_e2e)")
    ap.add_argument("--no-many", action="store_true", help="skip the run over many comparison individuals")
    ap.add_argument("--many-targets", type=int, default=60, help="comparison individuals of that run (configs[4] shape)")
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed steps before the warm-up steps until this much wall time has passed: the chip's clocks take "
                         "tens of ms of load to settle, more than a short --warmup at a fraction of a millisecond per step "
                         "provides (80 ms until round 3: --steps 20 --warmup 5 then gave 4.08 / 4.26 / 4.46e9 in three fresh "
                         "processes on one box, 4.41 / 4.47 with 300 ms, 4.44 with 600; profiles/r03_prewarm.txt)")
    ap.add_argument("--timed-only", action="store_true",
                    help="warm-up and timed steps only (no recount / engine-clock / upload / host legs): the command to put under "
                         "rocprofv3 --kernel-trace --stats, so that the per-kernel averages are those of the timed steps")
    ap.add_argument("--cold-rows", type=int, default=400_000, help="rows of .hap text for the cold end-to-end clock")
    ap.add_argument("--variant", type=int, default=None, help="ld_variant option of the engine")
    ap.add_argument("--cpw", type=int, default=None)
    ap.add_argument("--waves", type=int, default=None)
    ap.add_argument("--sync-steps", action="store_true", help="wait for the device after every step")
    ap.add_argument("--same-target", action="store_true",
                    help="every step runs the SAME comparison individual again (the timed steps of round 4: the individual's images "
                         "are then reused from the step before); default: two individuals in turn, a new one per step")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (repeatable)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as `python bench.py --gpus N` without a launcher: start the N ranks ourselves, as fresh child
        # processes, before this process has imported torch or made any GPU call (it never does), hand rank 0's
        # JSON line through (the children inherit stdout) and leave with the launcher's exit code
        sys.exit(subprocess.run(launcher_command(args.gpus, sys.argv[1:])).returncode)

    import torch
    import torch.distributed as dist
    import ibdgem_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher started a different number of ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    # BENCH_FORCE_DEVICE / BENCH_BACKEND exist only to rehearse the multi-rank path on a one-GPU box
    # (all ranks on one device, gloo instead of RCCL); the driver's runs use neither.
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local = int(os.environ["BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    # BENCH_DIST_ONE_RANK=1 (rehearsal only): a single rank goes through every torch.distributed call of the N-rank run
    # all the same -- process group over RCCL, barriers, reductions, the gather -- since a one-GPU box cannot hold two
    # RCCL ranks; the workload and its legs stay those of N = 1
    use_dist = world > 1 or os.environ.get("BENCH_DIST_ONE_RANK") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # Every rank derives the same global cut points from the (cheap) read counts of all rows;
    # only its own panel shard is materialised.
    L = args.sites
    if world > 1:
        nr_all, na_all = [], []
        for blk in range((L + BLOCK_ROWS - 1) // BLOCK_ROWS):
            _, _, nr, na = gen_counts(torch, dev, blk, args.seed)      # a few bytes per row; no panel bits
            nr_all.append(nr.cpu().numpy())
            na_all.append(na.cpu().numpy())
        nr_all = np.concatenate(nr_all)[:L]
        na_all = np.concatenate(na_all)[:L]
        from ibdgem_amd.sharding import shard_rows
        cuts = shard_rows(nr_all, na_all, args.window, world)
        row0, row1 = cuts[rank], cuts[rank + 1]
    else:
        row0, row1 = 0, L
    panel, n_ref, n_alt = build_shard(torch, dev, row0, row1, args.ids, args.target, args.seed)
    torch.cuda.synchronize()

    eng = ibdgem_amd.Engine(local, 0.02, 20)
    if args.variant is not None:
        eng.set_option("ld_variant", args.variant)
    if args.cpw is not None:
        eng.set_option("chunks_per_wave", args.cpw)
    if args.waves is not None:
        eng.set_option("waves_per_block", args.waves)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    eng.upload_panel_dev(panel.data_ptr(), panel.shape[0], args.ids)
    eng2 = None
    want_second = world == 1 and not args.no_many and not args.timed_only

    def second_context(panel_t):
        # a second context with the same panel, for engine_clock.two_contexts_alternating_ms.  Made when that leg begins, not
        # before the timed steps: two contexts hold six streams, more than the runtime's hardware queues (4 by default), so
        # that the first context's three streams would share queues while `value` is measured -- the step came out 2-5 %
        # slower than `--timed-only`'s in the same call (0.511 against 0.487 ms in the round's last evidence call)
        e2 = ibdgem_amd.Engine(local, 0.02, 20)
        for name, val in (("ld_variant", args.variant), ("chunks_per_wave", args.cpw), ("waves_per_block", args.waves)):
            if val is not None:
                e2.set_option(name, val)
        for kv in args.opt:
            k, v = kv.split("=")
            e2.set_option(k, int(v))
        e2.upload_panel_dev(panel_t.data_ptr(), panel_t.shape[0], args.ids)
        return e2
    sample_rows = min(args.cpu_sample_rows, panel.shape[0])
    sample_words = panel[:sample_rows].cpu().numpy().view(np.uint64) if rank == 0 else None
    words_all = None
    if rank == 0 and world == 1 and not args.no_e2e:       # the whole packed panel on the host (2.56 GB) for warm_e2e
        words_all = np.empty((panel.shape[0], panel.shape[1]), dtype=np.uint64)
        for a in range(0, panel.shape[0], 500_000):
            words_all[a:a + 500_000] = panel[a:a + 500_000].cpu().numpy().view(np.uint64)
    panel_kept = panel if want_second else None      # (2.56 GB held until the second context has copied it)
    del panel
    torch.cuda.empty_cache()
    n_rows = row1 - row0
    t_up = time.perf_counter()
    eng.upload_sites(np.arange(n_rows, dtype=np.uint32), n_ref, n_alt, args.window)
    first_upload_ms = (time.perf_counter() - t_up) * 1e3
    n_cov = int(((n_ref.astype(np.int32) + n_alt) > 0).sum())
    # (window, 32-row tile) segments of this rank's rows: the unit of work of the --LD kernels' inner loop
    cov_rows = np.flatnonzero((n_ref.astype(np.int32) + n_alt) > 0)
    seg_start = (np.arange(len(cov_rows)) % args.window == 0)
    seg_start[1:] |= (cov_rows[1:] >> 5) != (cov_rows[:-1] >> 5)
    n_segments_in_place = int(seg_start.sum())
    # ... and on the compacted tiles (the covered rows back to back: row j sits in tile j // 32)
    jj = np.arange(len(cov_rows))
    n_segments_compact = int(((jj % args.window == 0) | (jj % 32 == 0)).sum())
    del jj
    del cov_rows, seg_start
    targets = [args.target]
    # a NEW comparison individual per step (src/ibdgem.c:522 hands every individual of the panel to the same rows in turn):
    # two panel members take turns, so no step finds its individual's images, weights or background size left by the step
    # before.  (The reads were drawn from args.target's genotype; the cost of a step does not depend on the values.)
    turn = [targets] if args.same_target else [targets, [(args.target + 1) % args.ids]]
    n_run = [0]

    def step():
        eng.run(turn[n_run[0] % len(turn)], ld=True)
        n_run[0] += 1

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    # The engine re-lays a site list out into compacted tiles (its rows with reads back to back) once the runs on it have added
    # up to what the gather costs (22 runs of one comparison individual; DESIGN s3, s4.1 -- the reference's own loop runs every
    # individual of the panel over the same rows, src/ibdgem.c:522).  That happens here, untimed and reported: the
    # timed steps below are steps of a site list in use, `in_place_tiles` further down is the same step before it.
    # ... and once eight single runs on an upload have gone by, ONE pass keeps what the IBD0 terms have in common for every
    # comparison individual (option ibd0_after; DESIGN s4.1): later runs count the IBD1 sums only (the pass is made again on
    # the new tiles by the run that re-lays out).  Also here, untimed and reported.
    relayout = {"after_runs": None, "run_ms": None}
    ibd0_pass = {"after_runs": None, "run_ms": None}
    layout0 = eng.ld_layout()
    t_r = time.perf_counter()
    eng.run(targets, ld=True)
    eng.sync()
    if eng.last_count_unit() == 3:
        ibd0_pass = {"after_runs": 1, "run_ms": (time.perf_counter() - t_r) * 1e3}
    # (a single run counts 12 towards compact_targets = 256 -- 16 with option mx_counts 0 --: the 22nd run re-lays out)
    for k in range(40):
        if (eng.ld_layout() != layout0 or layout0 == 2) and eng.last_count_unit() != 2:
            break
        was = eng.ld_layout()
        t_r = time.perf_counter()
        eng.run(targets, ld=True)
        eng.sync()
        ms_r = (time.perf_counter() - t_r) * 1e3
        if eng.ld_layout() != was:
            relayout = {"after_runs": k + 2, "run_ms": ms_r}
        if eng.last_count_unit() == 3 and ibd0_pass["after_runs"] is None:
            ibd0_pass = {"after_runs": k + 2, "run_ms": ms_r}
    # clock settling (untimed, before the W warm-up steps of the contract): queued steps for --prewarm-ms of wall time
    eng.set_option("async", 1)
    t_pw = time.perf_counter()
    n_prewarm = 0
    while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
        for _ in range(8):
            step()
        eng.sync()
        n_prewarm += 8
    eng.set_option("async", 0)
    for _ in range(args.warmup):
        step()
    # The timed steps are queued back to back (the engine's "async" option: ibdg_run returns once
    # its kernels are enqueued, like any stream-ordered step loop) and the closing barrier waits
    # for all of them; each step's HIP events are read afterwards (the engine keeps the last 32).
    eng.set_option("async", 0 if args.sync_steps else 1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    dt_host = time.perf_counter() - t0          # host time to queue the steps (reported only)
    barrier()
    dt = time.perf_counter() - t0
    # what the closing barrier itself costs (it is inside the timed region above, after this rank's own steps): the same
    # barrier again with nothing to wait for, best of 5
    barrier_ms = None
    if use_dist:
        b_all = []
        for _ in range(5):
            tb = time.perf_counter()
            barrier()
            b_all.append((time.perf_counter() - tb) * 1e3)
        barrier_ms = min(b_all)
    eng.set_option("async", 0)
    ms_all = [eng.run_ms(b) for b in range(min(args.steps, 32))]
    ms_ld = [m["ld"] for m in ms_all]
    layout_timed = eng.ld_layout()
    count_unit = eng.last_count_unit()

    if args.timed_only:
        if rank == 0:
            print(json.dumps({"metric": "SNP-sites/sec in --LD mode, chr1, 2504-indiv panel", "value": n_cov / (dt / args.steps),
                              "unit": "sites/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ms_per_step": dt / args.steps * 1e3, "ld_launch_ms": float(np.mean(ms_ld)),
                              "ld_layout": layout_timed, "new_individual_per_step": len(turn) > 1,
                              "note": "--timed-only: this rank's clock, no other legs"}))
        eng.close()
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    # the step with the alt alleles recounted inside it (K0 in the timed region; reported beside `value`): queued
    # back to back like the timed steps
    eng.set_option("count_in_run", 1)
    eng.set_option("async", 1)
    for _ in range(10):
        eng.run(targets, ld=True)
    eng.sync()
    n_rc = max(20, min(args.steps, 100))
    t1 = time.perf_counter()
    for _ in range(n_rc):
        eng.run(targets, ld=True)
    eng.sync()
    dt_recount = (time.perf_counter() - t1) / n_rc
    eng.set_option("async", 0)
    alt_in_run_ms = float(np.mean([eng.run_ms(b)["alt_count"] for b in range(min(n_rc, 32))]))
    # k_alt_count alone on the chip: recount inside non-LD runs (nothing on the main stream beside it)
    for _ in range(6):
        eng.run(targets, ld=False)
    alt_ms = float(np.mean([eng.run_ms(b)["alt_count"] for b in range(5)]))
    eng.set_option("count_in_run", 0)
    eng.run(targets, ld=True)
    # the dominant kernel alone: 32 more queued steps timed through the kernel's own dispatch packet
    # (hipExtLaunchKernel start/stop events; slower per step than one event record, hence not in the
    # timed region above)
    ms_kernel = None
    try:
        eng.set_option("dispatch_events", 1)
        eng.set_option("async", 1)
        for _ in range(40):
            step()
        eng.sync()
        ms_kernel = [eng.run_kernel_ms(b) for b in range(32)]
    except ibdgem_amd.EngineError:
        pass                                 # strict kernel: no such figure
    eng.set_option("async", 0)
    eng.set_option("dispatch_events", 0)
    # the same step with every term counted in it (option ibd0_after 0: no pass over the site list is used), same tiles, same queue:
    # what the timed steps would cost without the engine's use of what a site list's comparisons have in common
    all_terms = None
    if count_unit == 3:
        eng.set_option("ibd0_after", 0)
        eng.set_option("async", 1)
        for _ in range(16):
            step()
        eng.sync()
        n_at = max(32, min(args.steps, 200))
        t_at = time.perf_counter()
        for _ in range(n_at):
            step()
        eng.sync()
        at_ms = (time.perf_counter() - t_at) / n_at * 1e3
        eng.set_option("async", 0)
        at_ld = float(np.mean([eng.run_ms(b)["ld"] for b in range(32)]))
        all_terms = {"ms_per_step": at_ms, "ld_launch_ms": at_ld, "count_unit": eng.last_count_unit(), "steps": n_at,
                     "sites_per_s": n_cov / (at_ms * 1e-3),
                     "hbm_frac": algorithmic_bytes_per_site(args.ids, 1) * n_cov / (at_ld * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "note": "the timed steps' queue with option ibd0_after 0: k_ld_popcount counts the IBD0 terms of every background "
                             "individual in every step again (ibdg_last_count_unit 2); same tiles, a new individual per step"}
        eng.set_option("ibd0_after", next((int(kv.split("=")[1]) for kv in args.opt if kv.startswith("ibd0_after=")), 8))
        for _ in range(2):
            step()
        eng.sync()
    # the same step on the panel's own tiles (what the first 15 runs on a site list cost; the timed steps of rounds 1-3):
    # compacted tiles forbidden, a fresh upload, queued steps after the same clock settling
    in_place = None
    if layout_timed == 2 and world == 1:
        eng.set_option("compact_tiles", -1)
        eng.set_option("ibd0_after", 0)       # (... and every run counts the IBD0 terms itself, as a site list's first runs do)
        eng.upload_sites(np.arange(n_rows, dtype=np.uint32), n_ref, n_alt, args.window)
        eng.set_option("async", 1)
        t_pw = time.perf_counter()
        while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
            for _ in range(8):
                eng.run(targets, ld=True)
            eng.sync()
        n_ip = max(32, min(args.steps, 200))
        t_ip = time.perf_counter()
        for _ in range(n_ip):
            eng.run(targets, ld=True)
        eng.sync()
        ip_ms = (time.perf_counter() - t_ip) / n_ip * 1e3
        eng.set_option("async", 0)
        ip_ld = float(np.mean([eng.run_ms(b)["ld"] for b in range(32)]))
        in_place = {"ms_per_step": ip_ms, "ld_launch_ms": ip_ld, "ld_layout": eng.ld_layout(), "steps": n_ip,
                    "sites_per_s": n_cov / (ip_ms * 1e-3),
                    "hbm_frac": algorithmic_bytes_per_site(args.ids, 1) * n_cov / (ip_ld * 1e-3) / 1e9 / HBM_PEAK_GBS}
        eng.set_option("compact_tiles", 0)
        eng.set_option("ibd0_after", next((int(kv.split("=")[1]) for kv in args.opt if kv.startswith("ibd0_after=")), 8))
        eng.upload_sites(np.arange(n_rows, dtype=np.uint32), n_ref, n_alt, args.window)
    # the other clocks of one comparison (not `value`): upload of its rows, the survey's engine clock, results to host
    if want_second:
        eng2 = second_context(panel_kept)
        del panel_kept
        torch.cuda.empty_cache()
    up, engine_clock, d2h = upload_and_engine_clocks(torch, eng, ibdgem_amd, n_ref, n_alt, args.window, targets, n_cov, eng2)
    if eng2 is not None:
        eng2.close()
        eng2 = None
    up["first_call_ms"] = first_upload_ms
    # BASELINE.json configs[1]'s shape on this rank's rows (not `value`): the non-LD step (per-site values + window products)
    non_ld = None
    if world == 1 and not args.no_many:
        eng.set_option("async", 1)
        # clock settling as for the timed steps (--prewarm-ms of wall time, untimed): the legs before this one leave the
        # GPU idle between host-side waits, and 100 steps of 0.04 ms are over before its clock is back up
        t_pw = time.perf_counter()
        while True:
            for _ in range(100):
                eng.run(targets, ld=False)
            eng.sync()
            if (time.perf_counter() - t_pw) * 1e3 >= args.prewarm_ms:
                break
        t0 = time.perf_counter()
        for _ in range(200):
            eng.run(targets, ld=False)
        eng.sync()
        nl_ms = (time.perf_counter() - t0) / 200 * 1e3
        eng.set_option("async", 0)
        nl_k = float(np.mean([eng.run_ms(i)["rows"] for i in range(16)]))
        nl_bytes = 4 + 0.25 + 24.24          # read counts + target alleles + three doubles (+ window results)
        non_ld = {"ms_per_step": nl_ms, "rows_per_s": n_rows / (nl_ms * 1e-3), "k_rows_windows_ms": nl_k,
                  "bytes_per_row": nl_bytes,
                  "hbm_frac": nl_bytes * n_rows / (nl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "kernel_hbm_frac": nl_bytes * n_rows / (nl_k * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "note": "non-LD comparison (per-row LIBD0/1/2 + window products, BASELINE.json configs[1] at this row count), "
                          "queued steps, host wall clock; one kernel (k_rows_windows: a wave per pair of windows computes their rows' "
                          "values, stores them and multiplies them up -- nothing is read back; the AF column is made by "
                          "ibdg_get_site_af when asked for, not by the run); hbm_frac on the step clock, kernel_hbm_frac on "
                          "the kernel's own events"}
    # BASELINE.json configs[4]'s shape over N ranks: every rank runs ITS window range against all the comparison individuals
    # (one ibdg_run; windows are independent, src/ibdgem.c:558-570, and so are comparison individuals, :522 -- a 2-D split
    # would change nothing per rank); device time and wall time per rank, gathered below
    many_rank = None
    if world > 1 and not args.no_many:
        many_tr = [(args.target + 5 * i) % args.ids for i in range(args.many_targets)]
        eng.set_option("site_results", 0)
        eng.run(many_tr, ld=True)
        eng.sync()
        best_dev, best_wall = None, None
        for _ in range(3):
            barrier()
            tw = time.perf_counter()
            eng.run(many_tr, ld=True)
            eng.sync()
            wall = (time.perf_counter() - tw) * 1e3
            dev_ms = eng.last_run_ms()["total"]
            best_dev = dev_ms if best_dev is None else min(best_dev, dev_ms)
            best_wall = wall if best_wall is None else min(best_wall, wall)
        eng.set_option("site_results", 1)
        many_rank = {"comparison_individuals": len(many_tr), "device_ms": best_dev, "wall_ms": best_wall,
                     "ld_layout": eng.ld_layout(), "windowed_sites": int(n_cov)}
    many = None
    if world == 1 and not args.no_many:
        many_t = [(args.target + 5 * i) % args.ids for i in range(args.many_targets)]
        eng.set_option("compact_tiles", -1)         # this part: the panel's own tiles, however many runs it takes (the upload before it left them)
        eng.run(many_t, ld=True)
        best = None
        for _ in range(3):
            eng.run(many_t, ld=True)
            ms = eng.last_run_ms()
            best = ms if best is None or ms["total"] < best["total"] else best
        T = len(many_t)
        b_site_T = algorithmic_bytes_per_site(args.ids, T)          # N/4 + 4 + T x 24.24 B per windowed site
        achieved_T = b_site_T * n_cov / (best["ld"] * 1e-3) / 1e9
        # the same run with no per-row results kept (option site_results 0: the host program's --summary-only)
        eng.set_option("site_results", 0)
        eng.run(many_t, ld=True)
        best0 = None
        for _ in range(3):
            eng.run(many_t, ld=True)
            ms = eng.last_run_ms()
            best0 = ms if best0 is None or ms["total"] < best0["total"] else best0
        eng.set_option("site_results", 1)
        # the same run from the compacted tiles of the site list (the engine switches by itself from
        # "compact_targets" = 256 comparison individuals; forced here): the re-layout is paid once, inside the first run
        eng.set_option("compact_tiles", 0)
        eng.set_option("compact_targets", 1)
        eng.upload_sites(np.arange(n_rows, dtype=np.uint32), n_ref, n_alt, args.window)
        eng.sync()
        t_c = time.perf_counter()
        eng.run(many_t, ld=True)
        first_c_ms = (time.perf_counter() - t_c) * 1e3
        layout_c = eng.ld_layout()
        best_c = None
        for _ in range(3):
            eng.run(many_t, ld=True)
            ms = eng.last_run_ms()
            best_c = ms if best_c is None or ms["total"] < best_c["total"] else best_c
        eng.set_option("compact_targets", 256)
        eng.upload_sites(np.arange(n_rows, dtype=np.uint32), n_ref, n_alt, args.window)      # back to the panel's own tiles
        b_site_T0 = args.ids / 4.0 + 4.0 + T * 0.24                  # ... without the 24 B per row and individual
        pmc = None
        pdir = os.path.join(REPO, "profiles")
        for fn in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
            if fn.endswith("_mfma_pmc.json"):
                with open(os.path.join(pdir, fn)) as fh:
                    j = json.load(fh)
                    k = (j.get("kernels") or j).get("ibdg::k_ld_mfma")       # (summarised or raw per-kernel averages)
                if k:
                    pmc = {"profile": fn, "VALUBusy_pct": k.get("VALUBusy"), "LdsUtil_pct": k.get("LdsUtil"),
                           "MfmaUtil_pct": k.get("MfmaUtil"), "valu_instructions_per_launch": k.get("SQ_INSTS_VALU"),
                           "lds_instructions_per_launch": k.get("SQ_INSTS_LDS")}
        many = {"comparison_individuals": T, "run_ms": best["total"], "ms_per_individual": best["total"] / T,
                "site_individual_pairs_per_s": n_cov * T / (best["total"] * 1e-3),
                "ld_launches_ms": best["ld"], "rows_kernel_ms": best["rows"],
                "roofline": {"bound": "hbm", "kernel": "k_ld_mfma (groups of 15)", "bytes_per_site": b_site_T,
                             "sites_per_launch": n_cov, "kernel_ms": best["ld"], "achieved": achieved_T, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": achieved_T / HBM_PEAK_GBS,
                             "busy_shares_from_profile": pmc,
                             "busy_shares_note": "replayed from the committed counter profile named in it, not measured by this run",
                             "note": "bytes_per_site = N/4 + 4 + T x 24.24 (SURVEY.md s8(d): the panel row once per launch, 24 B "
                                     "of per-row output per individual); kernel_ms = the --LD launches of the run (k_win_target_g, "
                                     "k_win_slot_g, k_ld_mfma, k_ld_finalize for all groups) from the engine's events.  This path is "
                                     "bound by vector-ALU issue and the LDS port in its window ends (four table products per "
                                     "background individual, comparison individual and window), far from the HBM roofline: the "
                                     "busy shares are those of the committed counter profile"},
                "without_per_row_results": {"run_ms": best0["total"], "ms_per_individual": best0["total"] / T,
                                            "bytes_per_site": b_site_T0,
                                            "hbm_frac": b_site_T0 * n_cov / (best0["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                            "device_memory_for_results_bytes": int(T * eng.n_windows * 24),
                                            "note": "option site_results 0: no T x rows x 24 B array, no per-row stores; LIBD2 of "
                                                    "the windows from the IBD2 pick of every row (k_rows_windows<false>)"},
                "from_compacted_tiles": {"ld_layout": layout_c, "run_ms": best_c["total"], "ms_per_individual": best_c["total"] / T,
                                         "ld_launches_ms": best_c["ld"],
                                         "first_run_wall_ms_with_the_relayout_inside": first_c_ms,
                                         "relayout_pays_from_individuals": (
                                             (first_c_ms - best_c["total"]) / max(1e-9, (best["total"] - best_c["total"]) / T)
                                             if best["total"] > best_c["total"] else None),
                                         "note": "rows with reads gathered back to back and transposed into tiles: 4.1 segments and 3.1 tile "
                                                 "words per window of 100 rows instead of 4.6 and 3.6, no rows without reads streamed"},
                "note": "one ibdg_run over that many comparison individuals against the resident panel (device time, best of 3): "
                        "groups of 15 through k_ld_mfma -- the sums that depend on the comparison individual as integer matrix "
                        "products (DESIGN.md s4.2); per-site values and window products of all of them included"}
    sparse = None
    if world == 1 and not args.no_many:
        sparse = sparse_pileup_clocks(torch, eng, ibdgem_amd, n_ref, n_alt, args.window, args.target, many_t, args.seed)
        eng.upload_sites(np.arange(n_rows, dtype=np.uint32), n_ref, n_alt, args.window)
    eng.run(targets, ld=True)
    win_full = eng.window_ll(0) if rank == 0 else None
    ld_variant = eng.last_ld_variant()

    tot = torch.tensor([dt, float(n_cov), float(n_rows)], dtype=torch.float64,
                       device=dev if backend == "nccl" else "cpu")
    if use_dist:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt_max, cov_total, rows_total = float(mx[0]), float(sm[1]), float(sm[2])
    else:
        dt_max, cov_total, rows_total = dt, float(n_cov), float(n_rows)

    mine = {"rank": rank, "rows": int(n_rows), "windowed_sites": int(n_cov), "windows": int(eng.n_windows),
            "ms_per_step": dt / args.steps * 1e3, "ld_launch_ms": float(np.mean(ms_ld)),
            "ld_launch_ms_min": float(np.min(ms_ld)),       # the fastest of those steps on the device's own clock
            "step_device_ms": float(np.mean([m["total"] for m in ms_all])),      # the engine's own events: both streams of a step
            "many_comparison_individuals": many_rank,
            "upload_sites_ms": up["pageable_ms"], "engine_clock_ms": engine_clock["ms"],
            "host_queue_ms_per_step": dt_host / args.steps * 1e3}
    per_rank = [mine]
    if use_dist:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)

    if rank == 0:
        ms_step = dt_max / args.steps * 1e3
        value = cov_total / (dt_max / args.steps)
        b_site = algorithmic_bytes_per_site(args.ids, len(targets))
        ld_ms = float(np.mean(ms_ld))
        # the roofline entry divides by the --LD launch time OF THE TIMED STEPS (HIP events on the engine's stream, mean over
        # the last <= 32 of them): everything a step's --LD part launches, the dominant kernel's fused finalising
        # workgroups included.  The kernel's own dispatch time (32 further steps of the same form) is kept beside it.
        dom_ms = ld_ms
        achieved = b_site * n_cov / (dom_ms * 1e-3) / 1e9
        kern = {k: float(np.mean([m[k] for m in ms_all])) for k in ms_all[0]}
        out = {
            "metric": "SNP-sites/sec in --LD mode, chr1, 2504-indiv panel",
            "value": value, "unit": "sites/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "clock": "step", "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"--LD, {L} SNP rows (synthetic chr1), {args.ids}-individual phased panel, "
                                   f"window {args.window}, 1 comparison individual (BASELINE.json configs[3])",
                       "rows": int(rows_total), "windowed_sites": int(cov_total), "n_ids": args.ids,
                       "windows_rank0": int(eng.n_windows),
                       "segments_rank0": n_segments_compact if layout_timed == 2 else n_segments_in_place,
                       "segments_rank0_in_place": n_segments_in_place,
                       "window": args.window, "targets": len(targets), "epsilon": 0.02, "max_cov": 20,
                       "tiles": ("compacted tiles of the site list, its rows with reads back to back (ld_layout 2): the engine re-laid the site list "
                                 f"out by itself during run {relayout['after_runs']} on it, before the warm-up steps; "
                                 "`in_place_tiles` is the same step before that") if layout_timed == 2 else
                                "the panel's own tiles (ld_layout 1)" if layout_timed == 1 else "none (strict kernel)",
                       "sharding": f"{world} contiguous window ranges, no collective on the data path"},
            "roofline": {"bound": "hbm", "kernel": ("k_ld_popcount, IBD1 form (a word's four table exponents by one v_mfma_scale_f32_16x16x128_f8f6f4; "
                                                    "IBD0 from one pass per site list)" if count_unit == 3 else
                                                    "k_ld_popcount (a word's weighted sums by one v_mfma_scale_f32_16x16x128_f8f6f4)"
                                                    if count_unit == 2 else "k_ld_popcount") if ld_variant == 2 else "k_ld_window",
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_bytes(args, world)[0],
                         "traffic_from_profile": traffic_bytes(args, world)[1],
                         "traffic_note": "replayed, not measured by this run: counter bytes per launch of the committed PMC passes "
                                         "of this workload (the file named in traffic_from_profile); counters cannot be read from "
                                         "inside the process",
                         "bytes_per_site": b_site, "sites_per_launch": n_cov, "kernel_ms": dom_ms,
                         "kernel_ms_note": "achieved = bytes_per_site x sites_per_launch / kernel_ms; kernel_ms = launch_ms",
                         "launch_ms": ld_ms,
                         "launch_ms_note": "HIP events on the engine's stream around the --LD launches of a step "
                                           "(k_target_weights + k_win_target_mx + k_ld_popcount with the previous step's "
                                           "finalising workgroups in it; with --same-target k_ld_popcount only), mean over the "
                                           f"last {len(ms_ld)} timed steps",
                         "dominant_kernel_only_ms": float(np.mean(ms_kernel)) if ms_kernel else None,
                         "dominant_kernel_only_note": "k_ld_popcount alone (the same fused form as in the timed steps), start/stop "
                                                      "events of its own dispatch packet, 32 extra steps after the timed region: "
                                                      "what rocprofv3 --kernel-trace --stats of --timed-only shows as its average",
                         "valu": valu_roofline(float(np.mean(ms_kernel)) if ms_kernel else ld_ms, int(eng.n_windows),
                                               (args.ids + 63) // 64) if world == 1 else None},
            "kernel_ms": kern,
            "ld_layout": layout_timed,
            "new_individual_per_step": len(turn) > 1,
            "relayout": dict(relayout, note="the run on this site list during which the engine gathered its rows with reads into compacted "
                             "tiles (k_gather_transpose32 + the segments again), host wall clock of that run, "
                             "untimed; the rule: the runs on one upload add up, a group of the matrix-core kernel as 20, an individual "
                             "of the counting kernels as 12 (16 with mx_counts 0), against compact_targets = 256 (DESIGN s3, s4.1) -- "
                             "null: no re-layout happened"),
            "ibd0_pass": dict(ibd0_pass, count_unit=count_unit,
                              note="the run on this upload during which the engine made ONE pass of the counting kernel that keeps every "
                                   "background individual's own IBD0 product per window (src/ibdgem.c:715, :743: it does not depend on the "
                                   "comparison individual, whose only trace in that sum is its own exclusion, :714), host wall clock of "
                                   "that run, untimed; the rule: eight single runs on one upload and background (option ibd0_after); "
                                   "count_unit 3 = the timed steps count the IBD1 sums only and their finalising step takes IBD0 from "
                                   "the pass, in the additions of a run that counts everything: same bits (tests/test_gpu_parity.py); "
                                   "`in_place_tiles` is a step that counts everything, on the panel's own tiles"),
            "all_terms_in_every_step": all_terms,
            "in_place_tiles": in_place,
            "prewarm": {"ms": args.prewarm_ms, "steps": n_prewarm,
                        "note": "untimed steps before the warm-up steps so that the clocks have settled when they start"},
            "host_queue_ms_per_step": dt_host / args.steps * 1e3,
            "alt_count_ms": alt_ms,
            "alt_count_note": "k_alt_count alone on the chip (2.56 GB panel); inside an --LD run it is throttled to 4 single-wave "
                              "workgroups per CU so that the --LD workgroups keep their wave slots: alt_count_in_ld_run_ms",
            "alt_count_in_ld_run_ms": alt_in_run_ms,
            "upload_sites_ms": up["pageable_ms"],
            "upload_sites": up,
            "engine_clock": engine_clock,
            "results_to_host": d2h,
            "per_rank": per_rank,
            "barrier_ms": barrier_ms,
            "barrier_note": None if not use_dist else "the closing barrier of the timed region (dist.barrier + device sync) with "
                            "nothing left to wait for, best of 5: what it adds to `steps` x ms_per_step",
            "value_with_recount": n_cov / dt_recount if world == 1 else None,
            "many_comparison_individuals": many,
            "sparse_pileup": sparse,
            "non_ld": non_ld,
            "rows_per_s_all_processed": rows_total / (dt_max / args.steps),
        }
        if many:
            many["vs_one_per_run"] = ms_step / many["ms_per_individual"]        # against the step of one comparison individual
        if world > 1 and per_rank[0].get("many_comparison_individuals"):
            pr = [p["many_comparison_individuals"] for p in per_rank]
            T = pr[0]["comparison_individuals"]
            wall = max(p["wall_ms"] for p in pr)
            out["many_comparison_individuals"] = {
                "comparison_individuals": T, "ranks": world,
                "wall_ms_max_over_ranks": wall, "device_ms_max_over_ranks": max(p["device_ms"] for p in pr),
                "ms_per_individual": wall / T,
                "site_individual_pairs_per_s": sum(p["windowed_sites"] for p in pr) * T / (wall * 1e-3),
                "per_rank": pr,
                "note": "BASELINE.json configs[4]'s shape over the ranks: each rank its contiguous window range x all comparison "
                        "individuals in one ibdg_run (no per-row results kept: window tables only), ranks started together "
                        "(barrier), wall and device time per rank, best of 3; the job's rate = all (site, individual) pairs / the "
                        "slowest rank's wall time"}
        if not args.no_cpu_baseline and world == 1:       # the CPU leg is timed at N=1 only
            s = sample_rows
            eng.upload_sites(np.arange(s, dtype=np.uint32), n_ref[:s], n_alt[:s], args.window)
            eng.run(targets, ld=True)
            out["cpu_baseline"] = cpu_baseline(sample_words, n_ref[:s], n_alt[:s], args.ids, args.target,
                                               args.window, eng.window_ll(0))
            cb = out["cpu_baseline"]
            # like for like: the --LD launches against the reference's LD stage alone (its LD run minus its non-LD
            # run on the same text); and the whole comparison (upload of its rows + kernels + window results to the
            # host = engine_clock) against the reference's whole run.  The step-vs-whole-run figure compares
            # different scopes and is kept under its own name.
            if cb.get("ld_stage_only_sites_per_s"):
                out["ld_kernels_vs_reference_ld_stage"] = (n_cov / (ld_ms * 1e-3)) / cb["ld_stage_only_sites_per_s"]
            out["engine_clock_vs_reference_end_to_end"] = engine_clock["sites_per_s"] / cb["value"]
            out["step_vs_reference_end_to_end"] = value / cb["value"]
        if words_all is not None:
            try:
                warm, cold = end_to_end_clocks(words_all, n_ref, n_alt, args.ids, args.target, args.window, args.cold_rows,
                                               win_full)
            except Exception as e:                     # the bench line survives a failing host-program leg
                warm, cold = {"error": repr(e)}, None
            out["warm_e2e"] = warm
            out["cold_e2e"] = cold
            if isinstance(warm, dict) and "summary_only_s" in warm and "cpu_baseline" in out:
                out["warm_e2e_vs_reference_end_to_end"] = (n_cov / warm["summary_only_s"]) / out["cpu_baseline"]["value"]
        print(compact_line(out, write_detail(out, world)), flush=True)
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
