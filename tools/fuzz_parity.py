"""Randomised parity sweep (not part of the pytest tiers): many small random configurations of the
engine against the oracle in one process -- panel width, rows, window, error rate, max coverage,
depth, sparsity of the pileup, background multiplicities, -N exclusion, batches of comparison
individuals (the counting kernels and the matrix-core kernel), launch geometry options.  Prints one line per failure and a summary.

    python tools/fuzz_parity.py [n_cases] [seed]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import ibdgem_amd
from ibdgem_amd import engine as E
import oracle_lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
orc = oracle_lib.Oracle(os.path.join(REPO, "oracle", "liboracle.so"))
TINY = 1e-290
bad = 0
t0 = time.time()
for case in range(n_cases):
    N = int(rng.choice([1, 2, 3, 31, 64, 65, 127, 128, 129, 200, 513, 700]))
    L = int(rng.integers(1, 1500))
    W = int(rng.choice([2, 3, 7, 33, 64, 100, 257]))
    eps = float(rng.choice([0.02, 0.001, 0.1, 0.3, 0.49]))
    M = int(rng.choice([1, 3, 7, 20, 31, 50]))
    cov = float(rng.choice([0.3, 1.0, 2.0, 6.0, 15.0]))
    f = rng.beta(0.3, 1.0, size=L).clip(1e-3, 0.999)
    alle = (rng.random((L, 2 * N)) < f[:, None]).astype(np.uint8)
    c = np.minimum(rng.poisson(cov, size=L), M)
    na = rng.binomial(c, f).astype(np.uint8)
    nr = (c - na).astype(np.uint8)
    keep = np.sort(rng.choice(L, size=max(1, int(L * rng.choice([1.0, 1.0, 0.5, 0.2, 0.05, 0.02, 0.005]))), replace=False))
    if rng.random() < 0.1:
        keep = rng.permutation(keep)                # rows out of file order: the compacted tiles (or the strict kernel)
    T = int(rng.choice([1, 1, 2, 4, 5, 9, 8, 15, 16, 23, 31, 40, 70]))     # 5 and more: the matrix-core kernel (k_ld_mfma); 70: more than the runs that prepare ahead hold
    T = min(T, N)
    targets = [int(t) for t in rng.choice(N, size=T, replace=False)]
    bg = None if rng.random() < 0.5 else rng.integers(0, 3, size=N).astype(np.uint8)
    pu = -1 if rng.random() < 0.6 else int(rng.integers(0, N))
    variant = int(rng.choice([0, 0, 1, 2, 3]))
    order = None
    if variant == 3 and bg is not None and rng.random() < 0.7:
        order = rng.permutation(np.repeat(np.arange(N), bg))        # the -B list in some file order
    opts = {}
    tiles = int(rng.choice([0, 0, 1, -1]))         # tiles of the exponent-counting kernels: auto, compacted, the panel's own
    if rng.random() < 0.5:
        opts = {"ring_slots": int(rng.choice([2, 3, 4, 8])), "windows_per_wave": int(rng.choice([1, 2, 5, 16, 64])),
                "guided_runs": int(rng.choice([0, 1, 4, 16])), "multi_target": int(rng.choice([0, 1])),
                "mfma_targets": int(rng.choice([0, 1, 1])), "mfma_min": int(rng.choice([1, 2, 8, 15])),
                "mx_counts": int(rng.choice([0, 1, 1]))}
    desc = f"case {case}: N={N} L={L} keep={len(keep)} W={W} eps={eps} M={M} cov={cov} T={T} bg={'y' if bg is not None else 'n'} pu={pu} variant={variant} tiles={tiles} {opts}"
    if case >= int(os.environ.get("FUZZ_PRINT_FROM", "1000000000")):     # (to name the case a GPU fault ends the process in)
        print("RUNNING", desc, flush=True)
    try:
        with E.Engine(0, eps, M) as eng:
            for k, v in opts.items():
                eng.set_option(k, v)
            eng.set_option("ld_variant", variant)
            eng.set_option("compact_tiles", tiles)
            more = {"compact_targets": int(rng.choice([3, 8, 96])), "compact_align": int(rng.choice([1, 1, 1, 32, 4, 16])),
                    # single individuals with their IBD0 terms from one pass over the site list: from the first run, after a few, never
                    "ibd0_after": int(rng.choice([1, 1, 2, 3, 8, 0]))}
            for k, v in more.items():
                eng.set_option(k, v)
            if case >= int(os.environ.get("FUZZ_PRINT_FROM", "1000000000")):
                print("   ", more, flush=True)
            eng.upload_panel(E.pack_alleles_fast(alle), N)
            eng.upload_sites(keep, nr[keep], na[keep], W)
            eng.set_background_order(order)
            # queued runs (option async) of OTHER comparison individuals in front of the one that is checked: the ring of
            # per-individual buffers, the preparation on the third stream, a finalising step left to the next launch
            queued = int(rng.integers(0, 7)) if rng.random() < 0.5 else 0
            if queued:
                eng.set_option("async", 1)
                eng.set_option("prep_ahead", int(rng.choice([1, 1, 0])))
                eng.set_option("end_in_dispatch", int(rng.choice([1, 1, 0])))
                for _ in range(queued):
                    other = [int(t) for t in rng.choice(N, size=int(rng.choice([1, 1, 1, T, min(N, T + 1)])), replace=False)]
                    if case >= int(os.environ.get("FUZZ_PRINT_FROM", "1000000000")):
                        print("    queued run of", len(other), "individuals", flush=True)
                    try:
                        eng.run(other, ld=bool(rng.random() < 0.9), bg_count=bg if order is not None or rng.random() < 0.8 else None, pu_id=pu)
                        if os.environ.get("FUZZ_SYNC") and case >= int(os.environ.get("FUZZ_PRINT_FROM", "1000000000")):
                            eng.sync()
                            print("      ... done, count unit", eng.last_count_unit(), "layout", eng.ld_layout(), flush=True)
                    except E.EngineError as e:
                        if variant == 2 and "not applicable" in str(e):
                            break
                        raise
                    if rng.random() < 0.15:
                        eng.window_ll(0)
            try:
                eng.run(targets, ld=True, bg_count=bg, pu_id=pu)
                if queued and rng.random() < 0.5:
                    eng.run(targets, ld=True, bg_count=bg, pu_id=pu)
            except E.EngineError as e:
                if variant == 2 and "not applicable" in str(e):
                    continue                      # forced exponent counting where the table is clamped etc.
                raise
            refids = order if order is not None else (None if bg is None else np.repeat(np.arange(N), bg))
            for i, t in enumerate(targets):
                res = orc.compare(alle[keep], nr[keep], na[keep], t, window=W, ld=True, eps=eps, max_cov=M,
                                  refids=refids, pu_id=pu)
                site, win = eng.site_ll(i), eng.window_ll(i)
                ok = (site.view(np.uint64) == res["site"].view(np.uint64)).all()
                ok = ok and len(win) == len(res["win"])
                if ok and len(win):
                    ok = (win[:, 2].view(np.uint64) == res["win"][:, 2].view(np.uint64)).all()
                    g, w_ = win[:, :2], res["win"][:, :2]
                    nan_ok = (np.isnan(g) == np.isnan(w_)).all()
                    fin = ~np.isnan(w_)
                    tiny = fin & (np.abs(w_) < TINY)
                    big = fin & ~tiny
                    rel = np.abs(g[big] - w_[big]) / np.abs(w_[big]) if big.any() else np.zeros(0)
                    ok = ok and nan_ok and (np.abs(g[tiny]) < TINY).all() and (rel.size == 0 or rel.max() <= 1e-10)
                    if variant == 3:          # reference order: every bit
                        ok = ok and ((g.view(np.uint64) == w_.view(np.uint64)) | (np.isnan(g) & np.isnan(w_))).all()
                if not ok:
                    bad += 1
                    print("MISMATCH", desc, "target", t, flush=True)
                    if os.environ.get("FUZZ_DETAIL"):          # which part: per-row values, window count, LIBD2, --LD columns
                        sd = np.flatnonzero((site.view(np.uint64) != res["site"].view(np.uint64)).any(axis=1))
                        print("   rows that differ:", len(sd), sd[:8], "| windows", len(win), "vs", len(res["win"]), flush=True)
                        if len(win) == len(res["win"]) and len(win):
                            wd = np.flatnonzero(win[:, 2].view(np.uint64) != res["win"][:, 2].view(np.uint64))
                            ld = np.flatnonzero(~np.isclose(win[:, :2], res["win"][:, :2], rtol=1e-10, atol=0, equal_nan=True).all(axis=1))
                            print("   LIBD2 windows that differ:", len(wd), wd[:8], "| --LD windows:", len(ld), ld[:8], flush=True)
                            if len(ld):
                                print("   first:", win[ld[0]], res["win"][ld[0]], flush=True)
                    break
    except Exception as e:                        # noqa: BLE001
        bad += 1
        print("ERROR", desc, repr(e)[:200], flush=True)
    if case % 25 == 24:
        print(f"... {case + 1} cases, {bad} bad, {time.time() - t0:.0f}s", flush=True)
print(f"fuzz: {n_cases} cases, {bad} failures, {time.time() - t0:.0f}s")
sys.exit(1 if bad else 0)
