"""GPU parity: the HIP path, called through the C ABI, against the oracle and
the committed golden vectors.

Bars (BASELINE.json north_star):
  * integer results (alt counts, window boundaries, NUM_SITES): bit-exact;
  * per-site LIBD0/1/2 (tab columns): bit-exact doubles (same operation order,
    host-built pow tables, no FMA contraction);
  * window LIBD2 and non-LD LIBD0/1: bit-exact (sequential product, same order);
  * --LD window LIBD0/LIBD1: relative 1e-10 wherever |ref| >= 1e-290, "both
    below 1e-290" otherwise (the sum over the background is a tree on the
    GPU and a serial loop in the reference; observed differences are ~1e-15).
"""
import os
import sys

import numpy as np
import pytest

import golden_io as G
from ibdgem_amd import engine as E

pytestmark = pytest.mark.gpu

LD_RTOL = 1e-10
TINY = 1e-290


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def assert_bits(got, want, what):
    got, want = np.asarray(got, float), np.asarray(want, float)
    assert got.shape == want.shape, what
    same = (bits(got) == bits(want)) | (np.isnan(got) & np.isnan(want))
    if not same.all():
        i = tuple(np.argwhere(~same)[0])
        raise AssertionError(f"{what}: {(~same).sum()}/{same.size} differ, first at {i}: "
                             f"{got[i]!r} ({got[i].hex()}) vs {want[i]!r} ({want[i].hex()})")


def assert_ld_close(got, want, what):
    got, want = np.asarray(got, float), np.asarray(want, float)
    assert got.shape == want.shape, what
    nan = np.isnan(want)
    assert (np.isnan(got) == nan).all(), f"{what}: NaN pattern"
    tiny = (np.abs(want) < TINY) & ~nan
    assert (np.abs(got[tiny]) < TINY).all(), f"{what}: tiny values"
    big = ~tiny & ~nan
    rel = np.abs(got[big] - want[big]) / np.abs(want[big])
    assert rel.size == 0 or rel.max() <= LD_RTOL, f"{what}: max rel err {rel.max():.3e}"
    return 0.0 if rel.size == 0 else float(rel.max())


def bg_counts(refids, n_ids):
    if refids is None:
        return None
    c = np.zeros(n_ids, dtype=np.uint8)
    for k in refids:
        c[k] += 1
    return c


@pytest.fixture(scope="module")
def eng():
    with E.Engine(0, 0.02, 20) as e:
        yield e


# --------------------------------------------------------------------------- fixtures of the reference
def test_reference_fixture_files(eng):
    panel = G.load_fixture_panel()
    eng.upload_panel(E.pack_alleles(panel.alleles), len(panel.names))
    for k in (1, 2, 3):
        tabs = {t: G.fixture_outputs(f"sample{k}", f"sample{t}") for t in (1, 2, 3)}
        tab0 = tabs[1][0]
        rows = np.array([panel.row_of_pos[int(p)] for p in tab0.pos], dtype=np.uint32)
        eng.upload_sites(rows, tab0.n_ref, tab0.n_alt, 100)
        eng.run([0, 1, 2], ld=False)
        af = eng.site_af()
        first, last, ncov = eng.windows()
        for t in (1, 2, 3):
            tab, summ = tabs[t]
            assert (tab.pos == tab0.pos).all()
            ll = eng.site_ll(t - 1)
            win = eng.window_ll(t - 1)
            for i in range(len(rows)):
                assert "%f" % af[i] == tab.af_txt[i]
                assert ["%e" % v for v in ll[i]] == tab.ll_txt[i], (k, t, i)
            assert len(win) == len(summ.ll)
            for w in range(len(win)):
                assert ["%e" % v for v in win[w]] == summ.ll_txt[w], (k, t, w)
            assert (ncov == summ.nsites).all()
            assert (tab.pos[first] == summ.start).all() and (tab.pos[last] == summ.end).all()


# --------------------------------------------------------------------------- 17-digit reference outputs
def _syn_cases():
    return [(tag, case) for tag in ("synA", "synB") for case in G.cases(tag)["cases"]]


VARIANTS = [1, 2]      # 1 = strict fp64 products, 2 = exponent counting (see include/ibdgem_hip.h)


@pytest.mark.parametrize("variant", VARIANTS + [3])
@pytest.mark.parametrize("tag,case", _syn_cases())
def test_reference_17digit_outputs(tag, case, variant, oracle):
    flags, panel, names, refids, pu_id, per_target = G.case_setup(tag, case)
    n_ids = len(panel.names)
    with E.Engine(0, flags["eps"], flags["max_cov"]) as eng:
        eng.set_option("ld_variant", variant)
        eng.upload_panel(E.pack_alleles_fast(panel.alleles), n_ids)
        cnt = eng.alt_counts(0, len(panel.pos))
        assert (cnt == panel.alleles.sum(axis=1)).all()
        for name in names:
            tab, summ, alle, fo, rows = per_target[name]
            t = panel.index(name)
            eng.upload_sites(rows, tab.n_ref, tab.n_alt, flags["window"], f_override=fo)
            eng.set_background_order(refids if variant == 3 else None)
            eng.run([t], ld=flags["ld"], bg_count=bg_counts(refids, n_ids), pu_id=pu_id)
            assert eng.n_sites == len(tab.pos) == tab.processed
            assert eng.last_ld_variant() == (variant if flags["ld"] else 0)
            assert_bits(eng.site_ll(0), tab.ll, f"{tag}/{case}/{name} per-site")
            af = eng.site_af()
            assert ["%f" % x for x in af] == tab.af_txt
            win = eng.window_ll(0)
            assert len(win) == len(summ.ll)
            first, last, ncov = eng.windows()
            assert (ncov == summ.nsites).all()
            if len(win):
                assert (tab.pos[first] == summ.start).all() and (tab.pos[last] == summ.end).all()
            assert_bits(win[:, 2], summ.ll[:, 2], f"{tag}/{case}/{name} window LIBD2")
            if flags["ld"] and variant == 3:     # reference order: the reference's own 17-digit values, bit for bit
                assert_bits(win, summ.ll, f"{tag}/{case}/{name} LD window, reference order")
            elif flags["ld"]:
                assert_ld_close(win[:, :2], summ.ll[:, :2], f"{tag}/{case}/{name} LD window")
            else:
                assert_bits(win, summ.ll, f"{tag}/{case}/{name} window")
            # and the oracle says the same as the reference's file (sanity of the checker)
            res = oracle.compare(alle, tab.n_ref, tab.n_alt, t, window=flags["window"], eps=flags["eps"],
                                 max_cov=flags["max_cov"], refids=refids, pu_id=pu_id, ld=flags["ld"],
                                 f_override=fo)
            assert_bits(res["win"], summ.ll, "oracle vs golden")


# --------------------------------------------------------------------------- seeded random vs oracle
def synth(seed, L, N, cov_mean=2.0):
    rng = np.random.default_rng(seed)
    f = np.clip(rng.beta(0.3, 1.0, size=L), 1e-3, 0.999)
    alle = (rng.random((L, 2 * N)) < f[:, None]).astype(np.uint8)
    cov = np.minimum(rng.poisson(cov_mean, size=L), 20)
    n_alt = rng.binomial(cov, f)
    return alle, (cov - n_alt).astype(np.uint8), n_alt.astype(np.uint8)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("N,L,W", [(3, 257, 100), (64, 500, 100), (65, 300, 7), (100, 1500, 100),
                                   (320, 700, 64), (321, 400, 100), (2504, 1200, 100), (700, 350, 350),
                                   (1100, 300, 50), (5000, 260, 100)])      # 18 chunks: 3 groups of 6 waves; 79: 10 of 8
def test_random_panels_against_oracle(oracle, N, L, W, variant):
    alle, nr, na = synth(1000 + N, L, N)
    targets = sorted({0, N // 2, N - 1})
    with E.Engine() as eng:
        eng.set_option("ld_variant", variant)
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, W)
        for pu in (-1, targets[-1]):
            eng.run(targets, ld=True, pu_id=pu)
            for i, t in enumerate(targets):
                res = oracle.compare(alle, nr, na, t, window=W, ld=True, pu_id=pu)
                assert_bits(eng.site_ll(i), res["site"], f"N={N} t={t} site")
                assert_bits(eng.site_af(), res["af"], "af")
                win = eng.window_ll(i)
                assert_bits(win[:, 2], res["win"][:, 2], "LIBD2")
                assert_ld_close(win[:, :2], res["win"][:, :2], f"N={N} t={t} pu={pu} LD")
                first, last, ncov = eng.windows()
                assert (first == res["first"]).all() and (last == res["last"]).all()
                assert (ncov == res["nsites"]).all()
        eng.run(targets[:1], ld=False)
        res = oracle.compare(alle, nr, na, targets[0], window=W, ld=False)
        assert_bits(eng.window_ll(0), res["win"], "non-LD windows")


@pytest.mark.parametrize("variant", VARIANTS)
def test_row_indirection_and_zero_coverage(oracle, variant):
    """Sites reference panel rows through row_index (filtered rows are never touched)."""
    N, Lp = 130, 900
    alle, nr, na = synth(7, Lp, N)
    rng = np.random.default_rng(8)
    keep = np.sort(rng.choice(Lp, size=500, replace=False))
    nr, na = nr[keep].copy(), na[keep].copy()
    nr[::7] = 0
    na[::7] = 0                                     # zero-coverage rows: printed, not windowed
    with E.Engine() as eng:
        eng.set_option("ld_variant", variant)
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(keep, nr, na, 50)
        eng.run([5], ld=True)
        res = oracle.compare(alle[keep], nr, na, 5, window=50, ld=True)
        assert_bits(eng.site_ll(0), res["site"], "site")
        assert_ld_close(eng.window_ll(0)[:, :2], res["win"][:, :2], "LD")
        assert (eng.windows()[2] == res["nsites"]).all()
        zero = (nr.astype(int) + na) == 0
        assert (eng.site_ll(0)[zero, 0] == 1.0).all() and (eng.site_ll(0)[zero, 2] == 1.0).all()


def test_tiling_options_do_not_change_results(oracle):
    N, L = 700, 450
    alle, nr, na = synth(11, L, N)
    ref = None
    for cpw in (1, 2, 3, 4, 5):
        for waves in (1, 3, 8):
            with E.Engine() as eng:
                eng.set_option("chunks_per_wave", cpw)
                eng.set_option("waves_per_block", waves)
                eng.upload_panel(E.pack_alleles_fast(alle), N)
                eng.upload_sites(np.arange(L), nr, na, 100)
                eng.run([3, 699], ld=True)
                got = np.stack([eng.window_ll(0), eng.window_ll(1)])
            if ref is None:
                ref = got
                res = oracle.compare(alle, nr, na, 3, window=100, ld=True)
                assert_ld_close(got[0][:, :2], res["win"][:, :2], "cpw=1")
            else:
                assert_ld_close(got[..., :2], ref[..., :2], f"cpw={cpw} waves={waves}")
                assert_bits(got[..., 2], ref[..., 2], "LIBD2")


@pytest.mark.parametrize("variant", VARIANTS)
def test_background_subsets_duplicates_and_empty(oracle, variant):
    N, L = 90, 400
    alle, nr, na = synth(21, L, N)
    with E.Engine() as eng:
        eng.set_option("ld_variant", variant)
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        for refids in ([1, 2, 3, 70, 89], [5, 5, 6, 88, 5], [4], list(range(0, 90, 3))):
            eng.run([4], ld=True, bg_count=bg_counts(refids, N), pu_id=6)
            res = oracle.compare(alle, nr, na, 4, window=100, ld=True, refids=refids, pu_id=6)
            assert_ld_close(eng.window_ll(0)[:, :2], res["win"][:, :2], f"bg={refids[:5]}")
        # background = {target}: n_refpanel = 0 -> NaN like the reference's 0/0
        eng.run([4], ld=True, bg_count=bg_counts([4], N))
        assert np.isnan(eng.window_ll(0)[:, :2]).all()
        assert not np.isnan(eng.window_ll(0)[:, 2]).any()


def test_counting_inside_run_and_determinism():
    N, L = 200, 600
    alle, nr, na = synth(31, L, N)
    outs = []
    for opt in (0, 1):
        with E.Engine() as eng:
            eng.set_option("count_in_run", opt)
            eng.upload_panel(E.pack_alleles_fast(alle), N)
            eng.upload_sites(np.arange(L), nr, na, 100)
            for _ in range(2):
                eng.run([0, 1], ld=True)
                outs.append((eng.site_ll(1), eng.window_ll(1), eng.site_af()))
            ms = eng.last_run_ms()
            assert (ms["alt_count"] > 0) == bool(opt) and ms["ld"] > 0 and ms["total"] >= ms["ld"]
            assert (eng.alt_counts(0, L) == alle.sum(axis=1)).all()
    for o in outs[1:]:
        for a, b in zip(o, outs[0]):
            assert_bits(a, b, "run-to-run / option determinism")


@pytest.mark.parametrize("M,cov,T", [(20, 2.0, 9), (40, 9.0, 6), (20, 2.0, 4)])
def test_groups_of_comparison_individuals_share_a_workgroup(oracle, M, cov, T):
    """T >= 4: groups of four comparison individuals go through k_ld_popcount_mt (target-independent
    counts taken once per group), the rest one per workgroup -- every bit as with the option off."""
    N, L = 150, 2600
    rng = np.random.default_rng(500 + M)
    f = rng.beta(0.4, 1.0, size=L).clip(1e-3, 0.999)
    alle = (rng.random((L, 2 * N)) < f[:, None]).astype(np.uint8)
    c = np.minimum(rng.poisson(cov, size=L), M)
    na = rng.binomial(c, f).astype(np.uint8)
    nr = (c - na).astype(np.uint8)
    targets = [int(t) for t in rng.choice(N, size=T, replace=False)]
    bg = rng.integers(0, 3, size=N).astype(np.uint8)
    got = {}
    for mt in (1, 0):
        with E.Engine(0, 0.02, M) as eng:
            eng.set_option("ld_variant", 2)
            eng.set_option("multi_target", mt)
            eng.set_option("mfma_targets", 0)        # T >= 5 would go through k_ld_mfma (next test)
            eng.upload_panel(E.pack_alleles_fast(alle), N)
            eng.upload_sites(np.arange(L), nr, na, 100)
            eng.run(targets, ld=True, bg_count=bg, pu_id=targets[1])
            assert eng.last_ld_variant() == 2
            got[mt] = [(eng.site_ll(i), eng.window_ll(i)) for i in range(T)]
    for i, t in enumerate(targets):
        assert_bits(got[1][i][0], got[0][i][0], f"site target {t}")
        assert_bits(got[1][i][1], got[0][i][1], f"window target {t}")
    for i in (0, T - 1):
        res = oracle.compare(alle, nr, na, targets[i], window=100, ld=True, max_cov=M,
                             refids=np.repeat(np.arange(N), bg), pu_id=targets[1])
        assert_ld_close(got[1][i][1][:, :2], res["win"][:, :2], f"M={M} target {targets[i]}")


@pytest.mark.parametrize("N,L,W,M,cov,T,tmin", [(150, 2600, 100, 20, 2.0, 9, 8), (150, 2600, 100, 40, 9.0, 17, 8),
                                                (70, 900, 7, 20, 2.0, 31, 8), (200, 500, 64, 3, 1.0, 23, 8),
                                                (3, 60, 2, 20, 2.0, 3, 1), (33, 700, 30, 20, 2.0, 33, 8), (90, 800, 100, 20, 2.0, 6, 5),
                                                (90, 800, 100, 20, 2.0, 21, 5),
                                                (300, 1500, 100, 50, 14.0, 15, 8),
                                                (150, 600, 100, 20, 2.0, 130, 8),     # nine groups: two batches of operands
                                                (64, 400, 257, 50, 20.0, 16, 1)])     # deep windows: exponents in the thousands
def test_many_comparison_individuals_through_the_matrix_cores(oracle, N, L, W, M, cov, T, tmin):
    """T >= mfma_min (4 since the single runs count on the matrix cores; 3 and 5 before): groups of 15 comparison individuals go through k_ld_mfma (the G(x,t) sums as integer matrix
    products, 32 background individuals per wave; the window end factored as V_x U_t tau^G), what is left through
    the counting kernels.  Per-row values and LIBD2 are the bits of single runs; the --LD columns agree with the
    counting kernels and, for EVERY comparison individual, with the oracle within the bar (the factored products
    round differently, so "the bits of a single run" is not the claim any more)."""
    rng = np.random.default_rng(900 + N + T)
    f = rng.beta(0.4, 1.0, size=L).clip(1e-3, 0.999)
    alle = (rng.random((L, 2 * N)) < f[:, None]).astype(np.uint8)
    c = np.minimum(rng.poisson(cov, size=L), M)
    na = rng.binomial(c, f).astype(np.uint8)
    nr = (c - na).astype(np.uint8)
    targets = [int(t) for t in rng.choice(N, size=T, replace=False)]
    bg = rng.integers(0, 3, size=N).astype(np.uint8)
    if N == 3:
        bg[:] = 1
    got = {}
    for mfma in (1, 0):
        with E.Engine(0, 0.02, M) as eng:
            eng.set_option("ld_variant", 2)
            eng.set_option("mfma_targets", mfma)
            eng.set_option("mfma_min", tmin)
            eng.set_option("multi_target", mfma)         # the comparison runs: one individual per workgroup
            eng.upload_panel(E.pack_alleles_fast(alle), N)
            eng.upload_sites(np.arange(L), nr, na, W)
            eng.run(targets, ld=True, bg_count=bg, pu_id=targets[1])
            assert eng.last_ld_variant() == 2
            got[mfma] = [(eng.site_ll(i), eng.window_ll(i)) for i in range(T)]
    worst = 0.0
    for i, t in enumerate(targets):
        assert_bits(got[1][i][0], got[0][i][0], f"site target {t}")
        assert_bits(got[1][i][1][:, 2], got[0][i][1][:, 2], f"LIBD2 target {t} (#{i} of {T})")
        assert_ld_close(got[1][i][1][:, :2], got[0][i][1][:, :2], f"matrix cores vs counting kernels, target {t} (#{i} of {T})")
        res = oracle.compare(alle, nr, na, t, window=W, ld=True, max_cov=M,
                             refids=np.repeat(np.arange(N), bg), pu_id=targets[1])
        worst = max(worst, assert_ld_close(got[1][i][1][:, :2], res["win"][:, :2], f"N={N} M={M} target {t} (#{i} of {T})"))
    print(f"N={N} T={T}: max rel err vs the oracle {worst:.2e}")


@pytest.mark.parametrize("N,L,W,T", [(150, 2600, 100, 70), (70, 900, 33, 47), (2504, 700, 100, 40), (520, 1300, 100, 31)])
def test_groups_per_launch_and_workgroup_sums_of_the_matrix_core_kernel(oracle, N, L, W, T):
    """How k_ld_mfma's groups of 15 are dealt to launches (option mfma_batch_groups: one group a launch, two, all of them --
    the groups of one run then sit next to each other on an XCD and share the tile words through its L2) changes no
    arithmetic: the same bits.  Whether a workgroup adds its eight waves' window sums up itself (mfma_wg_sum 1, the default:
    one partial sum per group of eight half chunks) or k_ld_finalize_g adds the half chunks' (0) changes the association of
    the IBD1 sums only: IBD0 and LIBD2 the same bits, IBD1 within 1e-13, and every individual within the bar of the oracle
    either way -- also where the last group of half chunks is short (N = 70: 3 half chunks; 520: 17; 2504: 79)."""
    alle, nr, na = synth(5150 + N + T, L, N)
    rng = np.random.default_rng(N + T)
    targets = [int(t) for t in rng.choice(N, size=T, replace=False)]
    got = {}
    for key, opts in {"one": {"mfma_batch_groups": 1}, "two": {"mfma_batch_groups": 2}, "all": {},
                      "per_half_chunk": {"mfma_wg_sum": 0}}.items():
        with E.Engine() as eng:
            eng.set_option("ld_variant", 2)
            for k, v in opts.items():
                eng.set_option(k, v)
            eng.upload_panel(E.pack_alleles_fast(alle), N)
            eng.upload_sites(np.arange(L), nr, na, W)
            eng.run(targets, ld=True, pu_id=targets[3])
            got[key] = eng.window_ll_all(T)
    assert_bits(got["one"], got["all"], "one group a launch vs all of them")
    assert_bits(got["two"], got["all"], "two groups a launch vs all of them")
    assert_bits(got["per_half_chunk"][:, :, 0], got["all"][:, :, 0], "IBD0")
    assert_bits(got["per_half_chunk"][:, :, 2], got["all"][:, :, 2], "LIBD2")
    a, b = got["per_half_chunk"][:, :, 1], got["all"][:, :, 1]
    ok = np.isfinite(b) & (b != 0)
    assert (np.isnan(a) == np.isnan(b)).all() and (np.abs(a[ok] - b[ok]) <= 1e-13 * np.abs(b[ok])).all()
    for i in (0, 3, T - 1):
        res = oracle.compare(alle, nr, na, targets[i], window=W, ld=True, pu_id=targets[3])
        for key in ("all", "per_half_chunk"):
            assert_ld_close(got[key][i][:, :2], res["win"][:, :2], f"{key}, individual #{i}")


@pytest.mark.parametrize("N,L,W,M,cov", [(70, 900, 100, 20, 2.0), (200, 500, 7, 40, 9.0), (131, 400, 64, 3, 1.0), (3, 60, 2, 20, 2.0)])
def test_reference_order_mode_is_bit_identical_to_the_oracle(oracle, N, L, W, M, cov):
    """ld_variant 3: strict per-individual products, then the background sums taken serially in the
    reference's order (list order of -B with duplicates, exclusions counted out) -- every LIBD0/LIBD1
    bit equals the oracle's, which equals the reference's (17-digit goldens above)."""
    rng = np.random.default_rng(N * 7 + L)
    f = rng.beta(0.4, 1.0, size=L).clip(1e-3, 0.999)
    alle = (rng.random((L, 2 * N)) < f[:, None]).astype(np.uint8)
    c = np.minimum(rng.poisson(cov, size=L), M)
    na = rng.binomial(c, f).astype(np.uint8)
    nr = (c - na).astype(np.uint8)
    targets = [int(t) for t in rng.choice(N, size=min(3, N), replace=False)]
    orders = [None, rng.permutation(np.repeat(np.arange(N), rng.integers(0, 3, size=N)))]
    with E.Engine(0, 0.02, M) as eng:
        eng.set_option("ld_variant", 3)
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, W)
        for order in orders:
            for pu in (-1, targets[0]):
                eng.set_background_order(order)
                eng.run(targets, ld=True, bg_count=bg_counts(order, N), pu_id=pu)
                assert eng.last_ld_variant() == 3
                for i, t in enumerate(targets):
                    res = oracle.compare(alle, nr, na, t, window=W, ld=True, max_cov=M, refids=order, pu_id=pu)
                    assert_bits(eng.site_ll(i), res["site"], "site")
                    assert_bits(eng.window_ll(i), res["win"], f"N={N} target {t} pu={pu} order={'list' if order is not None else 'default'}")
        with pytest.raises(E.EngineError, match="disagree"):
            eng.set_background_order([0])
            eng.run(targets, ld=True)
        eng.set_background_order(None)


def test_async_runs_queue_in_order():
    """"async": ibdg_run only enqueues; results and per-run timings are those of the synchronous calls."""
    N, L = 300, 2500
    alle, nr, na = synth(33, L, N)
    with E.Engine() as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        want = {}
        for t in (5, 6, 9):
            eng.run([t], ld=True)
            want[t] = (eng.site_ll(0), eng.window_ll(0))
        eng.set_option("async", 1)
        for t in (5, 6, 9, 6):                       # four runs queued without a host wait between them
            eng.run([t], ld=True)
        assert_bits(eng.window_ll(0), want[6][1], "async window")      # getters wait for the stream
        assert_bits(eng.site_ll(0), want[6][0], "async site")
        for back in range(4):
            ms = eng.run_ms(back)
            assert ms["ld"] > 0 and ms["total"] >= ms["ld"]
        eng.sync()
        eng.set_option("async", 0)
        eng.run([9], ld=True)
        assert_bits(eng.window_ll(0), want[9][1], "back to synchronous")
        with pytest.raises(E.EngineError):
            eng.run_ms(32)


def test_error_behaviour():
    alle, nr, na = synth(41, 50, 10)
    with E.Engine() as eng:
        with pytest.raises(E.EngineError, match="no panel"):
            eng.upload_sites(np.arange(5), nr[:5], na[:5], 100)
        eng.upload_panel(E.pack_alleles_fast(alle), 10)
        with pytest.raises(E.EngineError, match="no sites"):
            eng.run([0])
        with pytest.raises(E.EngineError, match="outside the panel"):
            eng.upload_sites([50], [1], [1], 100)
        with pytest.raises(E.EngineError, match="max_cov"):
            eng.upload_sites([0], [15], [6], 100)
        with pytest.raises(E.EngineError, match="window"):
            eng.upload_sites([0], [1], [1], 0)
        eng.upload_sites(np.arange(50), nr, na, 100)
        with pytest.raises(E.EngineError, match="not a panel individual"):
            eng.run([10])
        with pytest.raises(E.EngineError, match="unknown option"):
            eng.set_option("nope", 1)
        # empty site list: zero windows, no crash
        eng.upload_sites(np.zeros(0, np.uint32), np.zeros(0, np.uint8), np.zeros(0, np.uint8), 100)
        eng.run([0])
        assert eng.n_windows == 0 and eng.window_ll(0).shape == (0, 3)
    with pytest.raises(E.EngineError, match="-M"):
        E.Engine(0, 0.02, 0)
    with pytest.raises(E.EngineError, match="out of range"):
        E.Engine(99)


def test_all_sites_zero_coverage():
    alle, nr, na = synth(51, 30, 70)
    with E.Engine() as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), 70)
        eng.upload_sites(np.arange(30), np.zeros(30, np.uint8), np.zeros(30, np.uint8), 100)
        eng.run([1], ld=True)
        assert eng.n_windows == 0
        assert (eng.site_ll(0)[:, [0, 2]] == 1.0).all()


@pytest.mark.parametrize("N,L,W,T", [(70, 900, 100, 1), (150, 2500, 37, 3), (300, 700, 2, 6), (64, 300, 257, 2)])
def test_site_results_option_changes_no_window_bit(oracle, N, L, W, T):
    """Option "site_results": 0 keeps (and in --LD mode computes) nothing per row.  The window table must be the same
    bits in both modes, --LD or not, and equal the oracle's; rows without reads between and around the windows must
    not disturb the products (reference src/ibdgem.c:657-667)."""
    alle, nr, na = synth(77 + N, L, N)
    nr[:5] = na[:5] = 0                      # rows without reads before the first window ...
    nr[-7:] = na[-7:] = 0                    # ... and behind the last
    targets = [(3 + 11 * i) % N for i in range(T)]
    with E.Engine(0, 0.02, 20) as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, W)
        for ld in (False, True):
            eng.set_option("site_results", 1)
            eng.run(targets, ld=ld)
            win = [eng.window_ll(i) for i in range(T)]
            site = [eng.site_ll(i) for i in range(T)]
            af = eng.site_af()
            for i, t in enumerate(targets):
                res = oracle.compare(alle, nr, na, t, window=W, ld=ld)
                assert_bits(site[i], res["site"], "site")
                assert_bits(win[i][:, 2], res["win"][:, 2], "LIBD2")
                if ld:
                    assert_ld_close(win[i][:, :2], res["win"][:, :2], "LD window")
                else:
                    assert_bits(win[i], res["win"], "window products")
            eng.set_option("site_results", 0)
            eng.run(targets, ld=ld)
            for i in range(T):
                assert_bits(eng.window_ll(i), win[i], "windows, mode 0")
            with pytest.raises(E.EngineError, match="no per-site results"):
                eng.site_ll(0)
            assert_bits(eng.site_af(), af, "AF column (made on demand, whatever the run kept)")
        eng.set_option("site_results", 1)
        eng.run(targets[:1], ld=True)
        assert_bits(eng.site_af(), af, "AF column")


# --------------------------------------------------------------------------- kernel selection
def test_kernel_selection_and_fallback(oracle):
    """Auto picks the exponent-counting kernel when the P(D|G) table is the plain binomial form -- on the
    panel's own tiles when the rows are in file order, on the compacted tiles otherwise; with a clamped
    table the strict kernel (never an error)."""
    N, L = 150, 500
    alle, nr, na = synth(61, L, N)
    with E.Engine(0, 0.02, 20) as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        eng.run([1], ld=True)
        assert eng.last_ld_variant() == 2 and eng.ld_layout() == 1
        fast = eng.window_ll(0)
        eng.set_option("ld_variant", 1)
        eng.run([1], ld=True)
        assert eng.last_ld_variant() == 1
        assert_ld_close(fast[:, :2], eng.window_ll(0)[:, :2], "fast vs strict")
        assert_bits(fast[:, 2], eng.window_ll(0)[:, 2], "LIBD2")
        # rows out of file order: the panel's own tiles do not apply, the compacted ones do
        eng.set_option("ld_variant", 0)
        perm = np.arange(L)
        perm[[3, 4]] = perm[[4, 3]]
        eng.upload_sites(perm, nr[perm], na[perm], 100)
        eng.run([1], ld=True)
        assert eng.last_ld_variant() == 2 and eng.ld_layout() == 2
        res = oracle.compare(alle[perm], nr[perm], na[perm], 1, window=100, ld=True)
        assert_ld_close(eng.window_ll(0)[:, :2], res["win"][:, :2], "permuted rows")
        # ... unless the caller forbids them: then only the strict kernel is left
        eng.set_option("compact_tiles", -1)
        eng.upload_sites(perm, nr[perm], na[perm], 100)
        eng.run([1], ld=True)
        assert eng.last_ld_variant() == 1 and eng.ld_layout() == 0
        assert_ld_close(eng.window_ll(0)[:, :2], res["win"][:, :2], "permuted rows, strict")
        eng.set_option("ld_variant", 2)
        with pytest.raises(E.EngineError, match="not applicable"):
            eng.run([1], ld=True)
    # epsilon so small that e^a underflows: the reference clamps P(D|G) to DBL_MIN
    # (src/ibd-math.c:77-79); the product form does not hold, strict kernel is used
    with E.Engine(0, 1e-30, 20) as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        eng.run([1], ld=True)
        assert eng.last_ld_variant() == 1
        res = oracle.compare(alle, nr, na, 1, window=100, ld=True, eps=1e-30)
        assert_bits(eng.site_ll(0), res["site"], "site eps=1e-30")
        assert_ld_close(eng.window_ll(0)[:, :2], res["win"][:, :2], "eps=1e-30")


@pytest.mark.parametrize("eps,M,cov", [(0.02, 20, 9.0), (0.2, 7, 3.0), (0.001, 40, 15.0), (0.45, 3, 1.0)])
def test_exponent_counting_other_error_rates_and_depths(oracle, eps, M, cov):
    N, L = 200, 900
    rng = np.random.default_rng(int(cov * 10))
    f = np.clip(rng.beta(0.3, 1.0, size=L), 1e-3, 0.999)
    alle = (rng.random((L, 2 * N)) < f[:, None]).astype(np.uint8)
    c = np.minimum(rng.poisson(cov, size=L), M)
    na = rng.binomial(c, f).astype(np.uint8)
    nr = (c - na).astype(np.uint8)
    with E.Engine(0, eps, M) as eng:
        eng.set_option("ld_variant", 2)
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        for W in (100, 33):
            eng.upload_sites(np.arange(L), nr, na, W)
            eng.run([0, 199], ld=True, pu_id=5)
            for i, t in enumerate((0, 199)):
                res = oracle.compare(alle, nr, na, t, window=W, ld=True, eps=eps, max_cov=M, pu_id=5)
                assert_bits(eng.site_ll(i), res["site"], "site")
                assert_ld_close(eng.window_ll(i)[:, :2], res["win"][:, :2], f"eps={eps} M={M} W={W}")


def test_windows_per_wave_option(oracle):
    N, L = 130, 2000
    alle, nr, na = synth(71, L, N)
    ref = None
    for wpw, guided in ((1, 1), (2, 0), (7, 1), (7, 0), (32, 1), (1000, 0), (1000, 1)):
        with E.Engine() as eng:
            eng.set_option("ld_variant", 2)
            eng.set_option("windows_per_wave", wpw)
            eng.set_option("guided_runs", guided)     # runs shrinking towards the end of the grid, or uniform
            eng.upload_panel(E.pack_alleles_fast(alle), N)
            eng.upload_sites(np.arange(L), nr, na, 50)
            eng.run([2], ld=True)
            got = eng.window_ll(0)
        if ref is None:
            ref = got
            res = oracle.compare(alle, nr, na, 2, window=50, ld=True)
            assert_ld_close(got[:, :2], res["win"][:, :2], "wpw=1")
        else:
            assert_bits(got, ref, f"windows_per_wave={wpw} guided={guided}")


@pytest.mark.parametrize("ring,recbytes,wpw", [(8, 12288, 16), (4, 1024, 16), (8, 1024, 1), (3, 90000, 256), (4, 12288, 16)])
def test_exponent_counting_launch_geometries(oracle, ring, recbytes, wpw):
    """ring depth (default 2), LDS record budget (forces fewer windows per workgroup) and windows per
    wave do not change a single bit of the result."""
    N, L = 200, 3000
    alle, nr, na = synth(81, L, N)
    with E.Engine() as eng:
        eng.set_option("ld_variant", 2)
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        eng.run([3], ld=True)
        ref = eng.window_ll(0)
        res = oracle.compare(alle, nr, na, 3, window=100, ld=True)
        assert_ld_close(ref[:, :2], res["win"][:, :2], "default geometry")
    with E.Engine() as eng:
        eng.set_option("ld_variant", 2)
        eng.set_option("ring_slots", ring)
        eng.set_option("record_lds_bytes", recbytes)
        eng.set_option("windows_per_wave", wpw)
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        eng.run([3], ld=True)
        assert eng.last_ld_variant() == 2
        assert_bits(eng.window_ll(0), ref, f"ring={ring} recbytes={recbytes} wpw={wpw}")


def test_sparse_pileup_rows_far_apart(oracle):
    """Few covered rows spread over many panel rows: windows span hundreds of tiles."""
    N, Lp = 130, 60000
    rng = np.random.default_rng(91)
    alle, nr, na = synth(91, Lp, N)
    keep = np.sort(rng.choice(Lp, size=900, replace=False))
    nrk, nak = np.maximum(nr[keep], 1), na[keep]
    res = oracle.compare(alle[keep], nrk, nak, 7, window=100, ld=True)
    got = {}
    for variant, tiles in ((0, 0), (1, 0), (2, 0), (2, -1), (0, -1)):
        with E.Engine() as eng:
            eng.set_option("ld_variant", variant)
            eng.set_option("compact_tiles", tiles)
            eng.upload_panel(E.pack_alleles_fast(alle), N)
            eng.upload_sites(keep, nrk, nak, 100)
            eng.run([7], ld=True)
            # few covered rows among those spanned: the exponent-counting kernel on the compacted tiles of the site
            # list (src/ibdgem.c:596-601: rows without a pileup line never reach the window loop); with those
            # forbidden, the strict kernel rather than streaming every tile in between
            assert eng.last_ld_variant() == (1 if variant == 1 or (variant, tiles) == (0, -1) else 2)
            assert eng.ld_layout() == (2 if tiles == 0 else 1)
            assert_bits(eng.site_ll(0), res["site"], "site")
            assert_ld_close(eng.window_ll(0)[:, :2], res["win"][:, :2], f"sparse variant={variant} tiles={tiles}")
            got[variant, tiles] = eng.window_ll(0)
    # the counts are exact integers whichever tiles they are taken from: same bits
    assert_bits(got[2, 0], got[2, -1], "compacted vs the panel's own tiles")
    assert_bits(got[0, 0], got[2, 0], "auto = compacted")


def test_rows_tens_of_thousands_of_panel_rows_apart(oracle):
    """A handful of sites whose panel rows lie further apart than the control words of the panel's own tiles can say (more
    than 255 tile pairs between two segments of a run): the compacted tiles take them; with those forbidden, the strict
    kernel -- never an error, never a wrong window."""
    N, Lp = 70, 130000
    rng = np.random.default_rng(17)
    alle = (rng.random((Lp, 2 * N)) < 0.3).astype(np.uint8)
    keep = np.array([10, 11, 40000, 40001, 40002, 90000, 129998, 129999], dtype=np.uint32)
    nr = rng.integers(1, 4, size=len(keep)).astype(np.uint8)
    na = rng.integers(0, 3, size=len(keep)).astype(np.uint8)
    res = oracle.compare(alle[keep], nr, na, 5, window=3, ld=True)
    for tiles, variant, layout in ((0, 2, 2), (-1, 1, 0)):
        with E.Engine() as eng:
            eng.set_option("compact_tiles", tiles)
            eng.upload_panel(E.pack_alleles_fast(alle), N)
            eng.upload_sites(keep, nr, na, 3)
            eng.run([5], ld=True)
            assert eng.last_ld_variant() == variant and eng.ld_layout() == layout
            assert_bits(eng.site_ll(0), res["site"], "site")
            assert_bits(eng.window_ll(0)[:, 2], res["win"][:, 2], "LIBD2")
            assert_ld_close(eng.window_ll(0)[:, :2], res["win"][:, :2], f"tiles={tiles}")


@pytest.mark.parametrize("N,L,W,M,cov,eps,T", [(150, 2600, 100, 20, 2.0, 0.02, 1), (64, 700, 100, 50, 14.0, 0.02, 1),
                                                (131, 1500, 33, 31, 6.0, 0.001, 2), (300, 2000, 64, 20, 0.6, 0.3, 1),
                                                (700, 900, 257, 50, 20.0, 0.02, 1), (2504, 1300, 100, 20, 2.0, 0.1, 3),
                                                (70, 900, 2, 20, 2.0, 0.45, 1), (90, 3000, 100, 7, 3.0, 0.02, 1)])
@pytest.mark.parametrize("tiles", [-1, 1])
def test_matrix_core_counts_give_the_bits_of_the_vector_counts(oracle, N, L, W, M, cov, eps, T, tiles):
    """One comparison individual per workgroup (k_ld_popcount): the four weighted sums of a haplotype word by one
    v_mfma_scale_f32_16x16x128_f8f6f4 (option mx_counts 1, the default: weights 0..7 as FP6, bits as FP4, rows with eight
    reads or more through the rare-plane path; power tables as plain doubles scaled per step) against the twelve (mask,
    count) pairs of mx_counts 0: the sums are the same integers, so the results are the same bits -- deep pileups (planes
    beyond the third), other error rates (another scaling step), windows whose tables do not fit LDS; and both agree with
    the oracle."""
    rng = np.random.default_rng(N * 31 + W)
    f = np.clip(rng.beta(0.3, 1.0, size=L), 1e-3, 0.999)
    alle = (rng.random((L, 2 * N)) < f[:, None]).astype(np.uint8)
    c = np.minimum(rng.poisson(cov, size=L), M)
    na = rng.binomial(c, f).astype(np.uint8)
    nr = (c - na).astype(np.uint8)
    targets = [int(t) for t in rng.choice(N, size=T, replace=False)]
    out = {}
    for mx in (0, 1, 2):
        with E.Engine(0, eps, M) as eng:
            eng.set_option("mx_counts", min(mx, 1))
            # (2: the IBD1 form from the first run on -- the sums ARE the table exponents, IBD0 from one pass over the site list)
            eng.set_option("ibd0_after", 1 if mx == 2 else 0)
            eng.set_option("compact_tiles", tiles)
            eng.set_option("ld_variant", 2)
            eng.set_option("mfma_targets", 0)            # (T = 2, 3: single runs of k_ld_popcount, blockIdx.z = individual)
            eng.set_option("multi_target", 0)
            eng.upload_panel(E.pack_alleles_fast(alle), N)
            eng.upload_sites(np.arange(L), nr, na, W)
            eng.run(targets, ld=True, pu_id=targets[0] if T > 1 else -1)
            assert eng.last_ld_variant() == 2 and eng.last_count_unit() in ((1 + mx,) if mx < 2 else (2, 3))
            if mx == 2 and (N, W) in ((150, 100), (2504, 100), (90, 100)):
                assert eng.last_count_unit() == 3            # (the others: tables beyond LDS)
            out[mx] = [(eng.site_ll(i), eng.window_ll(i)) for i in range(T)]
    for i, t in enumerate(targets):
        assert_bits(out[1][i][0], out[0][i][0], f"t={t} per-row values")
        assert_bits(out[1][i][1], out[0][i][1], f"t={t} windows")
        assert_bits(out[2][i][1], out[0][i][1], f"t={t} windows, IBD1 form")
    res = oracle.compare(alle, nr, na, targets[0], window=W, ld=True, eps=eps, max_cov=M, pu_id=targets[0] if T > 1 else -1)
    assert_bits(out[1][0][1][:, 2], res["win"][:, 2], "LIBD2")
    assert_ld_close(out[1][0][1][:, :2], res["win"][:, :2], "matrix-core counts vs oracle")


@pytest.mark.parametrize("N,L,W,M,cov,T", [(150, 2600, 100, 20, 2.0, 1), (70, 900, 2, 20, 2.0, 2), (131, 1700, 33, 3, 1.0, 4),
                                            (300, 3000, 64, 20, 0.4, 9), (700, 5000, 257, 40, 9.0, 17), (64, 640, 32, 20, 2.0, 5),
                                            (2504, 1300, 100, 20, 2.0, 3), (130, 4000, 31, 20, 5.0, 31)])
def test_compacted_tiles_give_the_bits_of_the_panel_tiles(oracle, N, L, W, M, cov, T):
    """The compacted tiles of a site list (rows with reads only; back to back -- option compact_align 1, the default: windows
    straddle tiles -- or every window on a tile boundary, compact_align 32, or on a 4-row boundary) against the panel's own
    tiles: the exponents are exact integer sums over the same rows, so every --LD kernel (single, groups of four, matrix
    cores) returns the same bits from any of them; and all agree with the oracle."""
    alle, nr, na = synth(7000 + N + W, L, N, cov_mean=cov)
    na = np.minimum(na, M).astype(np.uint8)
    nr = np.minimum(nr, M - na.astype(int)).astype(np.uint8)
    rng = np.random.default_rng(N * W)
    targets = [int(t) for t in rng.choice(N, size=min(T, N), replace=False)]
    out = {}
    for tiles, align in ((-1, 1), (1, 1), (1, 32), (1, 4)):
        with E.Engine(0, 0.02, M) as eng:
            eng.set_option("compact_tiles", tiles)
            eng.set_option("compact_align", align)
            eng.set_option("ld_variant", 2)          # (a thin pileup on the panel's own tiles would take the strict kernel)
            eng.upload_panel(E.pack_alleles_fast(alle), N)
            eng.upload_sites(np.arange(L), nr, na, W)
            eng.run(targets, ld=True, pu_id=targets[-1] if T > 2 else -1)
            assert eng.last_ld_variant() == 2 and eng.ld_layout() == (2 if tiles == 1 else 1)
            out[tiles if align == 1 else align] = [(eng.site_ll(i), eng.window_ll(i)) for i in range(len(targets))]
    for i, t in enumerate(targets):
        for form in (1, 32, 4):
            assert_bits(out[form][i][0], out[-1][i][0], f"t={t} per-row values, form {form}")
            assert_bits(out[form][i][1], out[-1][i][1], f"t={t} windows, form {form}")
    for i in sorted({0, len(targets) - 1}):
        res = oracle.compare(alle, nr, na, targets[i], window=W, ld=True, pu_id=targets[-1] if T > 2 else -1, max_cov=M)
        assert_bits(out[1][i][0], res["site"], "site")
        assert_bits(out[1][i][1][:, 2], res["win"][:, 2], "LIBD2")
        assert_ld_close(out[1][i][1][:, :2], res["win"][:, :2], f"compacted tiles t={targets[i]}")


@pytest.mark.parametrize("eps,M,cov", [(0.02, 20, 2.0), (0.001, 40, 9.0), (0.3, 20, 6.0)])
def test_plain_double_powers_in_the_matrix_core_kernel(oracle, eps, M, cov):
    """k_ld_mfma looks tau^G up as a plain double (8 bytes) in windows none of whose powers leaves the double range, and
    as {mantissa, exponent} (16 bytes) otherwise: the same value, mV tau^G = (mV mtau) 2^etau; the plain form adds a lane's
    two haplotypes before it scales them, so the two forms agree to the last place or two, not bit for bit -- with a tiny
    error rate (tau = 4e-3: the plain form covers windows of up to ~125 reads only) both run in one launch."""
    N, L, T = 150, 2600, 17
    rng = np.random.default_rng(int(eps * 1000) + M)
    f = rng.beta(0.4, 1.0, size=L).clip(1e-3, 0.999)
    alle = (rng.random((L, 2 * N)) < f[:, None]).astype(np.uint8)
    c = np.minimum(rng.poisson(cov, size=L), M)
    na = rng.binomial(c, f).astype(np.uint8)
    nr = (c - na).astype(np.uint8)
    targets = [int(t) for t in rng.choice(N, size=T, replace=False)]
    got = {}
    for plain in (1, 0):
        with E.Engine(0, eps, M) as eng:
            eng.set_option("mfma_plain_tau", plain)
            eng.upload_panel(E.pack_alleles_fast(alle), N)
            eng.upload_sites(np.arange(L), nr, na, 100)
            eng.run(targets, ld=True)
            assert eng.last_ld_variant() == 2
            got[plain] = [eng.window_ll(i) for i in range(T)]
    for i in range(T):
        assert_bits(got[1][i][:, 2], got[0][i][:, 2], f"target {targets[i]} LIBD2")
        rel = assert_ld_close(got[1][i][:, :2], got[0][i][:, :2], f"target {targets[i]}")
        assert rel <= 1e-13, rel          # the two forms round differently in the last place, no more
        # ... and both are the reference's values (src/ibdgem.c:669-753 restated), not merely each other's
        ref = oracle.compare(alle, nr, na, targets[i], window=100, eps=eps, max_cov=M, ld=True)
        assert_bits(got[1][i][:, 2], ref["win"][:, 2], f"target {targets[i]} LIBD2 vs oracle")
        for plain in (1, 0):
            assert_ld_close(got[plain][i][:, :2], ref["win"][:, :2], f"target {targets[i]} vs oracle, mfma_plain_tau {plain}")


def test_many_comparison_individuals_switch_to_the_compacted_tiles(oracle):
    """The runs on one upload add up: once they hold "compact_targets" comparison individuals (a group of the matrix-core
    kernel counts as 20; an individual of the counting kernels as 16 when it counts by (mask, count) pairs, option
    mx_counts 0, and as 12 with its sums on the matrix cores) the site
    list is re-laid out once (the site list belongs to the pileup, src/ibdgem.c:522); later runs on the same upload keep
    the compacted tiles; results unchanged."""
    N, L = 200, 3000
    alle, nr, na = synth(4242, L, N)
    targets = list(range(3, 3 + 20))
    with E.Engine() as eng:
        eng.set_option("compact_targets", 80)
        eng.set_option("mfma_min", 5)
        eng.set_option("mx_counts", 0)
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        eng.run(targets[:4], ld=True)                    # 4 x 16 = 64
        assert eng.ld_layout() == 1
        four = [eng.window_ll(i) for i in range(4)]
        eng.run(targets, ld=True)                        # + two groups, 20 each = 104
        assert eng.ld_layout() == 2 and eng.last_ld_variant() == 2
        many = [eng.window_ll(i) for i in range(len(targets))]
        for i in (0, 7, 19):
            res = oracle.compare(alle, nr, na, targets[i], window=100, ld=True)
            assert_ld_close(many[i][:, :2], res["win"][:, :2], f"t={targets[i]}")
            assert_bits(many[i][:, 2], res["win"][:, 2], "LIBD2")
        eng.run(targets[:4], ld=True)
        assert eng.ld_layout() == 2
        for i in range(4):
            assert_bits(eng.window_ll(i), four[i], f"four again, t={targets[i]}")
        # a new upload starts from the panel's own tiles again
        eng.upload_sites(np.arange(L), nr, na, 100)
        assert eng.ld_layout() == 1
        # one individual at a time (the reference's own loop, src/ibdgem.c:522): the fifth run re-lays out, same bits
        per_run = []
        for k in range(7):
            eng.run([targets[k % 2]], ld=True)
            assert eng.ld_layout() == (1 if k < 4 else 2), k
            per_run.append((eng.site_ll(0), eng.window_ll(0)))
        for k in range(2, 7):
            assert_bits(per_run[k][0], per_run[k % 2][0], f"run {k} per-row values")
            assert_bits(per_run[k][1], per_run[k % 2][1], f"run {k} windows")
        # ... with the sums of a word on the matrix cores a single run counts as 12 (round 5: the rows back to back save it a
        # tenth of its time): the seventh run (84 >= 80) re-lays out, same bits; groups count as before
        eng.set_option("mx_counts", 1)
        eng.upload_sites(np.arange(L), nr, na, 100)
        for k in range(9):
            eng.run([targets[k % 2]], ld=True)
            # (the eighth single run on one upload also makes the pass for the IBD0 terms, option ibd0_after: count unit 3)
            assert eng.ld_layout() == (1 if k < 6 else 2) and eng.last_count_unit() == (2 if k < 7 else 3), k
            assert_bits(eng.window_ll(0), per_run[k % 2][1], f"matrix-core counts, run {k}")
        eng.set_option("compact_targets", 100)
        eng.upload_sites(np.arange(L), nr, na, 100)
        for k in range(3):
            eng.run(targets, ld=True)                    # two groups, 20 each: 40, 80, 120
            assert eng.ld_layout() == (1 if k < 2 else 2), k
        # ... and one run of many individuals on a fresh upload does so at once
        eng.set_option("compact_targets", 16)
        eng.upload_sites(np.arange(L), nr, na, 100)
        eng.run(targets, ld=True)
        assert eng.ld_layout() == 2
        # forbidden: never
        eng.set_option("compact_tiles", -1)
        eng.upload_sites(np.arange(L), nr, na, 100)
        for k in range(3):
            eng.run(targets, ld=True)
        assert eng.ld_layout() == 1


def test_queued_runs_leave_the_finalising_step_to_the_next_launch(oracle):
    """Option "async": a run of single comparison individuals leaves its finalising step (k_ld_finalize's arithmetic) to
    the next run's --LD launch when that run is over the same individuals, background and prepared sites -- otherwise,
    and whenever somebody reads results or replaces inputs first, a launch of its own makes up for it.  Whatever the
    sequence, the window tables are the bits of the synchronous runs (src/ibdgem.c:751-752 either way)."""
    N, L = 300, 5000
    alle, nr, na = synth(1234, L, N)
    bg = np.random.default_rng(5).integers(0, 3, size=N).astype(np.uint8)
    with E.Engine() as eng:
        eng.set_option("multi_target", 0)
        eng.set_option("mfma_targets", 0)
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        want = {}
        for key, (tg, kw) in {"a": ([3], {}), "b": ([5], {}), "c": ([3, 5, 9], {}), "d": ([3], {"bg_count": bg, "pu_id": 3})}.items():
            eng.run(tg, ld=True, **kw)
            want[key] = [eng.window_ll(i) for i in range(len(tg))]
        res_a = oracle.compare(alle, nr, na, 3, window=100, ld=True)
        assert_ld_close(want["a"][0][:, :2], res_a["win"][:, :2], "synchronous run vs oracle")
        for fin_next in (1, 0):
            eng.set_option("finalize_in_next", fin_next)
            eng.set_option("async", 1)
            for _ in range(4):
                eng.run([3], ld=True)                                    # three finalising steps ride along, the last one is made up for
            assert_bits(eng.window_ll(0), want["a"][0], f"queued runs, finalize_in_next {fin_next}")
            eng.run([3], ld=True)
            eng.run([5], ld=True)                                        # other individual: the pending step first
            eng.run([5], ld=True)
            assert_bits(eng.window_ll(0), want["b"][0], "after a change of individual")
            eng.run([3, 5, 9], ld=True)
            eng.run([3, 5, 9], ld=True)
            for i in range(3):
                assert_bits(eng.window_ll(i), want["c"][i], f"three single individuals per run, {i}")
            every = eng.window_ll_all(3)                                 # ibdg_get_window_ll_all: the same tables in one copy
            for i in range(3):
                assert_bits(every[i], want["c"][i], f"all tables at once, {i}")
            eng.run([3], ld=True, bg_count=bg, pu_id=3)
            eng.run([3], ld=True, bg_count=bg, pu_id=3)
            eng.run([3], ld=True)                                        # other background counts
            assert_bits(eng.window_ll(0), want["a"][0], "after a change of background")
            eng.run([3], ld=True, bg_count=bg, pu_id=3)
            eng.run([3], ld=True, bg_count=bg, pu_id=3)
            eng.sync()
            assert_bits(eng.window_ll(0), want["d"][0], "background multiplicities")
            # new sites between queued runs: the pending step belongs to the old ones
            eng.run([3], ld=True)
            eng.upload_sites(np.arange(L // 2), nr[:L // 2], na[:L // 2], 100)
            eng.run([3], ld=True)
            eng.run([3], ld=True)
            half = eng.window_ll(0)
            eng.set_option("async", 0)
            eng.run([3], ld=True)
            assert_bits(half, eng.window_ll(0), "after new sites")
            eng.upload_sites(np.arange(L), nr, na, 100)


def test_a_new_individual_per_queued_run(oracle):
    """The reference hands every individual of the panel to the same rows in turn (src/ibdgem.c:522): queued runs (option
    "async") over a NEW comparison individual each -- indices through the page-locked ring, weights and background size made on
    the device (k_target_weights), the finalising step of run i inside the --LD launch of run i + 1 with run i's own
    background size -- end, whatever their number, with the bits of that individual's synchronous run; the oracle agrees."""
    N, L = 300, 5000
    alle, nr, na = synth(4321, L, N)
    bg = np.random.default_rng(6).integers(0, 3, size=N).astype(np.uint8)
    bg[[3, 5]] = (2, 1)                      # the individuals' own multiplicities differ: so do their background sizes
    people = [3, 5, 8, 299, 64, 3, 127]
    with E.Engine() as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        for kw in ({}, {"bg_count": bg, "pu_id": 8}):
            want = {}
            for t in set(people):
                eng.run([t], ld=True, **kw)
                want[t] = eng.window_ll(0)
            refids = None if not kw else [n for n in range(N) for _ in range(int(bg[n]))]
            for t in (3, 299):
                ref = oracle.compare(alle, nr, na, t, window=100, ld=True, refids=refids, pu_id=kw.get("pu_id", -1))
                assert_ld_close(want[t][:, :2], ref["win"][:, :2], f"synchronous run of {t} vs oracle")
                assert_bits(want[t][:, 2], ref["win"][:, 2], f"LIBD2 of {t} vs oracle")
            eng.set_option("async", 1)
            for fin_next in (1, 0):
                eng.set_option("finalize_in_next", fin_next)
                for n in range(1, len(people) + 1):
                    for t in people[:n]:
                        eng.run([t], ld=True, **kw)
                    assert eng.last_ld_variant() == 2
                    assert_bits(eng.window_ll(0), want[people[n - 1]], f"{n} queued runs, finalize_in_next {fin_next}, {kw.keys()}")
                # more runs in flight than the ring of page-locked slots holds, then three individuals per run in turn
                for k in range(23):
                    eng.run([people[k % len(people)]], ld=True, **kw)
                assert_bits(eng.window_ll(0), want[people[22 % len(people)]], "23 queued runs")
                for tg in ([3, 5, 8], [5, 8, 299], [3, 5, 8]):
                    eng.run(tg, ld=True, **kw)
                every = eng.window_ll_all(3)
                for i, t in enumerate([3, 5, 8]):
                    assert_bits(every[i], want[t], f"three per run, {i}")
            eng.set_option("async", 0)
            eng.set_option("finalize_in_next", 1)
        with pytest.raises(E.EngineError, match="the last run had 3"):
            eng.window_ll_all(2)


def test_the_second_stream_never_overtakes_a_run_s_individuals():
    """Where a run's individuals are prepared on the MAIN stream (option prep_ahead 0, or more than 64 of them) in a queue of
    runs, the second stream -- per-row values and LIBD2 of the windows, which read the individuals' indices too -- starts
    behind the previous run's end: it must wait for this run's indices as well (found by the fuzzer as a memory fault: a
    fresh ring slot's garbage read as an individual).  Queued runs over other individuals each, then the per-row values and
    windows of the last one against its synchronous run, bit for bit."""
    N, L = 200, 20000
    alle, nr, na = synth(97531, L, N)
    with E.Engine() as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        want = {}
        for t in (3, 150):
            eng.run([t], ld=True)
            want[t] = (eng.site_ll(0), eng.window_ll(0))
        many = list(range(10, 80))                          # 70 > 64: prepared on the main stream whatever the option
        eng.run(many, ld=True)
        want_many = (eng.site_ll(69), eng.window_ll(69))
        eng.set_option("async", 1)
        for prep_ahead in (0, 1):
            eng.set_option("prep_ahead", prep_ahead)
            for rounds in range(6):
                for k in range(9):
                    eng.run([(7 * k + rounds) % N], ld=True)
                last = (3, 150)[rounds & 1]
                eng.run([last], ld=True)
                assert_bits(eng.site_ll(0), want[last][0], f"per-row values, prep_ahead {prep_ahead}, round {rounds}")
                assert_bits(eng.window_ll(0), want[last][1], f"windows, prep_ahead {prep_ahead}, round {rounds}")
            for rounds in range(3):
                eng.run([(11 * rounds) % N], ld=True)
                eng.run(list(range(100 + rounds, 170 + rounds)), ld=True)
                eng.run(many, ld=True)
                assert_bits(eng.site_ll(69), want_many[0], f"70 individuals, per-row values, round {rounds}")
                assert_bits(eng.window_ll(69), want_many[1], f"70 individuals, windows, round {rounds}")
        eng.set_option("async", 0)


@pytest.mark.parametrize("tiles", [-1, 1])
def test_ibd0_from_one_pass_over_the_site_list(oracle, tiles):
    """What a background individual's own genotype contributes to IBD0 (src/ibdgem.c:715, :743) does not depend on the
    comparison individual, whose only trace in that sum is its own exclusion (:714): once the single runs on an upload and a
    background have added up (option "ibd0_after"), ONE pass keeps those products, and later runs count the IBD1 sums only
    (ibdg_last_count_unit 3: the matrix instruction returns the table exponents themselves).  The additions are the same in
    the same order, so every run returns the BITS of the form that counts everything -- synchronous, queued with the
    finalising step inside the next launch or behind its own, with a background list and -N, beside groups of 15 and of 4,
    after a change of background and after new sites; the oracle agrees."""
    N, L = 300, 5200
    alle, nr, na = synth(4711, L, N, cov_mean=2.6)       # (some rows with eight reads or more: the rare-plane path)
    assert (nr.astype(int) + na).max() >= 8
    bg = np.random.default_rng(8).integers(0, 3, size=N).astype(np.uint8)
    bg[[3, 64]] = (2, 1)
    people = [3, 64, 8, 299, 127, 3, 63, 255, 256, 0]
    kws = ({}, {"bg_count": bg, "pu_id": 8})
    want = [{}, {}]
    with E.Engine() as eng:
        eng.set_option("compact_tiles", tiles)
        eng.set_option("ibd0_after", 0)
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        for j, kw in enumerate(kws):
            for t in sorted(set(people)):
                eng.run([t], ld=True, **kw)
                assert eng.last_count_unit() == 2
                want[j][t] = eng.window_ll(0)
            refids = None if not kw else [n for n in range(N) for _ in range(int(bg[n]))]
            ref = oracle.compare(alle, nr, na, 64, window=100, ld=True, refids=refids, pu_id=kw.get("pu_id", -1))
            assert_ld_close(want[j][64][:, :2], ref["win"][:, :2], "the form that counts everything vs oracle")
    with E.Engine() as eng:
        eng.set_option("compact_tiles", tiles)
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        assert eng.set_option("ibd0_after", 4) is None
        for j, kw in enumerate(kws):
            # the first three runs count everything, the fourth makes the pass; a new background starts over
            for k, t in enumerate(people):
                eng.run([t], ld=True, **kw)
                assert eng.last_count_unit() == (2 if k < 3 else 3), (j, k)
                assert_bits(eng.window_ll(0), want[j][t], f"run {k} of individual {t}, background {j}")
            eng.set_option("async", 1)
            for fin_next in (1, 0):
                eng.set_option("finalize_in_next", fin_next)
                for n in (1, 2, 5, len(people)):
                    for t in people[:n]:
                        eng.run([t], ld=True, **kw)
                    assert eng.last_count_unit() == 3
                    assert_bits(eng.window_ll(0), want[j][people[n - 1]], f"{n} queued runs, finalize_in_next {fin_next}")
            eng.set_option("finalize_in_next", 1)
            eng.set_option("async", 0)
            # three per run (one workgroup each), nineteen (a group of 15 on the matrix cores and four single ones), and with the
            # groups of four of the vector kernel
            eng.run([3, 64, 8], ld=True, **kw)
            for i, t in enumerate([3, 64, 8]):
                assert_bits(eng.window_ll(i), want[j][t], f"three per run, {t}")
            many = list(range(100, 109)) + [63, 255, 256, 0, 3, 64, 8, 299, 127]     # 15 + 3
            eng.run(many, ld=True, **kw)
            for i, t in enumerate(many):
                if t in want[j]:
                    got = eng.window_ll(i)
                    assert_bits(got[:, 0], want[j][t][:, 0], f"IBD0 of {t} beside a group of 15")
                    assert_bits(got[:, 2], want[j][t][:, 2], f"IBD2 of {t}")
                    (assert_bits if i >= 15 else assert_ld_close)(got[:, 1], want[j][t][:, 1], f"IBD1 of {t} beside a group of 15")
            eng.set_option("mfma_targets", 0)
            eng.run(many[-7:], ld=True, **kw)                 # a group of four of k_ld_popcount_mt + three single ones
            for i, t in enumerate(many[-7:]):
                assert_bits(eng.window_ll(i), want[j][t], f"beside a group of four, {t}")
            eng.set_option("mfma_targets", 1)
        # new sites: the count starts over, the old pass is not used
        half = L // 2
        eng.upload_sites(np.arange(half), nr[:half], na[:half], 100)
        eng.run([64], ld=True)
        assert eng.last_count_unit() == 2
        first = eng.window_ll(0)
        for _ in range(4):
            eng.run([64], ld=True)
        assert eng.last_count_unit() == 3
        assert_bits(eng.window_ll(0), first, "after new sites")
        ref = oracle.compare(alle[:half], nr[:half], na[:half], 64, window=100, ld=True)
        assert_ld_close(first[:, :2], ref["win"][:, :2], "new sites vs oracle")


@pytest.mark.parametrize("variant", [1, 3])
def test_a_pending_finalising_step_never_lands_behind_a_strict_run(variant):
    """A queued counting run leaves its finalising step pending; the SAME comparison run again under ld_variant 1 or 3 (strict
    products / reference order) writes its window table itself -- the pending step must be made up for BEFORE that run, not
    behind it (where it would overwrite the strict kernel's bits with the counting kernel's sums)."""
    N, L = 300, 5000
    alle, nr, na = synth(99, L, N)
    with E.Engine() as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        eng.set_option("ld_variant", variant)
        eng.run([3], ld=True)
        strict = eng.window_ll(0)
        eng.set_option("ld_variant", 0)
        eng.run([3], ld=True)
        counting = eng.window_ll(0)
        assert not np.array_equal(bits(strict[:, :2]), bits(counting[:, :2]))      # (they agree to 1e-14, not bit for bit)
        eng.set_option("async", 1)
        for _ in range(3):
            eng.run([3], ld=True)                      # ... the last one's finalising step is pending
        eng.set_option("ld_variant", variant)
        eng.run([3], ld=True)
        assert eng.last_ld_variant() == variant
        assert_bits(eng.window_ll(0), strict, f"ld_variant {variant} behind queued counting runs")
        eng.set_option("ld_variant", 0)
        eng.run([3], ld=True)
        eng.run([3], ld=False)                         # a non-LD run takes no pending step along either
        eng.sync()
        eng.run([3], ld=True)
        assert_bits(eng.window_ll(0), counting, "counting runs again")
        eng.set_option("async", 0)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_sequences_of_queued_runs_uploads_and_reads(seed):
    """Queued runs (option "async") in random order with reads, new sites, other individuals, other backgrounds, several
    individuals per run (single runs, groups of four, the matrix-core groups) and the finalising step left to the next
    launch or not: every table read is the bits of the same run made synchronously on a fresh engine state."""
    rng = np.random.default_rng(seed)
    N, L = 200, 4000
    alle, nr, na = synth(77 + seed, L, N)
    bg = rng.integers(0, 3, size=N).astype(np.uint8)
    site_sets = [np.arange(L), np.arange(L // 3, L), np.sort(rng.choice(L, size=L // 2, replace=False))]
    target_sets = [[3], [5], [3, 5, 9], list(range(20, 27)), list(range(40, 58))]
    kw_sets = [{}, {"bg_count": bg, "pu_id": 3}]

    def key(si, ti, ki):
        return si, ti, ki

    want = {}
    with E.Engine() as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        for si, rows in enumerate(site_sets):
            eng.upload_sites(rows, nr[rows], na[rows], 100)
            for ti, tg in enumerate(target_sets):
                for ki, kw in enumerate(kw_sets):
                    eng.run(tg, ld=True, **kw)
                    want[key(si, ti, ki)] = [eng.window_ll(i) for i in range(len(tg))]
        eng.set_option("async", 1)
        si = 0
        eng.upload_sites(site_sets[0], nr[site_sets[0]], na[site_sets[0]], 100)
        last = None
        for step in range(150):
            op = rng.random()
            if op < 0.6:
                ti, ki = (int(rng.integers(len(target_sets))), int(rng.integers(len(kw_sets)))) if last is None or rng.random() < 0.4 else last[1:]
                eng.run(target_sets[ti], ld=True, **kw_sets[ki])
                last = (si, ti, ki)
            elif op < 0.8 and last is not None:
                tg = target_sets[last[1]]
                i = int(rng.integers(len(tg)))
                assert_bits(eng.window_ll(i), want[last][i], f"seed {seed} step {step}: {last}, individual {i}")
            elif op < 0.88:
                si = int(rng.integers(len(site_sets)))
                eng.upload_sites(site_sets[si], nr[site_sets[si]], na[site_sets[si]], 100)
                last = None
            elif op < 0.94:
                eng.set_option("finalize_in_next", int(rng.integers(2)))
            else:
                eng.sync()
        if last is not None:
            assert_bits(eng.window_ll(0), want[last][0], f"seed {seed}: the last run {last}")


def test_dispatch_events_give_the_dominant_kernel_duration():
    """Option "dispatch_events": the --LD launches are timed through their own dispatch packets; the
    dominant kernel's duration lies inside the interval of the --LD launches, results unchanged."""
    N, L = 300, 4000
    alle, nr, na = synth(77, L, N)
    with E.Engine() as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        eng.run([5], ld=True)
        want = eng.window_ll(0)
        with pytest.raises(E.EngineError, match="did not use"):
            eng.set_option("ld_variant", 1)
            eng.run([5], ld=True)
            eng.run_kernel_ms(0)
        eng.set_option("ld_variant", 0)
        eng.set_option("dispatch_events", 1)
        for async_mode in (0, 1):
            eng.set_option("async", async_mode)
            for _ in range(3):
                eng.run([5], ld=True)
            eng.sync()
            for back in range(3):
                k, ms = eng.run_kernel_ms(back), eng.run_ms(back)
                assert 0 < k <= ms["ld"] <= ms["total"] + 1e-3, (k, ms)
            assert_bits(eng.window_ll(0), want, "dispatch events")
        eng.set_option("async", 0)


def test_results_are_the_bits_of_round_1():
    """Site preparation moved from the host to the device in round 2 (ibdg_prep.hip, the x87 products in
    integer arithmetic) and the --LD kernel was reworked: the window and per-site results must still be
    the very bits the round-1 library produced (tests/golden/r01_ld_hashes.json, tools/r01_hashes.py)."""
    import json
    REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import r01_hashes
    with open(os.path.join(REPO, "tests", "golden", "r01_ld_hashes.json")) as fh:
        want = json.load(fh)["sha256"]
    assert r01_hashes.digests() == want


# --------------------------------------------------------------------------- site preparation on the device
def test_panel_from_a_file(tmp_path):
    """ibdg_upload_panel_fd: the packed rows in an open file from a byte offset on (the host program's panel cache) -- the
    staging threads read the file themselves.  Same alt counts and the same bits of a run as the panel from host memory; a
    file that ends before the rows do is an error, not a short panel."""
    N, L = 2504, 30_000                     # 30 000 rows x 640 bytes = 19 MB: three 8 MB pieces for the staging threads
    alle, nr, na = synth(31, L, N)
    packed = E.pack_alleles_fast(alle)
    fn = tmp_path / "rows.bin"
    off = 4096 + 24
    with open(fn, "wb") as fh:
        fh.write(b"\xa5" * off)
        fh.write(packed.tobytes())
    with E.Engine() as eng:
        eng.upload_panel(packed, N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        eng.run([5, 9], ld=True)
        want = [eng.window_ll(i) for i in range(2)]
        counts = eng.alt_counts(0, L)
        fd = os.open(fn, os.O_RDONLY)
        try:
            eng.upload_panel_fd(fd, off, L, N)
            assert (eng.alt_counts(0, L) == counts).all()
            eng.upload_sites(np.arange(L), nr, na, 100)
            eng.run([5, 9], ld=True)
            for i in range(2):
                assert_bits(eng.window_ll(i), want[i], f"panel from the file, individual {i}")
            with pytest.raises(E.EngineError, match="file ends before the rows do"):
                eng.upload_panel_fd(fd, off + 8, L, N)
            with pytest.raises(E.EngineError, match="not an open file"):
                eng.upload_panel_fd(-1, 0, L, N)
        finally:
            os.close(fd)


def test_one_engine_two_panels(oracle):
    """A context that uploads a second panel with another number of individuals (same lane count) and
    runs the same comparison individuals must not reuse the background weights of the first."""
    alle_a, nr, na = synth(71, 400, 100)
    alle_b, _, _ = synth(72, 400, 120)
    with E.Engine() as eng:
        for alle, N in ((alle_a, 100), (alle_b, 120), (alle_a, 100)):
            eng.upload_panel(E.pack_alleles_fast(alle), N)
            with pytest.raises(E.EngineError, match="no sites"):
                eng.run([3, 50], ld=True)                 # the sites of the previous panel are gone
            eng.upload_sites(np.arange(400), nr, na, 100)
            bg = np.ones(N, dtype=np.uint8)
            bg[7] = 3
            for bgc, refids in ((None, None), (bg, list(range(N)) + [7, 7])):
                eng.run([3, 50], ld=True, bg_count=bgc)
                for i, t in enumerate((3, 50)):
                    res = oracle.compare(alle, nr, na, t, window=100, ld=True, refids=refids)
                    assert_bits(eng.site_ll(i), res["site"], f"N={N} site")
                    assert_ld_close(eng.window_ll(i)[:, :2], res["win"][:, :2], f"N={N} t={t} LD")


def windows_numpy(nr, na, W):
    cov = np.flatnonzero((nr.astype(int) + na) > 0)
    n_win = (len(cov) + W - 1) // W
    first = cov[::W][:n_win]
    last = np.array([cov[min(len(cov), (w + 1) * W) - 1] for w in range(n_win)], dtype=np.int64)
    ncov = np.array([min(len(cov), (w + 1) * W) - w * W for w in range(n_win)], dtype=np.int64)
    return first, last, ncov


def covered_ranks_check(lib_path=None):
    """Ranks of the covered rows (cov_site / rec_cov of the stage-A scatter kernel) at MORE than 256 workgroups of the
    preparation kernels -- several per CU -- through the finest view the ABI gives of them: windows of 2 covered rows
    (first / last site of every window = every covered row's rank), row indirection, zero-coverage rows of three
    densities; and the per-row values the records produce.  Returns the number of workgroups of the scatter kernel."""
    N = 3
    L = 700_003                                   # 684 workgroups of 1024 rows
    rng = np.random.default_rng(20261004)
    alle = (rng.random((1500, 2 * N)) < 0.3).astype(np.uint8)
    rows = np.sort(rng.integers(0, 1500, size=L)).astype(np.uint32)
    for p_zero in (0.135, 0.7, 0.02):
        cov = np.where(rng.random(L) < p_zero, 0, 1 + rng.integers(0, 5, size=L))
        na = rng.binomial(cov, 0.3).astype(np.uint8)
        nr = (cov - na).astype(np.uint8)
        want_first, want_last, want_ncov = windows_numpy(nr, na, 2)
        with E.Engine(lib_path=lib_path) as eng:
            eng.upload_panel(E.pack_alleles_fast(alle), N)
            eng.upload_sites(rows, nr, na, 2)
            first, last, ncov = eng.windows()
            assert len(first) == len(want_first)
            bad = np.flatnonzero((first != want_first) | (last != want_last) | (ncov != want_ncov))
            assert len(bad) == 0, (f"p_zero={p_zero}: {len(bad)} windows with wrong bounds, first at window {bad[0]} "
                                   f"(site {want_first[bad[0]]}, scatter workgroup {want_first[bad[0]] // 1024})")
            eng.run([1], ld=False)
            site = eng.site_ll(0)
            zero = (nr.astype(int) + na) == 0
            assert (site[zero] == 1.0).all() and (site[~zero, 0] < 1.0).all()
    return (L + 1023) // 1024


def test_covered_row_ranks_at_several_workgroups_per_cu():
    assert covered_ranks_check() > 256


@pytest.mark.parametrize("L,W,cov_mean", [(20000, 100, 2.0), (9000, 3, 0.3), (4096, 4096, 1.0), (4097, 17, 5.0),
                                          (70000, 1000, 2.0)])
def test_device_resident_inputs_identity_rows_and_pinned_arrays(oracle, L, W, cov_mean):
    """ibdg_upload_sites with pageable arrays, with page-locked arrays (ibdg_host_alloc), with row_index
    NULL, and ibdg_upload_sites_dev with torch tensors all give the same windows and results."""
    import torch
    N = 64
    alle, nr, na = synth(500 + L, L, N, cov_mean)
    want_first, want_last, want_ncov = windows_numpy(nr, na, W)
    got = []
    with E.Engine() as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        pins = [E.PinnedArray(L, np.uint32), E.PinnedArray(L, np.uint8), E.PinnedArray(L, np.uint8)]
        pins[0].array[:] = np.arange(L)
        pins[1].array[:] = nr
        pins[2].array[:] = na
        dr = torch.arange(L, dtype=torch.int32, device="cuda")
        dnr, dna = torch.from_numpy(nr).cuda(), torch.from_numpy(na).cuda()
        torch.cuda.synchronize()
        uploads = [lambda: eng.upload_sites(np.arange(L), nr, na, W),
                   lambda: eng.upload_sites(None, nr, na, W),
                   lambda: eng.upload_sites(pins[0].array, pins[1].array, pins[2].array, W),
                   lambda: eng.upload_sites_dev(dr.data_ptr(), dnr.data_ptr(), dna.data_ptr(), L, W),
                   lambda: eng.upload_sites_dev(None, dnr.data_ptr(), dna.data_ptr(), L, W)]
        for up in uploads:
            up()
            ms = eng.upload_ms()
            assert ms["call"] > 0 and ms["device_prep"] > 0
            first, last, ncov = eng.windows()
            assert (first == want_first).all() and (last == want_last).all() and (ncov == want_ncov).all()
            eng.run([9], ld=True)
            assert eng.last_ld_variant() == 2
            out = E.PinnedArray((L, 3), np.float64)
            got.append((eng.window_ll(0).copy(), eng.site_ll(0, out=out.array).copy()))
            out.close()
        for p in pins:
            p.close()
    for w, s in got[1:]:
        assert_bits(w, got[0][0], "windows of the upload variants")
        assert_bits(s, got[0][1], "sites of the upload variants")
    res = oracle.compare(alle, nr, na, 9, window=W, ld=True)
    assert_bits(got[0][1], res["site"], "site")
    assert_bits(got[0][0][:, 2], res["win"][:, 2], "LIBD2")
    assert_ld_close(got[0][0][:, :2], res["win"][:, :2], "LD")


def test_two_contexts_taking_turns_give_the_bits_of_one(oracle):
    """Comparisons streamed through two contexts on one GPU (bench.py engine_clock.two_contexts_alternating_ms): with
    "async" and "dev_inputs_ready" the preparation of one context's comparison runs under the --LD kernel of the
    other's; every window table is the one a lone context computes for that comparison."""
    import torch
    N, L, W = 96, 6000, 20
    alle, nr0, na0 = synth(4242, L, N, 1.2)
    rng = np.random.default_rng(77)
    comps = []
    for k in range(6):                       # six comparisons: different read counts and comparison individuals
        nr = rng.poisson(0.7, L).clip(0, 10).astype(np.uint8)
        na = rng.poisson(0.5, L).clip(0, 10).astype(np.uint8)
        comps.append((nr, na, int(rng.integers(0, N))))
    dev = [(torch.from_numpy(nr).cuda(), torch.from_numpy(na).cuda()) for nr, na, _ in comps]
    torch.cuda.synchronize()
    words = E.pack_alleles_fast(alle)
    lone = []
    with E.Engine() as eng:
        eng.upload_panel(words, N)
        for (nr, na, t), (dnr, dna) in zip(comps, dev):
            eng.upload_sites_dev(None, dnr.data_ptr(), dna.data_ptr(), L, W)
            eng.run([t], ld=True)
            lone.append(eng.window_ll(0).copy())
    with E.Engine() as a, E.Engine() as b:
        engs = [a, b]
        for e in engs:
            e.upload_panel(words, N)
            e.set_option("async", 1)
            e.set_option("dev_inputs_ready", 1)
        got = [None] * len(comps)

        def submit(i):
            e = engs[i & 1]
            e.upload_sites_dev(None, dev[i][0].data_ptr(), dev[i][1].data_ptr(), L, W)
            e.run([comps[i][2]], ld=True)
        submit(0)
        for i in range(1, len(comps)):
            submit(i)
            got[i - 1] = engs[(i - 1) & 1].window_ll(0).copy()
        got[-1] = engs[(len(comps) - 1) & 1].window_ll(0).copy()
    for i, (g, w) in enumerate(zip(got, lone)):
        assert_bits(g, w, f"comparison {i} through two contexts")
    nr, na, t = comps[3]
    res = oracle.compare(alle, nr, na, t, window=W, ld=True)
    assert_bits(got[3][:, 2], res["win"][:, 2], "LIBD2")
    assert_ld_close(got[3][:, :2], res["win"][:, :2], "LD")


def test_rows_in_any_order_and_the_first_offending_site(oracle):
    """Rows visited in a random order (several scan blocks, duplicates included) go through the compacted tiles and
    still equal the oracle; errors name the first offending site in list order, its row before its counts."""
    N, Lp, L = 70, 3000, 10000
    alle, _, _ = synth(81, Lp, N)
    rng = np.random.default_rng(82)
    rows = rng.integers(0, Lp, size=L).astype(np.uint32)
    cov = np.minimum(rng.poisson(1.5, size=L), 20)
    na = rng.binomial(cov, 0.3).astype(np.uint8)
    nr = (cov - na).astype(np.uint8)
    with E.Engine() as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(rows, nr, na, 100)
        eng.run([1], ld=True)
        assert eng.last_ld_variant() == 2 and eng.ld_layout() == 2
        res = oracle.compare(alle[rows], nr, na, 1, window=100, ld=True)
        assert_bits(eng.site_ll(0), res["site"], "site")
        assert_ld_close(eng.window_ll(0)[:, :2], res["win"][:, :2], "LD")
        first, last, ncov = eng.windows()
        assert (first == res["first"]).all() and (last == res["last"]).all() and (ncov == res["nsites"]).all()
        bad_rows, bad_nr = rows.copy(), nr.copy()
        bad_nr[5000] = 21
        bad_rows[7000] = Lp
        with pytest.raises(E.EngineError, match=r"site 5000 has n_ref\+n_alt=2[1-9]"):
            eng.upload_sites(bad_rows, bad_nr, na, 100)
        bad_rows[5000] = Lp + 3
        with pytest.raises(E.EngineError, match=rf"row_index\[5000\]={Lp + 3} outside the panel"):
            eng.upload_sites(bad_rows, bad_nr, na, 100)
        bad_rows[4999] = 2 ** 32 - 1
        with pytest.raises(E.EngineError, match=r"row_index\[4999\]=4294967295"):
            eng.upload_sites(bad_rows, bad_nr, na, 100)
        with pytest.raises(E.EngineError, match="no sites"):
            eng.run([1], ld=True)                     # a failed upload leaves no sites behind
        with pytest.raises(E.EngineError, match="outside the panel"):
            eng.upload_sites(None, np.ones(Lp + 1, np.uint8), np.ones(Lp + 1, np.uint8), 100)


@pytest.mark.parametrize("N,L", [(3, 131), (64, 70), (65, 300), (700, 257), (2504, 1003), (4000, 67), (5000, 130)])
def test_alt_counts_for_every_row_layout(N, L):
    """k_alt_count reads the panel as a flat stream of 16-byte units (rows per group = 64/gcd(units per
    row, 64); 4000 individuals: 65 units per row, the wave-per-row kernel) -- counts equal numpy's for
    every row, also when the rows do not fill the last group, both at upload and inside a run."""
    rng = np.random.default_rng(N)
    alle = (rng.random((L, 2 * N)) < rng.random((L, 1))).astype(np.uint8)
    want = alle.sum(axis=1, dtype=np.uint32)
    with E.Engine() as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        assert (eng.alt_counts(0, L) == want).all()
        eng.set_option("count_in_run", 1)
        eng.upload_sites(None, np.ones(L, np.uint8), np.zeros(L, np.uint8), 10)
        eng.run([0], ld=False)
        assert (eng.alt_counts(0, L) == want).all()
        assert_bits(eng.site_af(), want / float(2 * N), "AF")


def test_host_twins_return_the_device_bits():
    """ibdg_pdg_ibd0 / ibdg_pdg_ibd1 on the host (with ibdg_pdg_table and the alt counts) give, row by row,
    the doubles k_site wrote on the device."""
    N, L = 77, 400
    alle, nr, na = synth(91, L, N)
    lib = E.load_library()
    tab = np.empty((21, 21, 3), dtype=np.float64)
    assert lib.ibdg_pdg_table(0.02, 20, tab.ctypes.data) == 0
    with E.Engine() as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(None, nr, na, 100)
        eng.run([13], ld=False)
        site, af = eng.site_ll(0), eng.site_af()
        counts = eng.alt_counts(0, L)
    for s in range(L):
        f = float(counts[s]) / float(2 * N)
        assert f.hex() == float(af[s]).hex()
        p = [float(x) for x in tab[nr[s], na[s]]]
        a0, a1 = int(alle[s, 26]), int(alle[s, 27])
        assert lib.ibdg_pdg_ibd0(f, *p).hex() == float(site[s, 0]).hex(), s
        assert lib.ibdg_pdg_ibd1(a0, a1, f, *p).hex() == float(site[s, 1]).hex(), s
        assert p[a0 + a1] == site[s, 2]
