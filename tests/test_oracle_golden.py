"""Pin the oracle (oracle/ibd_oracle.c) against the reference.

  * the reference's own 18 fixture files (supplementary/ibdgem-test/output):
    text identity of every likelihood at the 7 digits the reference prints;
  * 17-digit outputs of the reference itself on synthetic inputs
    (tests/golden/syn*): BIT-exact, per site and per window, LD and non-LD;
  * the reference's ibd-math.c functions over the whole (n_ref,n_alt) grid
    (tests/golden/math_grid.tsv.gz): BIT-exact.

CPU only.  The oracle is the checker for the GPU parity tests, so it has to be
right first.
"""
import gzip
import math
import os

import numpy as np
import pytest

import golden_io as G


def bits(a):
    return np.asarray(a, dtype=np.float64).view(np.uint64)


def assert_bit_equal(got, want, what):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, what
    nan_g, nan_w = np.isnan(got), np.isnan(want)
    assert (nan_g == nan_w).all(), f"{what}: NaN pattern differs"
    ok = (bits(got) == bits(want)) | nan_g
    if not ok.all():
        i = np.argwhere(~ok)[0]
        raise AssertionError(f"{what}: first mismatch at {tuple(i)}: {got[tuple(i)]!r} vs {want[tuple(i)]!r}"
                             f" ({(~ok).sum()} of {ok.size})")


FIX = [(f"sample{k}", f"sample{t}") for k in (1, 2, 3) for t in (1, 2, 3)]


@pytest.mark.parametrize("sq,target", FIX)
def test_reference_fixture_files(oracle, sq, target):
    panel = G.load_fixture_panel()
    tab, summ = G.fixture_outputs(sq, target)
    rows = np.array([panel.row_of_pos[int(p)] for p in tab.pos])
    res = oracle.compare(panel.alleles[rows], tab.n_ref, tab.n_alt, panel.index(target), ld=False)
    assert tab.processed == len(tab.pos)
    # GT columns are the target's alleles
    t = panel.index(target)
    assert (panel.alleles[rows, 2 * t] == tab.a0).all() and (panel.alleles[rows, 2 * t + 1] == tab.a1).all()
    for i in range(len(rows)):
        assert "%f" % res["af"][i] == tab.af_txt[i]
        assert ["%e" % v for v in res["site"][i]] == tab.ll_txt[i], (i, res["site"][i], tab.ll_txt[i])
    assert len(res["win"]) == len(summ.ll)
    for w in range(len(summ.ll)):
        assert ["%e" % v for v in res["win"][w]] == summ.ll_txt[w]
        assert res["nsites"][w] == summ.nsites[w]
        assert tab.pos[res["first"][w]] == summ.start[w] and tab.pos[res["last"][w]] == summ.end[w]


def _syn_cases():
    out = []
    for tag in ("synA", "synB"):
        for case in G.cases(tag)["cases"]:
            out.append((tag, case))
    return out


@pytest.mark.parametrize("tag,case", _syn_cases())
def test_reference_17digit_outputs(oracle, tag, case):
    flags, panel, names, refids, pu_id, per_target = G.case_setup(tag, case)
    assert names
    for name in names:
        tab, summ, alle, fo, _ = per_target[name]
        res = oracle.compare(alle, tab.n_ref, tab.n_alt, panel.index(name), window=flags["window"],
                             eps=flags["eps"], max_cov=flags["max_cov"], refids=refids, pu_id=pu_id,
                             ld=flags["ld"], f_override=fo)
        assert_bit_equal(res["site"], tab.ll, f"{tag}/{case}/{name} per-site LL")
        for i in range(len(tab.pos)):
            assert "%f" % res["af"][i] == tab.af_txt[i]
        assert len(res["win"]) == len(summ.ll), f"{tag}/{case}/{name} window count"
        assert_bit_equal(res["win"], summ.ll, f"{tag}/{case}/{name} window LL")
        assert (res["nsites"] == summ.nsites).all()
        if len(summ.ll):
            assert (tab.pos[res["first"]] == summ.start).all() and (tab.pos[res["last"]] == summ.end).all()


def test_golden_cases_cover_the_edges():
    """The fixtures really contain the edge cases the suite claims to cover."""
    _, s = G.syn_outputs("synA", "ld_bg_self_nan", "UNKWN", "ind3")
    assert np.isnan(s.ll[:, :2]).all() and not np.isnan(s.ll[:, 2]).any()
    t, s = G.syn_outputs("synA", "ld_default", "UNKWN", "ind3")
    assert ((t.n_ref + t.n_alt) == 0).any(), "zero-coverage rows present"
    assert s.nsites[-1] < 100, "partial last window present"
    _, s2 = G.fixture_outputs("sample2", "sample3")
    assert (s2.ll[:, 2] == 0).all(), "fixture with underflow to zero"
    tv, _ = G.syn_outputs("synA", "ld_varsites", "UNKWN", "ind64")
    assert ((tv.a0 + tv.a1) > 0).all()
    td, _ = G.syn_outputs("synA", "ld_downsample", "UNKWN", "ind4")
    assert (td.n_ref.astype(int) + td.n_alt <= td.dp).all() and ((td.n_ref.astype(int) + td.n_alt) < td.dp).any()


def test_reference_math_grid(oracle):
    path = os.path.join(G.GOLD, "math_grid.tsv.gz")
    n = 0
    nck_cache = {}
    with gzip.open(path, "rt") as fh:
        for line in fh:
            if line.startswith("#"):
                continue
            c = line.rstrip("\n").split("\t")
            eps, M, r, a = float.fromhex(c[0]), int(c[1]), int(c[2]), int(c[3])
            f = float.fromhex(c[4])
            want = [float.fromhex(x) for x in c[5:]]
            key = (eps, M, r, a)
            if key not in nck_cache:
                nck_cache[key] = oracle.pdg(eps, M, r, a)
            p = nck_cache[key]
            got = p + [oracle.lib.orc_pDgf(f, *p)] + [oracle.lib.orc_pDgIBD1(x, y, f, *p)
                                                     for x, y in ((0, 0), (1, 0), (1, 1))]
            assert [float(x).hex() for x in got] == [float(x).hex() for x in want], (key, f)
            n += 1
    assert n > 5000


def test_nck_table(oracle):
    t = oracle.nck(20)
    for i in range(21):
        for j in range(21):
            assert t[i, j] == (math.comb(i, j) if j <= i else 0)
