"""Parity at BASELINE.json sizes (GPU).

configs[1]  non-LD, ~100k SNP rows, 1 comparison individual   -> bit-exact vs the oracle
configs[2]  --LD, ~100k rows, 100-individual panel, window 100 -> per-site bit-exact, LD 1e-10
configs[3]  --LD, 4M rows, 2504-individual panel, window 100   -> (a) EVERY window is recomputed by
            the oracle (a pool of host processes over ranges of windows, tests/oracle_pool.py;
            ~70 core-seconds) from the panel rows read back from the device input, (b) the strict and the
            exponent-counting kernels agree on every window, (c) two half-chromosome shards cut at a
            window boundary reproduce the whole, (d) a second run is bit-identical, (e) alt counts of
            sampled rows equal numpy popcounts.
The synthetic data are bench.py's (same generator, same seed)."""
import contextlib
import os
import shutil
import sys
import tempfile

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from ibdgem_amd import engine as E            # noqa: E402
from ibdgem_amd.sharding import shard_rows    # noqa: E402
import oracle_pool                            # noqa: E402

pytestmark = pytest.mark.gpu
TINY = 1e-290


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def ld_close(got, want, rtol=1e-10):
    got, want = np.asarray(got), np.asarray(want)
    tiny = np.abs(want) < TINY
    assert (np.abs(got[tiny]) < TINY).all()
    rel = np.abs(got[~tiny] - want[~tiny]) / np.abs(want[~tiny])
    assert rel.size == 0 or rel.max() <= rtol, rel.max()
    return 0.0 if rel.size == 0 else float(rel.max())


def synth_host(seed, L, N):
    rng = np.random.default_rng(seed)
    f = np.clip(rng.beta(0.3, 1.0, size=L), 1e-3, 0.999)
    alle = np.empty((L, 2 * N), dtype=np.uint8)
    step = max(1000, 40_000_000 // (2 * N))            # ~320 MB of random doubles per turn
    for a in range(0, L, step):
        b = min(L, a + step)
        alle[a:b] = rng.random((b - a, 2 * N)) < f[a:b, None]
    cov = np.minimum(rng.poisson(2.0, size=L), 20)
    n_alt = rng.binomial(cov, f)
    return alle, (cov - n_alt).astype(np.uint8), n_alt.astype(np.uint8)


@pytest.mark.parametrize("N", [64, 2504])
def test_config1_nonld_100k_rows(oracle, N):
    """BASELINE.json configs[1]: non-LD, ~100k rows, one comparison individual; SURVEY s8 sizes it at the 2504-individual
    panel (the panel's width only enters through the AF column and the comparison individual's two bits)."""
    L = 100_000
    alle, nr, na = synth_host(1, L, N)
    with E.Engine() as eng:
        packed = np.concatenate([E.pack_alleles_fast(alle[a:a + 10_000]) for a in range(0, L, 10_000)])
        eng.upload_panel(packed, N)
        del packed
        eng.upload_sites(np.arange(L), nr, na, 100)
        eng.run([11], ld=False)
        res = oracle.compare(alle, nr, na, 11, window=100, ld=False)
        assert (bits(eng.site_ll(0)) == bits(res["site"])).all()
        assert (bits(eng.site_af()) == bits(res["af"])).all()
        assert (bits(eng.window_ll(0)) == bits(res["win"])).all()
        first, last, ncov = eng.windows()
        assert (first == res["first"]).all() and (last == res["last"]).all() and (ncov == res["nsites"]).all()


def test_config2_ld_100k_rows_100_individuals(oracle):
    L, N = 100_000, 100
    alle, nr, na = synth_host(2, L, N)
    with E.Engine() as eng:
        eng.upload_panel(E.pack_alleles_fast(alle), N)
        eng.upload_sites(np.arange(L), nr, na, 100)
        res = oracle.compare(alle, nr, na, 42, window=100, ld=True)
        for variant in (1, 2):
            eng.set_option("ld_variant", variant)
            eng.run([42], ld=True)
            assert eng.last_ld_variant() == variant
            assert (bits(eng.site_ll(0)) == bits(res["site"])).all()
            win = eng.window_ll(0)
            assert (bits(win[:, 2]) == bits(res["win"][:, 2])).all()
            ld_close(win[:, :2], res["win"][:, :2])


@pytest.fixture(scope="module")
def chr1():
    """bench.py's 4M x 2504 workload, panel resident in an engine."""
    import torch
    import bench
    dev = torch.device("cuda", 0)
    L, N, target, seed = 4_000_000, 2504, 7, 20241008
    panel, nr, na = bench.build_shard(torch, dev, 0, L, N, target, seed)
    torch.cuda.synchronize()
    eng = E.Engine(0, 0.02, 20)
    eng.upload_panel_dev(panel.data_ptr(), L, N)
    state = dict(eng=eng, panel=panel, nr=nr, na=na, L=L, N=N, target=target, torch=torch)
    yield state
    eng.close()
    if state.get("shm"):
        shutil.rmtree(state["shm"], ignore_errors=True)


@contextlib.contextmanager
def shm_dir(chr1):
    """A directory in /dev/shm holding the packed panel and the read counts for the oracle pool (the
    panel is written once per session), cleaned of the per-call result files on exit."""
    d = chr1.get("shm")
    if d is None:
        d = chr1["shm"] = tempfile.mkdtemp(prefix="ibdg_pool_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        words = np.empty((chr1["L"], chr1["panel"].shape[1]), dtype=np.uint64)
        for a in range(0, chr1["L"], 500_000):              # 2.56 GB, copied in slices
            words[a:a + 500_000] = chr1["panel"][a:a + 500_000].cpu().numpy().view(np.uint64)
        np.save(os.path.join(d, "panel.npy"), words)
        del words
        np.save(os.path.join(d, "n_ref.npy"), chr1["nr"])
        np.save(os.path.join(d, "n_alt.npy"), chr1["na"])
    try:
        yield d
    finally:
        for fn in ("win.npy", "site.npy"):
            if os.path.exists(os.path.join(d, fn)):
                os.remove(os.path.join(d, fn))


def test_config3_chr1_2504_individuals(chr1, oracle):
    import bench
    eng, nr, na, L, N, t = chr1["eng"], chr1["nr"], chr1["na"], chr1["L"], chr1["N"], chr1["target"]
    eng.upload_sites(np.arange(L, dtype=np.uint32), nr, na, 100)
    eng.run([t], ld=True)
    assert eng.last_ld_variant() == 2
    win = eng.window_ll(0)
    site = eng.site_ll(0)
    first, last, ncov = eng.windows()
    n_win = len(win)
    assert n_win == (int(((nr.astype(int) + na) > 0).sum()) + 99) // 100 and (ncov[:-1] == 100).all()

    # (d) run-to-run determinism
    eng.run([t], ld=True)
    assert (bits(eng.window_ll(0)) == bits(win)).all() and (bits(eng.site_ll(0)) == bits(site)).all()

    # (b) strict kernel (reference operation order per individual) on every window
    eng.set_option("ld_variant", 1)
    eng.run([t], ld=True)
    strict = eng.window_ll(0)
    eng.set_option("ld_variant", 0)
    assert (bits(strict[:, 2]) == bits(win[:, 2])).all()
    worst = ld_close(win[:, :2], strict[:, :2])
    print(f"exponent counting vs strict over {n_win} windows: max rel {worst:.2e}")

    # (a) the oracle on EVERY window, from the rows as the device received them
    with shm_dir(chr1) as d:
        np.save(os.path.join(d, "win.npy"), win)
        np.save(os.path.join(d, "site.npy"), site)
        checked, worst_orc, bad = oracle_pool.check_all_windows(d, first, L, N, t, 100)
    assert not bad, bad[:5]
    assert checked == n_win
    print(f"oracle on all {checked} windows: max rel {worst_orc:.2e}")
    rng = np.random.default_rng(5)

    # (e) alt counts
    rows = rng.integers(0, L, 200)
    for r in rows:
        words = chr1["panel"][int(r)].cpu().numpy().view(np.uint64)
        assert eng.alt_counts(int(r), 1)[0] == sum(bin(int(x)).count("1") for x in words)

    # (f) the layout of bench.py's timed steps: the engine re-lays the site list out by itself once the runs on it have added up
    # (the 22nd single run) -- the rows with reads back to back -- and queued runs over a NEW individual each end with the bits of
    # that individual's synchronous run on the panel's own tiles (checked against the oracle on every window above)
    other = (t + 1) % N
    eng.run([other], ld=True)
    win_other = eng.window_ll(0)
    assert eng.ld_layout() == 1
    eng.set_option("async", 1)
    for k in range(30):
        eng.run([t if k & 1 else other], ld=True)
    # (... and, from the eighth single run on, with the IBD0 terms from one pass over the site list: ibdg_last_count_unit 3)
    assert eng.ld_layout() == 2 and eng.last_ld_variant() == 2 and eng.last_count_unit() == 3
    assert (bits(eng.window_ll(0)) == bits(win)).all() and (bits(eng.site_ll(0)) == bits(site)).all()
    eng.run([other], ld=True)
    assert (bits(eng.window_ll(0)) == bits(win_other)).all()
    eng.set_option("async", 0)

    # (c) two shards cut at a window boundary reproduce the whole (the multi-GPU decomposition)
    cuts = shard_rows(nr, na, 100, 2)
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        eng.upload_sites(np.arange(a, b, dtype=np.uint32), nr[a:b], na[a:b], 100)
        eng.run([t], ld=True)
        parts.append(eng.window_ll(0))
    assert (bits(np.concatenate(parts)) == bits(win)).all()


def test_config5_500_comparison_individuals_in_one_launch(chr1, oracle):
    """BASELINE configs[4] shape: chr1, 2504-individual panel, 500 comparison individuals batched
    in ONE ibdg_run (sites x targets x panel).  Batched results must equal single-target runs bit
    for bit, and sampled windows must agree with the oracle."""
    import bench
    eng, nr, na, L, N = chr1["eng"], chr1["nr"], chr1["na"], chr1["L"], chr1["N"]
    rng = np.random.default_rng(17)
    targets = np.sort(rng.choice(N, size=500, replace=False)).astype(np.uint32)
    eng.upload_sites(np.arange(L, dtype=np.uint32), nr, na, 100)
    eng.run(targets, ld=True, pu_id=int(targets[3]))
    assert eng.last_ld_variant() == 2
    ms = eng.last_run_ms()
    first, last, ncov = eng.windows()
    n_win = len(first)
    print(f"500 targets x {L} rows x {N} individuals: {ms['total']:.1f} ms device time, "
          f"{500 * int(ncov.sum()) / (ms['total'] * 1e-3):.3e} site-target pairs/s")
    picks = [0, 3, 250, 499]
    batched = {i: (eng.window_ll(i), eng.site_ll(i)) for i in picks}
    for i in picks:
        eng.run([int(targets[i])], ld=True, pu_id=int(targets[3]))
        one = eng.window_ll(0)
        assert (bits(one[:, 2]) == bits(batched[i][0][:, 2])).all(), f"target slot {i}: LIBD2"
        ld_close(batched[i][0][:, :2], one[:, :2])          # factored window end: within the bar, not the same bits
        assert (bits(eng.site_ll(0)) == bits(batched[i][1])).all(), f"target slot {i}: sites"
    # the oracle on every window of two of the 500 (slot 3 is also the pileup's own sample, -N)
    for i in (3, 499):
        with shm_dir(chr1) as d:
            np.save(os.path.join(d, "win.npy"), batched[i][0])
            np.save(os.path.join(d, "site.npy"), batched[i][1])
            checked, worst, bad = oracle_pool.check_all_windows(d, first, L, N, int(targets[i]), 100,
                                                                pu_id=int(targets[3]))
        assert not bad, bad[:5]
        assert checked == n_win
        print(f"target slot {i}: oracle on all {checked} windows, max rel {worst:.2e}")


def test_large_panel_from_host_memory_goes_through_the_staging_team():
    """ibdg_upload_panel of 256 MB and more from ordinary host memory is staged by a team of host threads
    through page-locked buffers; from a mapped file (what the host program's packed-panel cache is) and from
    an anonymous array, with the option on and off, the device must hold the same rows: alt counts of every
    row equal numpy's, and a comparison gives the same bits."""
    import mmap
    import tempfile
    N, L = 2504, 450_000                                       # 288 MB of packed rows
    rng = np.random.default_rng(77)
    words = rng.integers(0, 2 ** 63, size=(L, 80), dtype=np.int64).view(np.uint64)
    words[:, 78] &= np.uint64((1 << 8) - 1)                    # individuals 2496..2503: 8 bits of the last chunk
    words[:, 79] &= np.uint64((1 << 8) - 1)
    want = np.unpackbits(words.view(np.uint8), axis=1).sum(axis=1, dtype=np.uint32)
    nr = rng.integers(0, 3, size=L).astype(np.uint8)
    na = rng.integers(0, 3, size=L).astype(np.uint8)
    results = []
    with tempfile.NamedTemporaryFile(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as fh:
        fh.write(words.tobytes())
        fh.flush()
        mm = mmap.mmap(fh.fileno(), 0, prot=mmap.PROT_READ)
        mapped = np.frombuffer(mm, dtype=np.uint64).reshape(L, 80)
        with E.Engine() as eng:
            for src, staged in ((mapped, 1), (words, 1), (words, 0), (mapped, 0)):
                eng.set_option("staged_upload", staged)
                eng.upload_panel(src, N)
                assert (eng.alt_counts(0, L) == want).all()
                eng.upload_sites(None, nr, na, 100)
                eng.run([2500], ld=True)
                results.append(eng.window_ll(0))
        del mapped, src                                        # the mapping goes with its last array
    for r in results[1:]:
        assert (bits(r) == bits(results[0])).all()
