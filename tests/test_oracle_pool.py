"""The all-windows checker of the full-size GPU tests (tests/oracle_pool.py), exercised on the CPU: fed
the oracle's own results it must pass, fed a perturbed copy it must complain."""
import os
import tempfile

import numpy as np

import oracle_pool
from ibdgem_amd.engine import pack_alleles_fast


def test_pool_accepts_the_oracle_and_flags_a_wrong_window(oracle):
    rng = np.random.default_rng(8)
    L, N, W, t = 3000, 70, 37, 5
    f = np.clip(rng.beta(0.3, 1.0, size=L), 1e-3, 0.999)
    alle = (rng.random((L, 2 * N)) < f[:, None]).astype(np.uint8)
    cov = np.minimum(rng.poisson(1.2, size=L), 20)
    cov[:7] = 0                                        # rows without reads before the first window
    n_alt = rng.binomial(cov, f).astype(np.uint8)
    n_ref = (cov - n_alt).astype(np.uint8)
    res = oracle.compare(alle, n_ref, n_alt, t, window=W, ld=True, pu_id=9)
    packed = pack_alleles_fast(alle)
    assert (oracle_pool.unpack_rows(packed, N) == alle).all()
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "panel.npy"), packed)
        np.save(os.path.join(d, "n_ref.npy"), n_ref)
        np.save(os.path.join(d, "n_alt.npy"), n_alt)
        np.save(os.path.join(d, "site.npy"), res["site"])
        np.save(os.path.join(d, "win.npy"), res["win"])
        checked, worst, bad = oracle_pool.check_all_windows(d, res["first"], L, N, t, W, pu_id=9,
                                                            windows_per_job=16, workers=2)
        assert checked == len(res["win"]) and worst == 0.0 and not bad
        win = res["win"].copy()
        win[len(win) // 2, 0] *= 1 + 1e-8
        site = res["site"].copy()
        site[L - 1, 1] = np.nextafter(site[L - 1, 1], 1.0)
        np.save(os.path.join(d, "win.npy"), win)
        np.save(os.path.join(d, "site.npy"), site)
        checked, worst, bad = oracle_pool.check_all_windows(d, res["first"], L, N, t, W, pu_id=9,
                                                            windows_per_job=16, workers=2)
        assert worst > 1e-10 and len(bad) == 2
