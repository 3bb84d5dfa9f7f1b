"""Window results of seeded --LD workloads as SHA-256 digests (run on a GPU box).

    python tools/r01_hashes.py --lib build/r01/libibdgem_hip.so --out tests/golden/r01_ld_hashes.json

was run once with the library built from the round-1 tree (commit d64d2bf: `git archive d64d2bf
ibdgem_amd/csrc include | tar -x -C /tmp/r01 && make -C /tmp/r01/ibdgem_amd/csrc`) to pin the bits of
that round's exponent-counting kernel; tests/test_gpu_parity.py::test_results_are_the_bits_of_round_1
recomputes the digests with the current library.  Without --out the digests are printed.
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CASES = [  # name, rows, individuals, window, epsilon, max_cov, depth, targets, seed
    ("n300_w100", 6000, 300, 100, 0.02, 20, 2.0, [7], 1),
    ("n2504_w100", 12000, 2504, 100, 0.02, 20, 2.0, [11], 2),
    ("n100_w37_deep", 4000, 100, 37, 0.05, 30, 6.0, [3, 5, 8, 13, 21], 3),
    ("n700_w64_sparse", 9000, 700, 64, 0.01, 20, 0.4, [0, 699], 4),
    ("n65_w2", 500, 65, 2, 0.2, 8, 1.0, [64], 5),
]


def make_case(rows, n_ids, depth, max_cov, seed):
    rng = np.random.default_rng(seed)
    f = np.clip(rng.beta(0.3, 1.0, size=rows), 1e-3, 0.999)
    alle = (rng.random((rows, 2 * n_ids)) < f[:, None]).astype(np.uint8)
    cov = np.minimum(rng.poisson(depth, size=rows), max_cov)
    n_alt = rng.binomial(cov, f).astype(np.uint8)
    n_ref = (cov - n_alt).astype(np.uint8)
    return alle, n_ref, n_alt


def digests(lib_path=None):
    import ibdgem_amd
    from ibdgem_amd.engine import pack_alleles_fast
    out = {}
    for name, rows, n_ids, window, eps, max_cov, depth, targets, seed in CASES:
        alle, n_ref, n_alt = make_case(rows, n_ids, depth, max_cov, seed)
        with ibdgem_amd.Engine(0, eps, max_cov, lib_path=lib_path) as eng:
            eng.set_option("ld_variant", 2)
            try:
                # the exponent-counting kernels are what round 1 had and what these digests pin; five or more comparison
                # individuals would otherwise take the matrix-core kernel, whose factored products round differently
                eng.set_option("mfma_targets", 0)
            except ibdgem_amd.EngineError:
                pass                              # (the round-1 library has no such option)
            eng.upload_panel(pack_alleles_fast(alle), n_ids)
            eng.upload_sites(np.arange(rows, dtype=np.uint32), n_ref, n_alt, window)
            eng.run(targets, ld=True)
            h = hashlib.sha256()
            for t in range(len(targets)):
                h.update(eng.window_ll(t).tobytes())
                h.update(eng.site_ll(t).tobytes())
            first, last, ncov = eng.windows()
            h.update(first.tobytes() + last.tobytes() + ncov.tobytes())
            out[name] = h.hexdigest()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    d = digests(a.lib)
    if a.out:
        with open(a.out, "w") as fh:
            json.dump({"library": "round 1 (commit d64d2bf), ld_variant 2", "sha256": d}, fh, indent=1)
            fh.write("\n")
    print(json.dumps(d, indent=1))
