"""Long random queues of runs (option "async") over changing sets of comparison individuals on ONE site list, with the
queue-related options flipped at random -- every queue's last run against the synchronous run of the same individuals, bit
for bit (per-row values, window tables).  What tools/fuzz_parity.py does with queues of at most six runs on tiny inputs,
here with queues of up to forty on inputs whose kernels take long enough to overlap.

    python tools/stress_queue.py [n_queues] [seed] [rows]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ibdgem_amd import engine as E

n_queues = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = int(sys.argv[3]) if len(sys.argv) > 3 else 40000
N = 300
rng = np.random.default_rng(seed)
f = np.clip(rng.beta(0.3, 1.0, size=L), 1e-3, 0.999)
alle = (rng.random((L, 2 * N)) < f[:, None]).astype(np.uint8)
cov = np.minimum(rng.poisson(2.0, size=L), 20)
na = rng.binomial(cov, f).astype(np.uint8)
nr = (cov - na).astype(np.uint8)
bg = rng.integers(0, 3, size=N).astype(np.uint8)
sets = [[int(t)] for t in rng.choice(N, size=12, replace=False)]
sets += [[int(t) for t in rng.choice(N, size=k, replace=False)] for k in (2, 3, 4, 15, 16, 19, 31, 41, 70, 70)]
bits = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
bad = 0
t0 = time.time()
with E.Engine() as eng:
    eng.upload_panel(E.pack_alleles_fast(alle), N)
    eng.upload_sites(np.arange(L), nr, na, 100)
    want = {}
    for kw_i, kw in enumerate(({}, {"bg_count": bg, "pu_id": 7})):
        for si, s in enumerate(sets):
            eng.run(s, ld=True, **kw)
            want[(kw_i, si)] = (eng.site_ll(len(s) - 1), eng.window_ll_all(len(s)).copy())
    for q in range(n_queues):
        opts = {"prep_ahead": int(rng.integers(0, 2)), "end_in_dispatch": int(rng.integers(0, 2)),
                "finalize_in_next": int(rng.integers(0, 2)), "ibd0_after": int(rng.choice([0, 1, 8])),
                "mfma_targets": int(rng.choice([1, 1, 0])), "multi_target": int(rng.integers(0, 2))}
        for k, v in opts.items():
            eng.set_option(k, v)
        if rng.random() < 0.15:                       # a fresh upload now and then: the ring slots, the passes, the layout start over
            eng.set_option("compact_tiles", int(rng.choice([0, 0, 1, -1])))
            eng.upload_sites(np.arange(L), nr, na, 100)
        eng.set_option("async", 1)
        kw_i = int(rng.integers(0, 2))
        n_runs = int(rng.integers(1, 40))
        for r in range(n_runs):
            if rng.random() < 0.1:
                kw_i ^= 1
            si = int(rng.integers(0, len(sets)))
            eng.run(sets[si], ld=True, **(({}, {"bg_count": bg, "pu_id": 7})[kw_i]))
        s = sets[si]
        site, win = eng.site_ll(len(s) - 1), eng.window_ll_all(len(s))
        eng.set_option("async", 0)
        ws, ww = want[(kw_i, si)]
        # (the --LD columns of groups of 15 and of single runs differ in their last bits by design: compare like with like --
        #  the same kernels serve the same set of individuals whenever mfma_targets / multi_target are at their defaults)
        same_kernels = opts["mfma_targets"] == 1 and opts["multi_target"] == 1
        ok = (bits(site) == bits(ws)).all() and (bits(win[:, :, 2]) == bits(ww[:, :, 2])).all()
        if same_kernels:
            ok = ok and (bits(win[:, :, 0]) == bits(ww[:, :, 0])).all() and (bits(win[:, :, 1]) == bits(ww[:, :, 1])).all()
        else:
            fin = np.isfinite(ww[:, :, :2]) & (ww[:, :, :2] != 0)
            ok = ok and (np.abs(win[:, :, :2][fin] - ww[:, :, :2][fin]) <= 1e-10 * np.abs(ww[:, :, :2][fin])).all()
        if not ok:
            bad += 1
            print("MISMATCH queue", q, "runs", n_runs, "last set", si, len(s), "bg", kw_i, opts, flush=True)
        if q % 20 == 19:
            print(f"... {q + 1} queues, {bad} bad, {time.time() - t0:.0f}s", flush=True)
print(f"stress_queue: {n_queues} queues, {bad} failures, {time.time() - t0:.0f}s")
sys.exit(1 if bad else 0)
