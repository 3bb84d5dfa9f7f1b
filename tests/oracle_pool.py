"""Check EVERY window of a large comparison against the oracle with a pool of host processes.

The GPU test keeps the packed panel rows, the read counts and the engine's results in .npy files under
/dev/shm; each worker (a spawned process that never touches the GPU) memory-maps them, takes a range of
whole windows, unpacks its rows, runs oracle/liboracle.so on them and compares: per-site values and
LIBD2 bit for bit, --LD LIBD0/LIBD1 to the relative tolerance given.  Test infrastructure only.
"""
import os
import sys

import numpy as np

TINY = 1e-290
HERE = os.path.dirname(os.path.abspath(__file__))


def unpack_rows(words, n_ids):
    """packed uint64 [L][2*chunks] -> alleles uint8 [L][2*n_ids] ([2n] first, [2n+1] second haplotype)."""
    L = words.shape[0]
    chunks = words.shape[1] // 2
    by = np.ascontiguousarray(words).view(np.uint8).reshape(L, chunks, 2, 8)
    bits = np.unpackbits(by, axis=-1, bitorder="little")
    return np.ascontiguousarray(bits.transpose(0, 1, 3, 2).reshape(L, chunks * 64, 2)[:, :n_ids, :]).reshape(L, 2 * n_ids)


def check_block(job):
    """One range of windows [w0, w1).  Returns (windows checked, max rel err, list of complaints)."""
    (d, n_ids, target, window, pu_id, eps, max_cov, rtol, w0, w1, row0, row1) = job
    sys.path.insert(0, HERE)
    import oracle_lib
    orc = oracle_lib.Oracle(os.path.join(os.path.dirname(HERE), "oracle", "liboracle.so"))
    panel = np.load(os.path.join(d, "panel.npy"), mmap_mode="r")
    nr = np.load(os.path.join(d, "n_ref.npy"), mmap_mode="r")[row0:row1]
    na = np.load(os.path.join(d, "n_alt.npy"), mmap_mode="r")[row0:row1]
    win = np.load(os.path.join(d, "win.npy"), mmap_mode="r")[w0:w1]
    site = np.load(os.path.join(d, "site.npy"), mmap_mode="r")[row0:row1]
    alle = unpack_rows(np.asarray(panel[row0:row1]), n_ids)
    res = orc.compare(alle, nr, na, target, window=window, eps=eps, max_cov=max_cov, ld=True, pu_id=pu_id)
    bad = []
    if len(res["win"]) != w1 - w0:
        return 0, 0.0, [f"windows {w0}..{w1}: the oracle finds {len(res['win'])} windows in rows {row0}..{row1}"]
    u = lambda a: np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)
    if not (u(site) == u(res["site"])).all():
        bad.append(f"windows {w0}..{w1}: per-site values differ from the oracle")
    if not (u(win[:, 2]) == u(res["win"][:, 2])).all():
        bad.append(f"windows {w0}..{w1}: LIBD2 differs from the oracle")
    got, want = np.asarray(win[:, :2]), res["win"][:, :2]
    tiny = np.abs(want) < TINY
    if not (np.abs(got[tiny]) < TINY).all():
        bad.append(f"windows {w0}..{w1}: tiny values")
    rel = np.abs(got[~tiny] - want[~tiny]) / np.abs(want[~tiny])
    worst = float(rel.max()) if rel.size else 0.0
    if worst > rtol:
        bad.append(f"windows {w0}..{w1}: --LD columns off by {worst:.3e}")
    return w1 - w0, worst, bad


def check_all_windows(d, first, n_rows, n_ids, target, window, *, pu_id=-1, eps=0.02, max_cov=20, rtol=1e-10,
                      windows_per_job=128, workers=None):
    """first[w] = row of the first covered site of window w (the engine's).  Blocks of windows_per_job
    windows: rows [first[w0], first[w1]) -- the rows without reads behind a window's last covered row
    travel with it, as in the reference's loop (src/ibdgem.c:657-663)."""
    import multiprocessing as mp
    n_win = len(first)
    jobs = []
    for w0 in range(0, n_win, windows_per_job):
        w1 = min(n_win, w0 + windows_per_job)
        row0 = int(first[w0]) if w0 else 0
        row1 = int(first[w1]) if w1 < n_win else n_rows
        jobs.append((d, n_ids, target, window, pu_id, eps, max_cov, rtol, w0, w1, row0, row1))
    if workers is None:
        workers = max(1, min(14, len(os.sched_getaffinity(0)) - 1))
    with mp.get_context("spawn").Pool(workers) as pool:
        out = pool.map(check_block, jobs, chunksize=1)
    checked = sum(o[0] for o in out)
    worst = max([o[1] for o in out], default=0.0)
    bad = [m for o in out for m in o[2]]
    return checked, worst, bad
