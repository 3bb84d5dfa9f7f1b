"""Window-range sharding of one comparison across ranks (one rank per GPU).

Windows are independent units: the reference resets all window state at every window start
(reference src/ibdgem.c:558-570), so a chromosome can be cut at window boundaries and every
piece evaluated on its own device with no exchange; the per-window rows are concatenated in
rank order (host-side gather, no collective on the data path)."""
import numpy as np


def shard_rows(n_ref, n_alt, window, world):
    """Row cut points [c_0=0, c_1, ..., c_world=L] such that every rank's rows hold a whole
    number of windows (a window = `window` consecutive covered rows, covered = n_ref+n_alt > 0;
    uncovered rows stay with the window that is open when they are read, src/ibdgem.c:657-663)
    and window counts differ by at most one between ranks."""
    n_ref = np.asarray(n_ref)
    n_alt = np.asarray(n_alt)
    covered = (n_ref.astype(np.int32) + n_alt) > 0
    csum = np.cumsum(covered)
    total = int(csum[-1]) if len(csum) else 0
    n_win = (total + window - 1) // window
    cuts = [0]
    for r in range(1, world):
        w = (n_win * r) // world                    # windows before this cut
        if w == 0:
            cuts.append(0)
        elif w * window >= total:
            cuts.append(len(covered))
        else:
            # the cut goes right after the (w*window)-th covered row
            cuts.append(int(np.searchsorted(csum, w * window, side="left")) + 1)
    cuts.append(len(covered))
    return cuts


def windows_per_shard(n_ref, n_alt, window, cuts):
    covered = (np.asarray(n_ref).astype(np.int32) + np.asarray(n_alt)) > 0
    out = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        c = int(covered[a:b].sum())
        out.append((c + window - 1) // window)
    return out
