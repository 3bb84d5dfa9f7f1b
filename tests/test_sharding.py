"""Multi-GPU path on CPU: window-range sharding with world_size 2 over gloo.

Each rank takes its rows from ibdgem_amd.sharding.shard_rows, evaluates them on its own (here:
with the oracle, the GPU is not available in this tier), the per-window rows are gathered in
rank order and must equal the unsharded evaluation bit for bit -- i.e. cutting at window
boundaries needs no exchange between ranks."""
import os
import socket

import numpy as np
import pytest

from ibdgem_amd.sharding import shard_rows, windows_per_shard


def _synth(seed, L, N):
    rng = np.random.default_rng(seed)
    f = np.clip(rng.beta(0.3, 1.0, size=L), 1e-3, 0.999)
    alle = (rng.random((L, 2 * N)) < f[:, None]).astype(np.uint8)
    cov = np.minimum(rng.poisson(2.0, size=L), 20)
    n_alt = rng.binomial(cov, f)
    return alle, (cov - n_alt).astype(np.uint8), n_alt.astype(np.uint8)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("L,W", [(1000, 100), (1013, 37), (50, 100), (5, 2)])
def test_cuts_fall_on_window_boundaries(world, L, W):
    _, nr, na = _synth(L + W, L, 4)
    cuts = shard_rows(nr, na, W, world)
    assert cuts[0] == 0 and cuts[-1] == L and all(a <= b for a, b in zip(cuts, cuts[1:]))
    covered = (nr.astype(int) + na) > 0
    n_win = (covered.sum() + W - 1) // W
    per = windows_per_shard(nr, na, W, cuts)
    assert sum(per) == n_win and max(per) - min(per) <= 1
    for c in cuts[1:-1]:
        if 0 < c < L:
            assert covered[:c].sum() % W == 0 and covered[c - 1]        # a window just closed at the cut


def _worker(rank, world, port, q):
    import torch.distributed as dist
    import oracle_lib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    orc = oracle_lib.Oracle(os.path.join(repo, "oracle", "liboracle.so"))
    alle, nr, na = _synth(99, 1500, 70)                # every rank derives the same cut points
    cuts = shard_rows(nr, na, 100, world)
    a, b = cuts[rank], cuts[rank + 1]
    res = orc.compare(alle[a:b], nr[a:b], na[a:b], 5, window=100, ld=True, pu_id=9)
    mine = dict(win=res["win"], first=res["first"] + a, last=res["last"] + a, nsites=res["nsites"],
                site=res["site"])
    dist.barrier()
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)             # host-side gather of the result rows
    if rank == 0:
        full = orc.compare(alle, nr, na, 5, window=100, ld=True, pu_id=9)
        ok = True
        for key in ("win", "first", "last", "nsites", "site"):
            cat = np.concatenate([g[key] for g in gathered])
            ok = ok and cat.shape == full[key].shape and (cat.view(np.uint8) == full[key].view(np.uint8)).all()
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_over_gloo_reproduce_the_unsharded_result(oracle):
    import multiprocessing as mp      # the parent stays torch-free; the two ranks import torch themselves
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok
