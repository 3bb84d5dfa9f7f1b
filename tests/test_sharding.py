"""Multi-GPU path: window-range sharding with world_size 2 over gloo.

Each rank takes its rows from ibdgem_amd.sharding.shard_rows, evaluates them on its own, the
per-window rows are gathered in rank order and must equal the unsharded evaluation bit for bit --
i.e. cutting at window boundaries needs no exchange between ranks.  In the CPU tier the ranks
evaluate with the oracle (no device there); in the `-m gpu` tier the same worker drives
ibdgem_amd.Engine (one context per rank, both on the box's one GPU), and rank 0 also checks the
gathered result against the oracle."""
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


def _evaluate(use_engine, orc, alle, nr, na, target, pu_id):
    """One rank's rows (or the whole chromosome): per-site and per-window results with row numbers local."""
    if not use_engine:
        return orc.compare(alle, nr, na, target, window=100, ld=True, pu_id=pu_id)
    import torch
    torch.cuda.is_available()                          # torch's HIP runtime first, as in tests/conftest.py
    import ibdgem_amd
    from ibdgem_amd.engine import pack_alleles_fast
    with ibdgem_amd.Engine(0, 0.02, 20) as eng:        # this rank's context and its own slice of the panel
        eng.upload_panel(pack_alleles_fast(alle), alle.shape[1] // 2)
        eng.upload_sites(None, nr, na, 100)
        eng.run([target], ld=True, pu_id=pu_id)
        first, last, ncov = eng.windows()
        return dict(win=eng.window_ll(0), site=eng.site_ll(0), first=first, last=last, nsites=ncov)


def _worker(rank, world, port, q, use_engine):
    import torch.distributed as dist
    import oracle_lib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    orc = oracle_lib.Oracle(os.path.join(repo, "oracle", "liboracle.so"))
    L, N = (6000, 200) if use_engine else (1500, 70)
    alle, nr, na = _synth(99, L, N)                    # every rank derives the same cut points
    cuts = shard_rows(nr, na, 100, world)
    a, b = cuts[rank], cuts[rank + 1]
    res = _evaluate(use_engine, orc, alle[a:b], nr[a:b], na[a:b], 5, 9)
    mine = dict(win=res["win"], first=res["first"] + a, last=res["last"] + a, nsites=res["nsites"],
                site=res["site"])
    dist.barrier()
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)             # host-side gather of the result rows
    if rank == 0:
        full = _evaluate(use_engine, orc, alle, nr, na, 5, 9)
        ok = True
        for key in ("win", "first", "last", "nsites", "site"):
            cat = np.concatenate([g[key] for g in gathered])
            ok = ok and cat.shape == full[key].shape and (cat.view(np.uint8) == full[key].view(np.uint8)).all()
        if use_engine:                                 # and the gathered shards against the CPU checker
            ref = orc.compare(alle, nr, na, 5, window=100, ld=True, pu_id=9)
            win = np.concatenate([g["win"] for g in gathered])
            site = np.concatenate([g["site"] for g in gathered])
            ok = ok and (site.view(np.uint64) == ref["site"].view(np.uint64)).all()
            ok = ok and (win[:, 2].view(np.uint64) == ref["win"][:, 2].view(np.uint64)).all()
            ok = ok and float((np.abs(win[:, :2] - ref["win"][:, :2]) / np.abs(ref["win"][:, :2])).max()) <= 1e-10
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(use_engine):
    import multiprocessing as mp      # the parent stays torch-free; the two ranks import torch themselves
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, use_engine)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert ok


def test_two_ranks_over_gloo_reproduce_the_unsharded_result(oracle):
    _run_two_ranks(use_engine=False)


@pytest.mark.gpu
def test_two_ranks_over_gloo_engine_shards_reproduce_the_unsharded_result(oracle):
    """The same decomposition with the engine doing the work: two processes, one ibdg context each on the
    box's GPU, host-side gather over gloo; bit-identical to one unsharded engine run and within the
    parity bars of the oracle."""
    _run_two_ranks(use_engine=True)
