#!/usr/bin/env python3
"""bench.py -- SNP-sites/s of the --LD hot path on synthetic chr1-scale data.

Workload (BASELINE.json configs[3], the config the metric is quoted on):
  --LD, ~4M SNP rows, 2504-individual phased panel, window 100, 1 comparison
  individual, epsilon 0.02, max-cov 20; Poisson(2) read depth so ~13.5% of the
  rows have no informative read (printed, not windowed -- src/ibdgem.c:657-663).
A "step" = one full pass over all rows: per-site LIBD0/1/2 (k_site), the --LD
background loop + window averages (k_win_target, k_ld_popcount, k_ld_finalize)
and the window products (k_window_prod); inputs resident in HBM (the packed
panel in its two layouts and the per-row alt-allele counts, all produced once
by the panel upload -- they depend on the panel only, like the reference's -A
file), results left in HBM.  value = windowed sites processed by all ranks /
max-over-ranks wall time per step.  The JSON also carries the cost of
recomputing the alt counts inside the step ("alt_count_ms", "value_with_recount").

With --gpus N the SAME chromosome is cut into N contiguous window ranges, one
per rank (strong scaling; no data-path collective -- windows are independent,
src/ibdgem.c:558-570; torch.distributed is used only for the barrier and the
max-over-ranks reduction of the timing).

One JSON line is printed by rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BLOCK_ROWS = 1 << 16           # generation granule; data depend on the global row only


def algorithmic_bytes_per_site(n_ids, n_targets):
    """packed panel row N/4 + n_ref,n_alt 2 B + alt count 2 B, and per
    target 24 B of per-site output + 24 B/window."""
    return n_ids / 4.0 + 4.0 + n_targets * 24.24


# ----------------------------------------------------------------------------- synthetic data (GPU, torch = plumbing)
def gen_block(torch, dev, block, n_ids, target, seed):
    """Rows [block*BLOCK_ROWS, +BLOCK_ROWS): packed panel words, n_ref, n_alt.  Deterministic in
    (seed, block), independent of how blocks are spread over ranks."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed * 1000003 + block)
    R = BLOCK_ROWS
    chunks = (n_ids + 63) // 64
    u = torch.rand(R, generator=g, device=dev, dtype=torch.float64)
    f = torch.clamp(u ** (1.0 / 0.3), 1e-3, 0.999).to(torch.float32)      # Beta(0.3, 1)
    bits = torch.rand(R, chunks * 64, 2, generator=g, device=dev) < f[:, None, None]
    bits[:, n_ids:, :] = False
    # [R][chunk][bit][plane] -> word[2*chunk+plane] bit
    b = bits.view(R, chunks, 64, 2).permute(0, 1, 3, 2).to(torch.int64)
    weights = (torch.ones(64, dtype=torch.int64, device=dev) << torch.arange(64, device=dev))
    words = (b * weights).sum(dim=-1).reshape(R, chunks * 2).contiguous()
    a0 = bits[:, target, 0].to(torch.float32)
    a1 = bits[:, target, 1].to(torch.float32)
    cov = torch.poisson(torch.full((R,), 2.0, device=dev), generator=g).clamp_(max=20)
    p_alt = (a0 + a1) * 0.5 * (1 - 2 * 0.02) + 0.02          # g=0: eps, g=1: 0.5, g=2: 1-eps
    n_alt = torch.binomial(cov, p_alt, generator=g)
    n_ref = cov - n_alt
    return words, n_ref.to(torch.uint8), n_alt.to(torch.uint8)


def build_shard(torch, dev, row0, row1, n_ids, target, seed):
    """Panel words (device tensor) and read counts (host arrays) for global rows [row0,row1)."""
    words, nrs, nas = [], [], []
    b0, b1 = row0 // BLOCK_ROWS, (row1 + BLOCK_ROWS - 1) // BLOCK_ROWS
    for blk in range(b0, b1):
        w, nr, na = gen_block(torch, dev, blk, n_ids, target, seed)
        lo = max(row0, blk * BLOCK_ROWS) - blk * BLOCK_ROWS
        hi = min(row1, (blk + 1) * BLOCK_ROWS) - blk * BLOCK_ROWS
        words.append(w[lo:hi])
        nrs.append(nr[lo:hi].cpu().numpy())
        nas.append(na[lo:hi].cpu().numpy())
    panel = torch.cat(words, dim=0).contiguous()
    return panel, np.concatenate(nrs), np.concatenate(nas)


# ----------------------------------------------------------------------------- CPU baseline (reference binary)
def unpack_rows(words, n_ids):
    """packed uint64 [L][2*chunks] -> alleles uint8 [L][2*n_ids] ([2n]=first, [2n+1]=second)."""
    L = words.shape[0]
    chunks = words.shape[1] // 2
    by = words.view(np.uint8).reshape(L, chunks, 2, 8)
    bits = np.unpackbits(by, axis=-1, bitorder="little")            # [L][chunk][plane][64]
    return np.ascontiguousarray(bits.transpose(0, 1, 3, 2).reshape(L, chunks * 64, 2)[:, :n_ids, :]).reshape(L, 2 * n_ids)


def write_reference_inputs(d, alleles, n_ref, n_alt, n_ids):
    L = alleles.shape[0]
    hap = np.full((L, 4 * n_ids), ord(" "), dtype=np.uint8)
    hap[:, 0::2] = alleles + ord("0")
    hap[:, -1] = ord("\n")
    hap.tofile(os.path.join(d, "p.hap"))
    with open(os.path.join(d, "p.legend"), "w") as fh:
        fh.write("ID pos allele0 allele1\n")
        fh.write("".join(f"rs{i} {100 + 10 * i} A C\n" for i in range(L)))
    with open(os.path.join(d, "p.indv"), "w") as fh:
        fh.write("".join(f"ind{n}\n" for n in range(n_ids)))
    with open(os.path.join(d, "p.pileup"), "w") as fh:
        out = []
        for i in range(L):
            r, a = int(n_ref[i]), int(n_alt[i])
            c = r + a
            if c == 0:
                out.append(f"1\t{100 + 10 * i}\tN\t0\t*\t*\t*\n")
            else:
                out.append(f"1\t{100 + 10 * i}\tN\t{c}\t{'A' * r}{'C' * a}\t{'I' * c}\t{'I' * c}\n")
        fh.write("".join(out))


def cpu_baseline(words_host, n_ref, n_alt, n_ids, target, window, gpu_win):
    """Time the reference binary (oracle/_ref/ibdgem, unmodified sources, -O0 as shipped) on the
    first rows of the same workload, 1 thread; also check its summary file against the GPU's
    windows for those rows (7 printed digits)."""
    exe = os.path.join(REPO, "oracle", "_ref", "ibdgem")
    L = words_host.shape[0]
    n_cov = int(((n_ref.astype(np.int32) + n_alt) > 0).sum())
    if not os.path.exists(exe):
        # the reference binary did not travel: time our CPU restatement instead and say so
        sys.path.insert(0, os.path.join(REPO, "tests"))
        import oracle_lib
        so = os.path.join(REPO, "oracle", "liboracle.so")
        if not os.path.exists(so):
            subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "liboracle.so"], check=True,
                           stdout=subprocess.DEVNULL)
        orc = oracle_lib.Oracle(so)
        alle = unpack_rows(words_host, n_ids)
        t0 = time.perf_counter()
        orc.compare(alle, n_ref, n_alt, target, window=window, ld=True)
        dt = time.perf_counter() - t0
        return dict(value=n_cov / dt, unit="sites/s", cores=1, kind="port",
                    sample=f"first {L} rows ({n_cov} windowed) of the workload, LD loop only (no parsing), oracle/liboracle.so")
    alle = unpack_rows(words_host, n_ids)
    with tempfile.TemporaryDirectory(dir=os.environ.get("TMPDIR", "/tmp")) as d:
        write_reference_inputs(d, alle, n_ref, n_alt, n_ids)
        base = [exe, "-H", "p.hap", "-L", "p.legend", "-I", "p.indv", "-P", "p.pileup", "-s", f"ind{target}",
                "-O", d]
        times = {}
        cpu0 = min(os.sched_getaffinity(0))             # one process pinned to one core

        def pin():
            os.sched_setaffinity(0, {cpu0})

        for mode, extra in (("ld", ["--LD"]), ("nonld", [])):
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                subprocess.run(base + extra, cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               preexec_fn=pin)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            times[mode] = best
            if mode == "ld":
                rows = [l.split("\t") for l in open(os.path.join(d, f"UNKWN.ind{target}.summary.txt"))
                        if not l.startswith("#")]
        # the same sources at -O2 (labelled; the north-star ratio uses the build as shipped)
        exe_o2 = os.path.join(REPO, "oracle", "_ref", "ibdgem_O2")
        o2 = None
        if os.path.exists(exe_o2):
            t0 = time.perf_counter()
            subprocess.run([exe_o2] + base[1:] + ["--LD"], cwd=d, check=True, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, preexec_fn=pin)
            o2 = n_cov / (time.perf_counter() - t0)
        parity = None
        if gpu_win is not None:
            ok = len(rows) <= len(gpu_win)
            for w, r in enumerate(rows[:-1]):          # the sample's last window is cut short by the slice
                ok = ok and ["%e" % v for v in gpu_win[w]] == r[3:6]
            parity = bool(ok)
    ld_stage = max(times["ld"] - times["nonld"], 1e-9)
    return dict(value=n_cov / times["ld"], unit="sites/s", cores=1, kind="reference",
                sample=(f"first {L} rows ({n_cov} windowed) of the same workload as reference text inputs, "
                        f"unmodified reference built -O0 as shipped, 1 process pinned to 1 core, best of 2, end-to-end --LD run "
                        f"{times['ld']:.2f}s; non-LD {times['nonld']:.2f}s"),
                ld_stage_only_sites_per_s=n_cov / ld_stage,
                O2_rebuild_sites_per_s=o2,
                summary_matches_gpu_7digits=parity)


def traffic_bytes(args, world):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/*_ld_traffic.json, tools/pmc_traffic.sh) when they were taken on this very
    workload; None otherwise (counters cannot be read from inside the process)."""
    best = None
    pdir = os.path.join(REPO, "profiles")
    for fn in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        if not fn.endswith("_ld_traffic.json"):
            continue
        with open(os.path.join(pdir, fn)) as fh:
            t = json.load(fh)
        c = t.get("config", {})
        if world == 1 and (c.get("sites"), c.get("n_ids"), c.get("window")) == (args.sites, args.ids, args.window):
            best = t.get("dominant_kernel_hbm_bytes_per_launch")
    return best


# ----------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)       # ~0.9 ms each; the clocks settle after ~30 steps
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--sites", type=int, default=4_000_000, help="SNP rows of the synthetic chromosome")
    ap.add_argument("--ids", type=int, default=2504)
    ap.add_argument("--window", type=int, default=100)
    ap.add_argument("--target", type=int, default=7)
    ap.add_argument("--seed", type=int, default=20241008)
    ap.add_argument("--cpu-sample-rows", type=int, default=100_000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--variant", type=int, default=None, help="ld_variant option of the engine")
    ap.add_argument("--cpw", type=int, default=None)
    ap.add_argument("--waves", type=int, default=None)
    ap.add_argument("--sync-steps", action="store_true", help="wait for the device after every step")
    ap.add_argument("--opt", action="append", default=[], help="engine option name=value (repeatable)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ibdgem_amd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU path)")
    # BENCH_FORCE_DEVICE / BENCH_BACKEND exist only to rehearse the multi-rank path on a one-GPU box
    # (all ranks on one device, gloo instead of RCCL); the driver's runs use neither.
    if os.environ.get("BENCH_FORCE_DEVICE") is not None:
        local = int(os.environ["BENCH_FORCE_DEVICE"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    # Every rank derives the same global cut points from the (cheap) read counts of all rows;
    # only its own panel shard is materialised.
    L = args.sites
    if world > 1:
        nr_all, na_all = [], []
        for blk in range((L + BLOCK_ROWS - 1) // BLOCK_ROWS):
            g = torch.Generator(device=dev)
            # read counts need the target's alleles, i.e. the block's bits: generate and drop
            w, nr, na = gen_block(torch, dev, blk, args.ids, args.target, args.seed)
            nr_all.append(nr.cpu().numpy())
            na_all.append(na.cpu().numpy())
            del w
        nr_all = np.concatenate(nr_all)[:L]
        na_all = np.concatenate(na_all)[:L]
        from ibdgem_amd.sharding import shard_rows
        cuts = shard_rows(nr_all, na_all, args.window, world)
        row0, row1 = cuts[rank], cuts[rank + 1]
    else:
        row0, row1 = 0, L
    panel, n_ref, n_alt = build_shard(torch, dev, row0, row1, args.ids, args.target, args.seed)
    torch.cuda.synchronize()

    eng = ibdgem_amd.Engine(local, 0.02, 20)
    if args.variant is not None:
        eng.set_option("ld_variant", args.variant)
    if args.cpw is not None:
        eng.set_option("chunks_per_wave", args.cpw)
    if args.waves is not None:
        eng.set_option("waves_per_block", args.waves)
    for kv in args.opt:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    eng.upload_panel_dev(panel.data_ptr(), panel.shape[0], args.ids)
    sample_rows = min(args.cpu_sample_rows, panel.shape[0])
    sample_words = panel[:sample_rows].cpu().numpy().view(np.uint64) if rank == 0 else None
    del panel
    torch.cuda.empty_cache()
    n_rows = row1 - row0
    eng.upload_sites(np.arange(n_rows, dtype=np.uint32), n_ref, n_alt, args.window)
    n_cov = int(((n_ref.astype(np.int32) + n_alt) > 0).sum())
    targets = [args.target]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        eng.sync()

    for _ in range(args.warmup):
        eng.run(targets, ld=True)
    # The timed steps are queued back to back (the engine's "async" option: ibdg_run returns once
    # its kernels are enqueued, like any stream-ordered step loop) and the closing barrier waits
    # for all of them; each step's HIP events are read afterwards (the engine keeps the last 32).
    eng.set_option("async", 0 if args.sync_steps else 1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.run(targets, ld=True)
    dt_host = time.perf_counter() - t0          # host time to queue the steps (reported only)
    barrier()
    dt = time.perf_counter() - t0
    eng.set_option("async", 0)
    ms_all = [eng.run_ms(b) for b in range(min(args.steps, 32))]
    ms_ld = [m["ld"] for m in ms_all]

    # cost of recounting the alt alleles inside the step (reported, not part of `value`)
    eng.set_option("count_in_run", 1)
    eng.run(targets, ld=True)
    t1 = time.perf_counter()
    for _ in range(3):
        eng.run(targets, ld=True)
    eng.sync()
    dt_recount = (time.perf_counter() - t1) / 3
    alt_ms = eng.last_run_ms()["alt_count"]
    eng.set_option("count_in_run", 0)
    # the dominant kernel alone: 32 more queued steps timed through the kernel's own dispatch packet
    # (hipExtLaunchKernel start/stop events; slower per step than one event record, hence not in the
    # timed region above)
    ms_kernel = None
    try:
        eng.set_option("dispatch_events", 1)
        eng.set_option("async", 1)
        for _ in range(40):
            eng.run(targets, ld=True)
        eng.sync()
        ms_kernel = [eng.run_kernel_ms(b) for b in range(32)]
    except ibdgem_amd.EngineError:
        pass                                 # strict kernel: no such figure
    eng.set_option("async", 0)
    eng.set_option("dispatch_events", 0)
    # the survey's "engine clock": one run plus its results copied to host memory (not `value`)
    eng.run(targets, ld=True)
    t2 = time.perf_counter()
    eng.run(targets, ld=True)
    w_host = eng.window_ll(0)
    t3 = time.perf_counter()
    s_host = eng.site_ll(0)
    t4 = time.perf_counter()
    d2h = {"run_plus_window_results_ms": (t3 - t2) * 1e3, "per_site_results_ms": (t4 - t3) * 1e3,
           "per_site_results_GBps": s_host.nbytes / (t4 - t3) / 1e9}
    del w_host, s_host
    ld_variant = eng.last_ld_variant()

    tot = torch.tensor([dt, float(n_cov), float(n_rows)], dtype=torch.float64,
                       device=dev if backend == "nccl" else "cpu")
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt_max, cov_total, rows_total = float(mx[0]), float(sm[1]), float(sm[2])
    else:
        dt_max, cov_total, rows_total = dt, float(n_cov), float(n_rows)

    if rank == 0:
        ms_step = dt_max / args.steps * 1e3
        value = cov_total / (dt_max / args.steps)
        b_site = algorithmic_bytes_per_site(args.ids, len(targets))
        ld_ms = float(np.mean(ms_ld))
        achieved = b_site * n_cov / (ld_ms * 1e-3) / 1e9
        kern = {k: float(np.mean([m[k] for m in ms_all])) for k in ms_all[0]}
        out = {
            "metric": "SNP-sites/sec in --LD mode, chr1, 2504-indiv panel",
            "value": value, "unit": "sites/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"--LD, {L} SNP rows (synthetic chr1), {args.ids}-individual phased panel, "
                                   f"window {args.window}, 1 comparison individual (BASELINE.json configs[3])",
                       "rows": int(rows_total), "windowed_sites": int(cov_total), "n_ids": args.ids,
                       "window": args.window, "targets": len(targets), "epsilon": 0.02, "max_cov": 20,
                       "sharding": f"{world} contiguous window ranges, no collective on the data path"},
            "roofline": {"bound": "hbm", "kernel": "k_ld_popcount" if ld_variant == 2 else "k_ld_window",
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_bytes(args, world),
                         "bytes_per_site": b_site, "sites_per_launch": n_cov, "launch_ms": ld_ms,
                         "launch_ms_note": "HIP events on the engine's stream around the --LD launches "
                                           "(k_win_target + k_ld_popcount + k_ld_finalize), mean over the last "
                                           f"{len(ms_ld)} timed steps",
                         "dominant_kernel_only_ms": float(np.mean(ms_kernel)) if ms_kernel else None,
                         "dominant_kernel_only_note": "k_ld_popcount alone, start/stop events of its own dispatch "
                                                      "packet, 32 extra steps after the timed region"},
            "kernel_ms": kern,
            "host_queue_ms_per_step": dt_host / args.steps * 1e3,
            "alt_count_ms": alt_ms,
            "results_to_host": d2h if world == 1 else None,
            "value_with_recount": n_cov / dt_recount if world == 1 else None,
            "rows_per_s_all_processed": rows_total / (dt_max / args.steps),
        }
        if not args.no_cpu_baseline and world == 1:       # the CPU leg is timed at N=1 only
            s = sample_rows
            eng.upload_sites(np.arange(s, dtype=np.uint32), n_ref[:s], n_alt[:s], args.window)
            eng.run(targets, ld=True)
            out["cpu_baseline"] = cpu_baseline(sample_words, n_ref[:s], n_alt[:s], args.ids, args.target,
                                               args.window, eng.window_ll(0))
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
