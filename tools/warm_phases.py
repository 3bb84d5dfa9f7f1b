"""Phase clocks of the host program on the warm 4M-row run (packed-panel cache + text -> summary file), three runs:
    python tools/warm_phases.py   (on a GPU box)"""
import os, sys, tempfile, subprocess, time, struct
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
rows = 4_000_000
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
words = panel.cpu().numpy().view(np.uint64)
del panel
exe = os.path.join(bench.REPO, "ibdgem_amd", "host", "ibdgem")
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    bench.write_pileup_and_legend(d, n_ref, n_alt, 2504, rows)
    open(os.path.join(d, "p.hap"), "w").write("placeholder\n")
    st = os.stat(os.path.join(d, "p.hap"))
    bench.write_panel_cache(os.path.join(d, "p.cache"), words, 2504, st)
    base = [exe, "-H", "p.hap", "-L", "p.legend", "-I", "p.indv", "-P", "p.pileup", "-s", "ind7", "--LD", "--threads", "16", "--panel-cache", "p.cache", "-O", "o"]
    os.makedirs(os.path.join(d, "o"))
    for label, extra_env, extra in (("--summary-only, default (_exit once the files are closed)", {}, ["--summary-only"]),
                                    ("--summary-only, IBDGEM_CACHE_MAP=1 (the panel cache mapped and uploaded from the mapping, as until round 4)", {"IBDGEM_CACHE_MAP": "1"}, ["--summary-only"]),
                                    ("--summary-only, default again", {}, ["--summary-only"]),
                                    ("--summary-only, IBDGEM_CACHE_MAP=1 again", {"IBDGEM_CACHE_MAP": "1"}, ["--summary-only"]),
                                    ("--summary-only, IBDGEM_KEEP_TEARDOWN=1 (ibdg_destroy, orderly exit)", {"IBDGEM_KEEP_TEARDOWN": "1"}, ["--summary-only"]),
                                    ("with the per-site table, default", {}, []),
                                    ("with the per-site table, IBDGEM_KEEP_TEARDOWN=1", {"IBDGEM_KEEP_TEARDOWN": "1"}, []),
                                    ("with the per-site table, one host thread", {}, ["--threads", "1"]),
                                    ("--summary-only again", {}, ["--summary-only"])):
        print(label)
        for rep in range(4):
            t0 = time.perf_counter()
            r = subprocess.run(base + extra, cwd=d, env=dict(os.environ, IBDGEM_TIMING="1", **extra_env), capture_output=True, text=True)
            dt = time.perf_counter() - t0
            ph = [l[8:] for l in r.stderr.splitlines() if l.startswith("## time")]
            covered = sum(float(x.rsplit(" ", 1)[1]) for x in ph)
            print(f"  {dt:.3f} s wall, {covered:.3f} s in phases, {dt - covered:.3f} s outside |", " | ".join(ph[-3:] if os.environ.get("SHORT") else ph))
