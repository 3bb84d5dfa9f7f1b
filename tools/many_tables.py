"""Eight comparison individuals with their per-site tables (the default run of the host program), for several numbers of
individuals whose files are written at once (IBDGEM_OUT_SLOTS) and formatter threads per individual (IBDGEM_OUT_THREADS):
    python tools/many_tables.py   (on a GPU box)"""
import os, sys, tempfile, subprocess, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
rows = 4_000_000
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
words = panel.cpu().numpy().view(np.uint64)
del panel
torch.cuda.empty_cache()
exe = os.path.join(bench.REPO, "ibdgem_amd", "host", "ibdgem")
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    bench.write_pileup_and_legend(d, n_ref, n_alt, 2504, rows)
    open(os.path.join(d, "p.hap"), "w").write("placeholder\n")
    st = os.stat(os.path.join(d, "p.hap"))
    bench.write_panel_cache(os.path.join(d, "p.cache"), words, 2504, st)
    del words
    n_ind = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    names = ",".join(f"ind{7 + 5 * i}" for i in range(n_ind))
    base = [exe, "-H", "p.hap", "-L", "p.legend", "-I", "p.indv", "-P", "p.pileup", "-s", names, "--LD", "--threads", "16",
            "--panel-cache", "p.cache", "-O", "o"]
    os.makedirs(os.path.join(d, "o"))
    sweep = ((1, 16), (2, 8), (3, 8), (3, 16), (4, 8), (4, 4), (6, 4), (6, 8), (3, 6)) if n_ind == 8 else ((1, 16), (4, 8), (6, 8))
    for slots, thr in sweep:
        best = None
        best_ph = None
        for rep in range(3):
            r = subprocess.run(base, cwd=d, env=dict(os.environ, IBDGEM_TIMING="1", IBDGEM_OUT_SLOTS=str(slots), IBDGEM_OUT_THREADS=str(thr)),
                               capture_output=True, text=True)
            ph = {}
            for l in r.stderr.splitlines():
                if l.startswith("## time "):
                    k, v = l[8:].rsplit(" ", 1)
                    ph[k] = ph.get(k, 0.0) + float(v)
            own = sum(v for k, v in ph.items() if k.startswith("per individual: engine") or k.startswith("per individual: output")
                      or k.startswith("per individual: waiting") or k.startswith("output files of the last"))
            if best is None or own < best:
                best, best_ph = own, ph
        print(f"{n_ind} individuals, files of {slots} at once, {thr} formatter threads each: {best / n_ind * 1e3:.1f} ms per individual (engine + output, best of 3)", flush=True)
        if n_ind != 8:
            print("    " + " | ".join(f"{k[:45]} {v:.3f}" for k, v in best_ph.items() if "individual" in k or "columns" in k), flush=True)
