"""Many comparison individuals with --summary-only (the whole-panel job of the reference's use: one pileup against every
individual of the panel): engine + output per individual from the host program's phase clocks, and the wall clock.
    python tools/many_summaries.py [individuals] [IBDGEM_OUT_SLOTS values ...]   (on a GPU box)"""
import os, sys, tempfile, subprocess, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
rows = 4_000_000
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
if os.environ.get("THIN"):                      # reads on one row in THIN only (a thin pileup: the reference's real inputs)
    keep = np.random.default_rng(5).random(rows) < 1.0 / float(os.environ["THIN"])
    n_ref = np.where(keep, n_ref, 0).astype(n_ref.dtype)
    n_alt = np.where(keep, n_alt, 0).astype(n_alt.dtype)
words = panel.cpu().numpy().view(np.uint64)
del panel
torch.cuda.empty_cache()
exe = os.environ.get("IBDGEM_EXE") or os.path.join(bench.REPO, "ibdgem_amd", "host", "ibdgem")
n_ind = int(sys.argv[1]) if len(sys.argv) > 1 else 240
slot_list = [int(a) for a in sys.argv[2:]] or [0, 1]          # 0: the program's own choice (12 with --summary-only)
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    bench.write_pileup_and_legend(d, n_ref, n_alt, 2504, rows)
    open(os.path.join(d, "p.hap"), "w").write("placeholder\n")
    st = os.stat(os.path.join(d, "p.hap"))
    bench.write_panel_cache(os.path.join(d, "p.cache"), words, 2504, st)
    del words
    names = ",".join(f"ind{(7 + 5 * i) % 2504}" for i in range(n_ind))
    base = [exe, "-H", "p.hap", "-L", "p.legend", "-I", "p.indv", "-P", "p.pileup", "-s", names, "--LD", "--threads", "16",
            "--panel-cache", "p.cache", "-O", "o", "--summary-only"]
    os.makedirs(os.path.join(d, "o"))
    for slots in slot_list:
        best = None
        for rep in range(2):
            t0 = time.perf_counter()
            r = subprocess.run(base, cwd=d, env=dict(os.environ, IBDGEM_TIMING="1", **({"IBDGEM_OUT_SLOTS": str(slots)} if slots else {})), capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                print(r.stderr[-600:])
                sys.exit(1)
            ph = {}
            for l in r.stderr.splitlines():
                if l.startswith("## time "):
                    k, v = l[8:].rsplit(" ", 1)
                    ph[k] = ph.get(k, 0.0) + float(v)
            own = sum(v for k, v in ph.items() if k.startswith("per individual") or k.startswith("output files of the last"))
            if best is None or own < best[0]:
                best = (own, wall, ph)
        own, wall, ph = best
        print(f"{n_ind} individuals, --summary-only, IBDGEM_OUT_SLOTS={slots or 'default'}: {own / n_ind * 1e3:.2f} ms per individual (engine + output), wall {wall:.2f} s", flush=True)
        print("    " + " | ".join(f"{k[:50]} {v:.3f}" for k, v in ph.items() if "individual" in k or "output" in k), flush=True)
