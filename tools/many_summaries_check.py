"""The host program, --summary-only, many comparison individuals named with -s, run again and again: every run must write
every summary file with the same bytes (a race between strtok in the -s parser and the runtime's start-up thread once
cut the list short at random -- fixed with strtok_r).  python tools/many_summaries_check.py [individuals] [exe ...]"""
import os, sys, tempfile, subprocess, time, hashlib, glob
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
rows = 1_000_000
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
words = panel.cpu().numpy().view(np.uint64)
del panel

n_ind = int(sys.argv[1]) if len(sys.argv) > 1 else 90
with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
    bench.write_pileup_and_legend(d, n_ref, n_alt, 2504, rows)
    open(os.path.join(d, "p.hap"), "w").write("placeholder\n")
    st = os.stat(os.path.join(d, "p.hap"))
    bench.write_panel_cache(os.path.join(d, "p.cache"), words, 2504, st)
    names = ",".join(f"ind{(7 + 5 * i) % 2504}" for i in range(n_ind))
    ref = None
    for thr, dbg in [("16", e) for e in (sys.argv[2:] or ["ibdgem_amd/host/ibdgem"]) for _ in range(int(os.environ.get("REPS", "25")))]:
        out = os.path.join(d, "o" + str(time.time()))
        os.makedirs(out)
        exe = os.path.join(bench.REPO, dbg)
        base = [exe, "-H", "p.hap", "-L", "p.legend", "-I", "p.indv", "-P", "p.pileup", "-s", names, "--LD", "--threads", "16",
                "--panel-cache", "p.cache", "-O", out, "--summary-only"]
        t0 = time.perf_counter()
        r = subprocess.run(base, cwd=d, env=dict(os.environ, IBDGEM_TIMING="1", IBDGEM_OUT_SLOTS="6", IBDGEM_OUT_THREADS=thr, IBDGEM_DEBUG_LOOP="1", LD_LIBRARY_PATH=os.path.join(bench.REPO, "ibdgem_amd") + ":/opt/rocm/lib"), capture_output=True, text=True)
        wall = time.perf_counter() - t0
        files = sorted(glob.glob(out + "/*.summary.txt"))
        h = hashlib.sha256()
        total = 0
        for f in files:
            b = open(f, "rb").read()
            total += len(b)
            h.update(b)
        print(f"threads {thr} dbg {dbg}: rc {r.returncode}, wall {wall:.2f} s, {len(files)} summary files, {total} bytes, {h.hexdigest()[:16]}")
        run_lines = [l for l in r.stderr.splitlines() if l.startswith("Running ")]
        have = {os.path.basename(f).split(".")[1] for f in files}
        missing = [f"ind{(7 + 5 * i) % 2504}" for i in range(n_ind) if f"ind{(7 + 5 * i) % 2504}" not in have]
        print("   ", len(run_lines), "comparisons announced; missing files:", missing[:12])
        print("   ", [l for l in r.stderr.splitlines() if not l.startswith("## time") and not l.startswith("Running ")][-30:] if len(files) != n_ind else "ok")
