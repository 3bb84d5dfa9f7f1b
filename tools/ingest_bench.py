"""Genotype-ingest rate of the host program beside the reference binary.

    python tools/ingest_bench.py [rows] [n_ids] [workdir]

Writes a synthetic IMPUTE panel (rows x n_ids, plain text) and a pileup, then times
`ibdgem --plan` (parse + pack + filter chain for one comparison individual, no device) for several
thread counts and with the packed-panel cache, and -- when oracle/_ref/ibdgem exists -- the
unmodified reference's non-LD run on the same files (it parses the same text, once per individual).
Prints one JSON line."""
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
n_ids = int(sys.argv[2]) if len(sys.argv) > 2 else 2504
work = sys.argv[3] if len(sys.argv) > 3 else tempfile.mkdtemp(prefix="ibdg_ingest_")
os.makedirs(work, exist_ok=True)
rng = np.random.default_rng(1)
hap = os.path.join(work, "p.hap")
with open(hap, "wb") as fh:
    for r0 in range(0, rows, 20000):
        n = min(20000, rows - r0)
        f = rng.beta(0.3, 1.0, size=(n, 1))
        bits = (rng.random((n, 2 * n_ids)) < f).astype(np.uint8)
        txt = np.full((n, 4 * n_ids), ord(" "), dtype=np.uint8)
        txt[:, 0::2] = bits + ord("0")
        txt[:, -1] = ord("\n")
        fh.write(txt.tobytes())
with open(os.path.join(work, "p.legend"), "w") as fh:
    fh.write("id position a0 a1\n")
    for i in range(rows):
        fh.write(f"rs{i} {1000 + 30 * i} A G\n")
with open(os.path.join(work, "p.indv"), "w") as fh:
    for n in range(n_ids):
        fh.write(f"ind{n}\n")
with open(os.path.join(work, "p.pileup"), "w") as fh:
    for i in range(rows):
        fh.write(f"1\t{1000 + 30 * i}\tN\t2\tAG\tII\t]]\n")
size_gb = os.path.getsize(hap) / 1e9
base = ["-H", "p.hap", "-L", "p.legend", "-I", "p.indv", "-P", "p.pileup", "-s", "ind7"]
exe = os.path.join(REPO, "ibdgem_amd", "host", "ibdgem")


def timed(cmd):
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        subprocess.run(cmd, cwd=work, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


out = {"rows": rows, "n_ids": n_ids, "hap_text_gb": round(size_gb, 3), "cpus": len(os.sched_getaffinity(0)), "host": {}}
for th in (1, 2, 4, 8, 16):
    if th > 2 * out["cpus"]:
        break
    dt = timed([exe, *base, "--plan", "--threads", str(th)])
    out["host"][f"threads_{th}"] = {"s": round(dt, 3), "rows_per_s": round(rows / dt), "text_gb_per_s": round(size_gb / dt, 2)}
cache = os.path.join(work, "p.cache")
timed([exe, *base, "--plan", "--panel-cache", cache])
dt = timed([exe, *base, "--plan", "--panel-cache", cache])
out["host"]["cached"] = {"s": round(dt, 3), "rows_per_s": round(rows / dt)}
ref = os.path.join(REPO, "oracle", "_ref", "ibdgem")
if os.path.exists(ref):
    o = os.path.join(work, "ref_out")
    os.makedirs(o, exist_ok=True)
    dt = timed([ref, *base, "-O", o])
    out["reference_nonld_one_individual"] = {"s": round(dt, 3), "rows_per_s": round(rows / dt)}
# end to end on a device (text in -> output files): ingest + engine + formatter, --LD, one individual
if os.path.exists("/dev/kfd"):
    out["end_to_end_LD"] = {}
    for th in (1, 16):
        o = os.path.join(work, f"out{th}")
        os.makedirs(o, exist_ok=True)
        dt = timed([exe, *base, "--LD", "-O", o, "--threads", str(th)])
        out["end_to_end_LD"][f"threads_{th}"] = {"s": round(dt, 3), "rows_per_s": round(rows / dt)}
    o = os.path.join(work, "out_warm")
    os.makedirs(o, exist_ok=True)
    dt = timed([exe, *base, "--LD", "-O", o, "--threads", "16", "--panel-cache", cache])   # cache written above
    out["end_to_end_LD"]["threads_16_panel_cache"] = {"s": round(dt, 3), "rows_per_s": round(rows / dt)}
    # many comparison individuals against one pileup (BASELINE.json configs[4] shape): 64 of them, summary files only
    o = os.path.join(work, "out_many")
    os.makedirs(o, exist_ok=True)
    many = ",".join(f"ind{7 + 5 * i}" for i in range(64))
    dt = timed([exe, "-H", "p.hap", "-L", "p.legend", "-I", "p.indv", "-P", "p.pileup", "-s", many, "--LD", "-O", o,
                "--threads", "16", "--summary-only"])
    out["end_to_end_LD"]["64_individuals_summary_only"] = {"s": round(dt, 3), "individual_rows_per_s": round(64 * rows / dt)}
    if os.path.exists(ref):
        o = os.path.join(work, "ref_out_ld")
        os.makedirs(o, exist_ok=True)
        dt = timed([ref, *base, "--LD", "-O", o])
        out["end_to_end_LD"]["reference"] = {"s": round(dt, 3), "rows_per_s": round(rows / dt)}
        same = all(open(os.path.join(o, f)).read().split("\n", 1)[1] ==
                   open(os.path.join(work, "out16", f)).read().split("\n", 1)[1]
                   for f in ("UNKWN.ind7.tab.txt", "UNKWN.ind7.summary.txt"))
        out["end_to_end_LD"]["files_identical_to_reference_after_line_1"] = same
print(json.dumps(out))
