"""What the engine's own re-layout of a site list costs (host wall clock of the run it happens in), the first time in a
context (buffers allocated) and on later uploads; and a steady stream of single runs before and after it.
    python tools/relayout_cost.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, ibdgem_amd
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
eng = ibdgem_amd.Engine(0, 0.02, 20, lib_path=os.environ.get("IBDG_LIB") or None)
for kv in os.environ.get("IBDG_OPTS", "").split(","):
    if kv:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
t0 = time.perf_counter()
eng.upload_panel_dev(panel.data_ptr(), panel.shape[0], 2504)
eng.sync()
print(f"panel upload (device-resident rows): {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
del panel
torch.cuda.empty_cache()
idx = np.arange(rows, dtype=np.uint32)
for rnd in range(3):
    eng.upload_sites(idx, n_ref, n_alt, 100)
    eng.sync()
    walls = []
    for k in range(20):
        t0 = time.perf_counter()
        eng.run([7], ld=True)
        eng.sync()
        walls.append((time.perf_counter() - t0) * 1e3)
        if eng.ld_layout() == 2 and len(walls) and "sw" not in locals():
            sw = k
    print(f"upload {rnd}: switched in run {sw + 1}: {walls[sw]:.3f} ms; runs before {np.median(walls[:sw]):.3f} ms, after {np.median(walls[sw + 1:]):.3f} ms (synchronous, host wall)", flush=True)
    del sw
    eng.set_option("async", 1)
    for _ in range(300):
        eng.run([7], ld=True)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(300):
        eng.run([7], ld=True)
    eng.sync()
    print(f"   queued steps on layout {eng.ld_layout()}: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms per step", flush=True)
    eng.set_option("async", 0)
