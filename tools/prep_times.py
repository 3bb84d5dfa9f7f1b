"""The device side of ibdg_upload_sites on the bench workload: 20 calls of ibdg_upload_sites_dev with the arrays
resident (run under `rocprofv3 --kernel-trace --stats` for the per-kernel times of k_prep_*):
    python tools/prep_times.py [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, ibdgem_amd
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
eng = ibdgem_amd.Engine(0, 0.02, 20, lib_path=os.environ.get("IBDG_LIB") or None)
for kv in os.environ.get("IBDG_OPTS", "").split(","):          # e.g. IBDG_OPTS=compact_tiles=1: the re-layout inside every upload
    if kv:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
eng.upload_panel_dev(panel.data_ptr(), rows, 2504)
del panel
torch.cuda.empty_cache()
d_nr, d_na = torch.from_numpy(n_ref).cuda(), torch.from_numpy(n_alt).cuda()
torch.cuda.synchronize()
best = None
for i in range(20):
    t0 = time.perf_counter()
    eng.upload_sites_dev(None, d_nr.data_ptr(), d_na.data_ptr(), rows, 100)
    eng.sync()
    dt = (time.perf_counter() - t0) * 1e3
    best = dt if best is None else min(best, dt)
print(f"upload_sites_dev + sync, {rows} rows: best of 20 {best:.4f} ms; engine's clocks of the last call {eng.upload_ms()}")
