"""Device time of one ibdg_run over T comparison individuals on the bench workload
(BASELINE.json configs[4] shape on one GPU): python tools/multi_target.py [rows] [T ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, ibdgem_amd
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
Ts = [int(a) for a in sys.argv[2:]] or [1, 4, 16, 64]
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
eng = ibdgem_amd.Engine(0, 0.02, 20, lib_path=os.environ.get("IBDG_LIB") or None)
for kv in os.environ.get("IBDG_OPTS", "").split(","):
    if kv:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
eng.upload_panel_dev(panel.data_ptr(), panel.shape[0], 2504)
del panel
torch.cuda.empty_cache()
eng.upload_sites(np.arange(rows, dtype=np.uint32), n_ref, n_alt, 100)
n_cov = int(((n_ref.astype(np.int32) + n_alt) > 0).sum())
for T in Ts:
    targets = [(7 + 5 * i) % 2504 for i in range(T)]
    eng.run(targets, ld=True)
    best = None
    for _ in range(3):
        eng.run(targets, ld=True)
        ms = eng.last_run_ms()
        best = ms if best is None or ms["total"] < best["total"] else best
    print(f"T={T}: total {best['total']:.3f} ms, ld {best['ld']:.3f} ms, per target {best['ld'] / T:.4f} ms, "
          f"{n_cov * T / best['total'] / 1e6:.1f}e9 site-target pairs/s", flush=True)
