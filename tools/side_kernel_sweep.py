"""Step time of the bench workload for different sizes of the per-site kernel's grid beside the --LD kernel:
    python tools/side_kernel_sweep.py   (on a GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench, ibdgem_amd
dev = torch.device("cuda", 0)
rows = 4_000_000
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
eng = ibdgem_amd.Engine(0, 0.02, 20)
eng.upload_panel_dev(panel.data_ptr(), rows, 2504); del panel
eng.upload_sites(None, n_ref, n_alt, 100)
eng.set_option("async", 1)
for _ in range(200): eng.run([7], ld=True)
eng.sync()
def rate(n=200):
    t0 = time.perf_counter()
    for _ in range(n): eng.run([7], ld=True)
    eng.sync()
    return (time.perf_counter() - t0) / n * 1e3
for b in (0, 1, 2, 4, 8, 16, 0, 4):
    eng.set_option("site_blocks_per_cu", b)
    rate(30)
    ms = rate()
    eng.set_option("async", 0); k = {n: float(np.mean([eng.run_ms(i)[n] for i in range(16)])) for n in ("rows", "ld")}; eng.set_option("async", 1)
    print(f"site blocks/CU {b}: {ms:.4f} ms/step  k_rows_windows {k['rows']:.3f}  ld launches {k['ld']:.3f}")
