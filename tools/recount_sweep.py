import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench, ibdgem_amd
dev = torch.device("cuda", 0)
rows = 4_000_000
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
eng = ibdgem_amd.Engine(0, 0.02, 20)
eng.upload_panel_dev(panel.data_ptr(), rows, 2504); del panel
eng.upload_sites(None, n_ref, n_alt, 100)
n_cov = int(((n_ref.astype(np.int32) + n_alt) > 0).sum())
eng.set_option("async", 1)
for _ in range(200): eng.run([7], ld=True)
eng.sync()
def rate(n=100):
    t0 = time.perf_counter()
    for _ in range(n): eng.run([7], ld=True)
    eng.sync()
    return (time.perf_counter() - t0) / n * 1e3
print("no recount", round(rate(), 4), "ms/step")
eng.set_option("count_in_run", 1)
for b in (3, 4, 5, 6, 4):
    eng.set_option("recount_blocks_per_cu", b)
    rate(20)
    ms = rate()
    eng.set_option("async", 0); a = np.mean([eng.run_ms(i)["alt_count"] for i in range(16)]); eng.set_option("async", 1)
    print(f"recount blocks/CU {b}: {ms:.4f} ms/step -> {n_cov / ms / 1e6:.3f}e9 sites/s, alt_count {a:.3f} ms")
