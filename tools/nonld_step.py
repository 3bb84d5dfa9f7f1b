import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch, bench, ibdgem_amd
dev = torch.device("cuda", 0)
rows = 4_000_000
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
for lib in sys.argv[1:]:
    eng = ibdgem_amd.Engine(0, 0.02, 20, lib_path=os.path.abspath(lib))
    eng.upload_panel_dev(panel.data_ptr(), rows, 2504)
    eng.upload_sites(None, n_ref, n_alt, 100)
    eng.set_option("async", 1)
    for _ in range(100): eng.run([7], ld=False)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(200): eng.run([7], ld=False)
    eng.sync()
    ms = (time.perf_counter() - t0) / 200 * 1e3
    eng.set_option("async", 0)
    k = float(np.mean([eng.run_ms(i)["rows"] for i in range(16)]))
    print(f"{lib}: non-LD step {ms:.4f} ms  k_rows_windows {k:.4f}")
    eng.close()
