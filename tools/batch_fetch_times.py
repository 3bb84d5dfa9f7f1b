"""Where the time of a batch of 30 comparison individuals goes on the host side: ibdg_run, then one ibdg_get_window_ll per
individual (pageable / page-locked destination): python tools/batch_fetch_times.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, ibdgem_amd
rows = 4_000_000
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
eng = ibdgem_amd.Engine(0, 0.02, 20)
eng.set_option("site_results", 0)
eng.set_option("compact_tiles", 1)
eng.upload_panel_dev(panel.data_ptr(), panel.shape[0], 2504)
del panel
torch.cuda.empty_cache()
eng.upload_sites(np.arange(rows, dtype=np.uint32), n_ref, n_alt, 100)
n_win = eng.n_windows
pin = ibdgem_amd.PinnedArray((n_win, 3), np.float64)
page = np.empty((n_win, 3), np.float64)
for rnd in range(4):
    tg = [(7 + 5 * (30 * rnd + i)) % 2504 for i in range(30)]
    t0 = time.perf_counter()
    eng.run(tg, ld=True)
    t1 = time.perf_counter()
    for i in range(30):
        eng.window_ll(i, out=pin.array)
    t2 = time.perf_counter()
    for i in range(30):
        eng.window_ll(i, out=page)
    t3 = time.perf_counter()
    print(f"batch {rnd}: ibdg_run {1e3 * (t1 - t0):.2f} ms (device {eng.last_run_ms()['total']:.2f}), 30 fetches page-locked {1e3 * (t2 - t1):.2f} ms, pageable {1e3 * (t3 - t2):.2f} ms", flush=True)
