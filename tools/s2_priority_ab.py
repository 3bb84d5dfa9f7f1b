"""The second stream's priority (highest / middle / lowest) against the step clock and one comparison's latency: an
experimental build reads IBDG_EXP_S2_PRIO at ibdg_create.  python tools/s2_priority_ab.py <lib>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, ibdgem_amd
rows = 4_000_000
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
d_nr, d_na = torch.from_numpy(n_ref).cuda(), torch.from_numpy(n_alt).cuda()
engs = []
for prio in ("hi", "mid", "lo"):
    os.environ["IBDG_EXP_S2_PRIO"] = prio
    e = ibdgem_amd.Engine(0, 0.02, 20, lib_path=os.path.abspath(sys.argv[1]))
    e.upload_panel_dev(panel.data_ptr(), rows, 2504)
    e.upload_sites(None, n_ref, n_alt, 100)
    engs.append((prio, e))
del panel
torch.cuda.empty_cache()
torch.cuda.synchronize()
pin = None
for rnd in range(3):
    for prio, e in engs:
        e.set_option("async", 1)
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) < 0.08:
            for _ in range(8):
                e.run([7], ld=True)
            e.sync()
        t0 = time.perf_counter()
        for _ in range(300):
            e.run([7], ld=True)
        e.sync()
        step = (time.perf_counter() - t0) / 300 * 1e3
        if pin is None:
            pin = ibdgem_amd.PinnedArray((e.n_windows, 3), np.float64)

        def once():
            e.upload_sites_dev(None, d_nr.data_ptr(), d_na.data_ptr(), rows, 100)
            e.run([7], ld=True)
            e.window_ll(0, out=pin.array)
        for _ in range(10):
            once()
        lat = []
        for _ in range(20):
            t0 = time.perf_counter()
            once()
            lat.append((time.perf_counter() - t0) * 1e3)
        e.set_option("async", 0)
        print(f"stream2 priority {prio}: step {step:.4f} ms | one comparison min {min(lat):.4f} median {np.median(lat):.4f} ms", flush=True)
