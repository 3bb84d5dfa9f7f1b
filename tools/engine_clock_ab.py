"""One comparison's engine clock (bench.py engine_clock.alt_counts_amortised_ms: ibdg_upload_sites_dev + ibdg_run +
ibdg_get_window_ll into page-locked memory, option async) for two builds of the library, alternating on one box:
    python tools/engine_clock_ab.py ibdgem_amd/libibdgem_hip_prev.so ibdgem_amd/libibdgem_hip.so"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, ibdgem_amd
rows = 4_000_000
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
d_nr, d_na = torch.from_numpy(n_ref).cuda(), torch.from_numpy(n_alt).cuda()
engs = []
for path in sys.argv[1:]:
    e = ibdgem_amd.Engine(0, 0.02, 20, lib_path=os.path.abspath(path))
    e.upload_panel_dev(panel.data_ptr(), rows, 2504)
    engs.append((path, e))
del panel
torch.cuda.empty_cache()
torch.cuda.synchronize()


def best_of(n, fn):
    b = None
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        dt = (time.perf_counter() - t0) * 1e3
        b = dt if b is None else min(b, dt)
    return b


pins = {}
for rnd in range(3):
    for path, e in engs:
        def up():
            e.upload_sites_dev(None, d_nr.data_ptr(), d_na.data_ptr(), rows, 100)
            e.sync()
        up()
        t_up = best_of(20, up)
        if path not in pins:
            pins[path] = ibdgem_amd.PinnedArray((e.n_windows, 3), np.float64)

        def once():
            e.upload_sites_dev(None, d_nr.data_ptr(), d_na.data_ptr(), rows, 100)
            e.run([7], ld=True)
            e.window_ll(0, out=pins[path].array)
        e.set_option("async", 1)
        once()
        t_cmp = best_of(10, once)
        t_ready = None
        try:
            e.set_option("dev_inputs_ready", 1)
            once()
            t_ready = best_of(10, once)
            e.set_option("dev_inputs_ready", 0)
        except ibdgem_amd.EngineError:
            pass
        e.set_option("async", 0)
        print(f"{os.path.basename(path)}: upload_sites_dev + sync {t_up:.4f} ms | comparison {t_cmp:.4f} ms | with dev_inputs_ready "
              f"{'n/a' if t_ready is None else f'{t_ready:.4f} ms'} | engine's upload clocks {e.upload_ms()}", flush=True)
a = pins[sys.argv[1]].array
print("window tables bit-equal across the builds:", all(np.array_equal(a.view(np.uint64), p.array.view(np.uint64)) for p in pins.values()))
