"""The non-LD step (k_rows_windows alone on the chip, bench.py non_ld) for several builds of the library, alternating:
    python tools/nonld_ab.py ibdgem_amd/libibdgem_hip.so ibdgem_amd/libibdgem_hip_x.so"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, ibdgem_amd
rows = 4_000_000
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
engs = []
for path in sys.argv[1:]:
    e = ibdgem_amd.Engine(0, 0.02, 20, lib_path=os.path.abspath(path))
    e.upload_panel_dev(panel.data_ptr(), rows, 2504)
    e.upload_sites(None, n_ref, n_alt, 100)
    engs.append((path, e))
del panel
torch.cuda.empty_cache()
sweep = [int(x) for x in os.environ.get("ROWS_BLOCKS", "").split(",") if x] or [None]
for rnd in range(3):
    for path, e in engs:
      for rb in sweep:
        if rb is not None:
            try:
                e.set_option("rows_blocks_per_cu", rb)
            except ibdgem_amd.EngineError:
                if rb != sweep[0]:
                    continue
        for res in (1, 0):
            e.set_option("site_results", res)
            e.set_option("async", 1)
            for _ in range(100):
                e.run([7], ld=False)
            e.sync()
            t0 = time.perf_counter()
            for _ in range(200):
                e.run([7], ld=False)
            e.sync()
            ms = (time.perf_counter() - t0) / 200 * 1e3
            e.set_option("async", 0)
            k = float(np.mean([e.run_ms(i)["rows"] for i in range(16)]))
            print(f"{os.path.basename(path)} rows_blocks_per_cu={rb} site_results={res}: {ms:.4f} ms per step, kernel {k:.4f} ms", flush=True)

# the builds' results against the first one's: every bit of the window table and of the per-row values
ref = None
for path, e in engs:
    e.set_option("site_results", 1)
    e.run([7], ld=False)
    got = (e.window_ll(0).copy(), e.site_ll(0).copy())
    if ref is None:
        ref = got
    else:
        same = all(np.array_equal(x.view(np.uint64), y.view(np.uint64)) for x, y in zip(got, ref))
        print(f"{os.path.basename(path)}: results bit-equal to {os.path.basename(engs[0][0])}: {same}")
