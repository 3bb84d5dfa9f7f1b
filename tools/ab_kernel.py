"""A/B of two builds of the engine on one box (boxes differ by a few per cent, so never compare across calls):

    python tools/ab_kernel.py build/a/libibdgem_hip.so build/b/libibdgem_hip.so [more.so ...] [rows]

All libraries get the bench workload (same panel, same sites); the dominant --LD kernel is timed through
its own dispatch events, in alternating rounds A B A B ..., and the results of the two are compared bit
for bit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import ibdgem_amd

libs = [a for a in sys.argv[1:] if a.endswith(".so")]
rows = ([int(a) for a in sys.argv[1:] if a.isdigit()] or [4_000_000])[0]
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
torch.cuda.synchronize()
engs = []
for lib in libs:
    e = ibdgem_amd.Engine(0, 0.02, 20, lib_path=os.path.abspath(lib))
    e.upload_panel_dev(panel.data_ptr(), rows, 2504)
    e.upload_sites(np.arange(rows, dtype=np.uint32), n_ref, n_alt, 100)
    engs.append(e)
del panel
torch.cuda.empty_cache()
wins = []
for e in engs:
    e.run([7], ld=True)
    wins.append(e.window_ll(0))
print("results identical:", all(bool((wins[0].view(np.uint64) == w.view(np.uint64)).all()) for w in wins[1:]))
times = [[] for _ in libs]
steps = [[] for _ in libs]
for rnd in range(6):
    for i, e in enumerate(engs):
        e.set_option("dispatch_events", 1)
        e.set_option("async", 1)
        for _ in range(60):
            e.run([7], ld=True)
        e.sync()
        times[i].append(float(np.mean([e.run_kernel_ms(b) for b in range(32)])))
        e.set_option("dispatch_events", 0)
        import time
        t0 = time.perf_counter()
        for _ in range(100):
            e.run([7], ld=True)
        e.sync()
        steps[i].append((time.perf_counter() - t0) * 10)
        e.set_option("async", 0)
for i, lib in enumerate(libs):
    print(f"{lib}: dominant kernel {np.median(times[i]):.4f} ms (rounds {', '.join(f'{t:.4f}' for t in times[i])}); "
          f"step {np.median(steps[i]):.4f} ms")
for i in range(1, len(libs)):
    print(f"{libs[i]} / {libs[0]}  kernel time: {np.median(times[i]) / np.median(times[0]):.4f}   step: {np.median(steps[i]) / np.median(steps[0]):.4f}")
