"""Per-step device times of queued (async) runs on the bench workload: python tools/step_times.py [rows] [steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, ibdgem_amd
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
eng = ibdgem_amd.Engine(0, 0.02, 20)
eng.upload_panel_dev(panel.data_ptr(), panel.shape[0], 2504)
del panel
eng.upload_sites(np.arange(rows, dtype=np.uint32), n_ref, n_alt, 100)
for _ in range(3):
    eng.run([7], ld=True)
eng.set_option("async", 1)
for rep in range(3):
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.run([7], ld=True)
    t1 = time.perf_counter()
    eng.sync()
    t2 = time.perf_counter()
    ms = [eng.run_ms(b)["ld"] for b in range(min(steps, 32))][::-1]
    print(f"rep {rep}: queue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms, per step {1e3*(t2-t0)/steps:.4f}; "
          f"sum of ld intervals {sum(ms):.3f}")
    print("  ", " ".join(f"{m:.3f}" for m in ms))
    try:
        print("   kernel only:", " ".join(f"{eng.run_kernel_ms(b):.3f}" for b in range(min(steps, 32)))[:400])
    except ibdgem_amd.EngineError as e:
        print("   (no kernel-only times:", e, ")")
