"""hipMalloc wall time inside a process that runs the engine (why the first re-layout of a context took 270 ms)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, ibdgem_amd
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
def probe(tag, gb=2.83, free=True):
    p = C.c_void_p()
    t0 = time.perf_counter()
    rc = hip.hipMalloc(C.byref(p), int(gb * 1e9))
    t1 = time.perf_counter()
    if free:
        hip.hipFree(p)
    print(f"{tag}: hipMalloc {gb} GB {1e3 * (t1 - t0):.2f} ms rc {rc}", flush=True)
    return p
rows = 4_000_000
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
probe("after torch init")
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
torch.cuda.synchronize()
probe("after the synthetic panel (torch holds it)")
eng = ibdgem_amd.Engine(0, 0.02, 20)
eng.set_option("compact_tiles", -1)
eng.upload_panel_dev(panel.data_ptr(), panel.shape[0], 2504)
eng.sync()
probe("after the panel upload")
del panel
torch.cuda.empty_cache()
probe("after torch gave its panel back")
probe("again")
eng.upload_sites(np.arange(rows, dtype=np.uint32), n_ref, n_alt, 100)
eng.sync()
probe("after the site upload")
for k in range(15):
    eng.run([7], ld=True)
eng.sync()
probe("after 15 runs")
probe("again")
probe("0.5 GB", 0.5)
probe("1.5 GB", 1.5)
probe("2.1 GB", 2.1)
probe("2.2 GB", 2.2)
probe("4 GB", 4.0)
