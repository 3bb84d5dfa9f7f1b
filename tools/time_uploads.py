"""Host-side cost of the two uploads on the bench workload (run on the GPU box): python tools/time_uploads.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, ibdgem_amd
dev = torch.device("cuda", 0)
rows = 4_000_000
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
eng = ibdgem_amd.Engine(0, 0.02, 20)
t0 = time.perf_counter(); eng.upload_panel_dev(panel.data_ptr(), panel.shape[0], 2504); t1 = time.perf_counter()
print("upload_panel_dev (transpose + alt counts)", round(t1 - t0, 4), "s")
host = panel.cpu().numpy().view(np.uint64)
for staged in (0, 1, 0, 1):
    eng.set_option("staged_upload", staged)
    t0 = time.perf_counter(); eng.upload_panel(host, 2504); t1 = time.perf_counter()
    print(f"upload_panel from pageable host memory ({host.nbytes / 1e9:.2f} GB), staged_upload={staged}", round(t1 - t0, 4), "s", f"{host.nbytes / (t1 - t0) / 1e9:.1f} GB/s")
import mmap, tempfile
with tempfile.NamedTemporaryFile(dir="/dev/shm") as fh:       # the same bytes as a mapped file (4 KiB pages from the page cache)
    fh.write(host.tobytes()); fh.flush()
    mm = mmap.mmap(fh.fileno(), 0, prot=mmap.PROT_READ)
    arr = np.frombuffer(mm, dtype=np.uint64).reshape(host.shape)
    for staged in (0, 1, 0, 1):
        eng.set_option("staged_upload", staged)
        t0 = time.perf_counter(); eng.upload_panel(arr, 2504); t1 = time.perf_counter()
        print(f"upload_panel from a mapped file, staged_upload={staged}", round(t1 - t0, 4), "s", f"{host.nbytes / (t1 - t0) / 1e9:.1f} GB/s")
    del arr; mm.close()
del host
idx = np.arange(rows, dtype=np.uint32)
for _ in range(2):
    t0 = time.perf_counter(); eng.upload_sites(idx, n_ref, n_alt, 100); t1 = time.perf_counter()
    print("upload_sites 4M rows", round(t1 - t0, 4), "s", eng.upload_ms())
d_nr, d_na = torch.from_numpy(n_ref).cuda(), torch.from_numpy(n_alt).cuda()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); eng.upload_sites_dev(None, d_nr.data_ptr(), d_na.data_ptr(), rows, 100); t1 = time.perf_counter()
    print("upload_sites_dev 4M rows (device arrays, implicit rows)", round(t1 - t0, 5), "s", eng.upload_ms())
