"""k_alt_count alone on a 4M x 2504 packed panel of random bits (run on the GPU box):
    python tools/time_alt_count.py [rows] [n_ids]
Prints the kernel's event time inside ibdg_run (count_in_run) and the implied read rate."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import ibdgem_amd

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
n_ids = int(sys.argv[2]) if len(sys.argv) > 2 else 2504
words = 2 * ((n_ids + 63) // 64)
panel = torch.randint(-2**62, 2**62, (rows, words), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
eng = ibdgem_amd.Engine(0, 0.02, 20)
eng.upload_panel_dev(panel.data_ptr(), rows, n_ids)
n = min(rows, 100_000)
eng.upload_sites(None, np.ones(n, np.uint8), np.zeros(n, np.uint8), 100)
eng.set_option("count_in_run", 1)
ms = []
for _ in range(30):
    eng.run([0], ld=False)
    ms.append(eng.last_run_ms()["alt_count"])
best, med = min(ms), float(np.median(ms))
nbytes = rows * words * 8
print(f"k_alt_count {rows} x {n_ids}: best {best:.4f} ms ({nbytes / best / 1e6:.0f} GB/s), median {med:.4f} ms ({nbytes / med / 1e6:.0f} GB/s)")
