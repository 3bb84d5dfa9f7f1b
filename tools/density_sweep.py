"""Engine clock of one comparison against the density of the pileup, from the panel's own tiles and from the compacted
tiles of the site list: the measurement behind the option "compact_density" (docs/DESIGN_rounds_1-4.md s4.1b).

    python tools/density_sweep.py [panel_rows]

A 2504-individual panel of `panel_rows` rows; for each share s of rows that have a pileup line (every such row with reads)
the engine clock -- ibdg_upload_sites_dev + ibdg_run (--LD, one comparison individual) + the window table to host memory --
with compact_tiles -1 and ld_variant 2 (the panel's own tiles, whatever the density), compact_tiles 1 (compacted), and
compact_tiles -1 with ld_variant 0 (no compacted tiles: the strict kernel wherever the site list counts as thin -- what
round 3 did below one row in nine)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench, ibdgem_amd
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
dev = torch.device("cuda", 0)
panel, n_ref, n_alt = bench.build_shard(torch, dev, 0, rows, 2504, 7, 20241008)
eng = ibdgem_amd.Engine(0, 0.02, 20)
eng.upload_panel_dev(panel.data_ptr(), panel.shape[0], 2504)
del panel
torch.cuda.empty_cache()
rng = np.random.default_rng(5)
print(f"{'share':>6} {'sites':>8} | in-place ms  compacted ms  strict fallback ms | faster")
for share in (1.0, 0.6, 0.5, 0.4, 0.33, 0.25, 0.2, 0.15, 0.11, 0.08, 0.05, 0.02):
    if share == 1.0:
        keep = np.arange(rows, dtype=np.uint32)
        nr, na = n_ref.copy(), n_alt.copy()
        nr[(nr.astype(np.int32) + na) == 0] = 1
    else:
        keep = np.sort(rng.choice(rows, size=int(rows * share), replace=False)).astype(np.uint32)
        nr, na = n_ref[keep].copy(), n_alt[keep].copy()
        nr[(nr.astype(np.int32) + na) == 0] = 1
    k = len(keep)
    d_idx = torch.from_numpy(keep.view(np.int32)).cuda()
    d_nr, d_na = torch.from_numpy(nr).cuda(), torch.from_numpy(na).cuda()
    torch.cuda.synchronize()
    res = {}
    for name, tiles, variant in (("in-place", -1, 2), ("compacted", 1, 0), ("strict", -1, 0)):
        eng.set_option("compact_tiles", tiles)
        eng.set_option("ld_variant", variant)
        eng.set_option("async", 1)
        def once():
            eng.upload_sites_dev(d_idx.data_ptr(), d_nr.data_ptr(), d_na.data_ptr(), k, 100)
            eng.run([7], ld=True)
            return eng.window_ll(0)
        for _ in range(3):
            once()
        ms = []
        for _ in range(8):
            t0 = time.perf_counter()
            once()
            ms.append((time.perf_counter() - t0) * 1e3)
        eng.set_option("async", 0)
        res[name] = min(ms)
    best = min(("in-place", "compacted"), key=lambda n: res[n])
    print(f"{share:6.2f} {k:8d} | {res['in-place']:11.3f} {res['compacted']:13.3f} {res['strict']:18.3f} | {best}", flush=True)
eng.close()
