"""profiles/<round>_ld_traffic.json from the two counter passes tools/pmc_traffic.sh leaves in
gpurun_out/<tag>_traffic.json (the file bench.py reads for `roofline.traffic`):

    python tools/traffic_summary.py gpurun_out/r03_traffic.json profiles/r03_ld_traffic.json

HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB as rocprofv3 reports them, averaged over dispatches):
on gfx950 FETCH_SIZE reports half the bytes of 16-byte-per-lane streaming reads (MI355X_MICROARCH.md, HBM
section) -- k_alt_count, which reads the 2.56 GB panel once, is the check of that factor in the same file."""
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
raw = json.load(open(src))
kernels = {}
for name, kb in raw["FETCH_SIZE"].items():
    w = raw["WRITE_SIZE"].get(name, 0.0)
    kernels[name.replace("void ", "").strip()] = {"FETCH_SIZE_KB": kb, "WRITE_SIZE_KB": w, "hbm_bytes_corrected": (2 * kb + w) * 1024}
# the kernel of the timed steps: the IBD1 form (last template argument true) where the run reached it -- the form that counts
# everything is then the site list's first runs and the one pass that writes every individual's IBD0 product
cands = [k for k in kernels if "k_ld_popcount<" in k]
dom = next((k for k in cands if k.rstrip().endswith("true, true, true>")), None) or max(cands, key=lambda k: kernels[k]["hbm_bytes_corrected"])
out = {
    "source": "tools/pmc_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two passes) -- python bench.py "
              "--timed-only --steps 3 --warmup 1; summarised by tools/traffic_summary.py",
    "config": {"sites": 4000000, "n_ids": 2504, "window": 100, "targets": 1},
    "unit": "KB as reported by rocprofv3, averaged over dispatches",
    "gfx950_correction": "FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads (MI355X_MICROARCH.md, HBM): fetch bytes "
                         "are doubled, write bytes taken as reported; k_alt_count (2.56 GB panel read once) confirms the factor",
    "kernels": kernels,
    "dominant_kernel": dom,
    "dominant_kernel_hbm_bytes_per_launch": kernels[dom]["hbm_bytes_corrected"],
}
json.dump(out, open(dst, "w"), indent=1)
print(dom, out["dominant_kernel_hbm_bytes_per_launch"])
