"""Runs the covered-row rank check of tests/test_gpu_parity.py (684 workgroups of the stage-A scatter kernel, windows
of two covered rows) against an alternative build of the library: python tools/scatter_regs_check.py <lib.so>.
Used in round 4 on a build of commit 9588f54's ibdg_prep.hip whose k_prep_site_scatter kept each row's record in
registers between its ballot and its store (block_scatter with a uint2 per row handed from the flag pass to the emit
pass: 24 VGPRs, the lane number in v23): RANKS WRONG in 1-13 of 684 workgroups per upload, never one of the first 256.
The cause is the hardware hazard of tools/ubench/shift64_top_vgpr.hip (docs/DESIGN_rounds_1-4.md s4.4); today's preparation kernels
have no 64-bit shift by a per-lane amount left, so the variant cannot be rebuilt from them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_parity as T
try:
    n = T.covered_ranks_check(os.path.abspath(sys.argv[1]))
    print(f"ranks right in all {n} workgroups")
except AssertionError as e:
    print("RANKS WRONG:", e)
    sys.exit(1)
