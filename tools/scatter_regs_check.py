"""Runs the covered-row rank check of tests/test_gpu_parity.py (684 workgroups of the stage-A scatter kernel, windows
of two covered rows) against an alternative build of the library: python tools/scatter_regs_check.py <lib.so>.
Used once in round 4 on a build whose scatter kernel keeps each row's record in registers between its ballot and its
store."""
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
