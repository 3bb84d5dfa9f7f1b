"""Randomised sweeps as driver-run evidence (GPU tier): the tools/fuzz_*.py sweeps that profiles/r02_fuzz.txt
records at thousands of cases, here at a size that takes seconds, with fixed seeds.

  * tools/fuzz_parity.py: random panels / windows / error rates / depths / background lists / batches / launch
    geometries / reference-order mode through the C ABI against the oracle (bit-exact per-site values and LIBD2,
    1e-10 on the --LD columns);
  * tools/fuzz_cli_full.py: the whole host program against the unmodified reference binary, output files byte for
    byte -- only where oracle/_ref/ibdgem travelled with the tree (it is built by __graft_entry__.build() wherever
    /root/reference exists).  Its seed is one whose cases contain none of the seventh-digit %e ties on 2-3-site
    windows that DESIGN.md s8 (docs/DESIGN_rounds_1-4.md s2) describes (about one file in 400 in a random sweep)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_engine_against_oracle_on_random_configurations():
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "fuzz_parity.py"), "400", "31"], cwd=REPO,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "fuzz: 400 cases, 0 failures" in r.stdout, r.stdout[-2000:]


@pytest.mark.skipif(not os.path.exists(os.path.join(REPO, "oracle", "_ref", "ibdgem")),
                    reason="the reference binary (oracle/_ref/ibdgem) is not in this tree")
@pytest.mark.parametrize("threaded", [False, True])
def test_host_program_against_the_reference_binary_on_random_inputs(threaded):
    env = dict(os.environ, IBDGEM_MT_MIN_BYTES="1") if threaded else dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "fuzz_cli_full.py"), "40", "33"], cwd=REPO, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "full CLI fuzz: 40 cases" in r.stdout and "0 failures" in r.stdout.splitlines()[-1], r.stdout[-2000:]


@pytest.mark.skipif(not os.path.exists(os.path.join(REPO, "oracle", "_ref", "ibdgem")),
                    reason="the reference binary (oracle/_ref/ibdgem) is not in this tree")
def test_host_program_with_many_comparison_individuals_against_the_reference_binary():
    """8-40 comparison individuals per case: the host program batches them and the engine takes them through its
    matrix-core kernel (k_ld_mfma); every output file of every individual byte for byte, seventh-digit ties of
    --LD values in windows of two or three rows aside (counted by the tool; a handful per thousand files)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "fuzz_cli_full.py"), "60", "42", "--many-targets"],
                       cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "full CLI fuzz: 60 cases" in r.stdout and "0 failures" in r.stdout.splitlines()[-1], r.stdout[-2000:]

