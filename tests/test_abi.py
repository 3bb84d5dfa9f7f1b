"""CPU-side checks of the C ABI: the library loads, exports every symbol the
header declares, and its host-only helpers (P(D|G) table, row packing) agree
with the oracle / numpy.  No compute entry point is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from ibdgem_amd import engine as E

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(REPO, "include", "ibdgem_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ibdg_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = E.load_library()
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libibdgem_hip.so does not export {n}"
    assert sorted(E.SYMBOLS) == names, "python binding table out of sync with include/ibdgem_hip.h"
    assert lib.ibdg_abi_version() == 5


def test_no_torch_types_in_header():
    text = open(os.path.join(REPO, "include", "ibdgem_hip.h")).read()
    assert "torch" not in text.lower().replace("a torch tensor's data_ptr", "") and "at::" not in text
    assert 'extern "C"' in text


@pytest.mark.parametrize("eps,M", [(0.02, 20), (0.05, 6), (1e-3, 40), (1e-30, 20), (0.5, 1)])
def test_pdg_table_matches_oracle_bitwise(oracle, eps, M):
    lib = E.load_library()
    d = M + 1
    tab = np.empty((d, d, 3), dtype=np.float64)
    assert lib.ibdg_pdg_table(eps, M, tab.ctypes.data) == 0
    for r in range(d):
        for a in range(d - r):
            want = oracle.pdg(eps, M, r, a)
            assert [x.hex() for x in tab[r, a]] == [float(x).hex() for x in want], (r, a)
        for a in range(d - r, d):
            assert np.isnan(tab[r, a]).all()      # unreachable: n_ref+n_alt > max_cov is filtered


def test_pdg_table_rejects_bad_arguments():
    lib = E.load_library()
    buf = np.empty(3 * 129 * 129)
    assert lib.ibdg_pdg_table(0.02, 0, buf.ctypes.data) != 0
    assert lib.ibdg_pdg_table(0.02, 128, buf.ctypes.data) != 0
    assert lib.ibdg_pdg_table(0.02, 20, None) != 0


@pytest.mark.parametrize("n_ids", [1, 3, 63, 64, 65, 130, 2504])
def test_row_packing(n_ids):
    lib = E.load_library()
    rng = np.random.default_rng(n_ids)
    alle = (rng.random((4, 2 * n_ids)) < 0.4).astype(np.uint8)
    rows = E.pack_alleles(alle)
    assert rows.shape[1] == lib.ibdg_row_words(n_ids) == 2 * ((n_ids + 63) // 64)
    assert (rows == E.pack_alleles_fast(alle)).all()
    for i in range(4):
        for n in (0, n_ids // 2, n_ids - 1):
            w, b = 2 * (n // 64), n % 64
            assert (int(rows[i, w]) >> b) & 1 == alle[i, 2 * n]
            assert (int(rows[i, w + 1]) >> b) & 1 == alle[i, 2 * n + 1]
        assert sum(bin(int(x)).count("1") for x in rows[i]) == alle[i].sum()
        # the IMPUTE text form of the same row (reference hap_buf, src/ibdgem.c:638-639)
        text = " ".join(map(str, alle[i])).encode()
        out = np.zeros(rows.shape[1], dtype=np.uint64)
        assert lib.ibdg_pack_hap_text(text, n_ids, out.ctypes.data) == 0
        assert (out == rows[i]).all()
    out = np.zeros(lib.ibdg_row_words(n_ids + 1), dtype=np.uint64)       # the row is cleared at the width asked for
    assert lib.ibdg_pack_hap_text(b"0 1", n_ids + 1, out.ctypes.data) == 1          # too short
    out = np.zeros(rows.shape[1], dtype=np.uint64)
    if n_ids >= 3:
        bad = bytearray(" ".join(map(str, alle[0])).encode())
        bad[4] = ord("2")
        assert lib.ibdg_pack_hap_text(bytes(bad), n_ids, out.ctypes.data) == 1      # not 0/1


def test_create_fails_loudly_without_a_device():
    """Asks the HIP runtime for devices; runs in a process of its own so that a `-m "not gpu"` session
    never initialises a GPU runtime."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from ibdgem_amd import engine as E\n"
        "lib = E.load_library()\n"
        "if lib.ibdg_device_count() > 0:\n"
        "    print('GPU'); sys.stdout.flush(); raise SystemExit(0)\n"
        "try:\n"
        "    E.Engine(); print('created')\n"
        "except E.EngineError as e:\n"
        "    print('ERR1', e)\n"
        "print('NULL' if lib.ibdg_create(0, 0.02, 0) is None else 'ctx')\n"
        "print('ERR2', lib.ibdg_last_error(None).decode())\n"
        "sys.stdout.flush()\n" % REPO)
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True).stdout
    if out.startswith("GPU"):
        pytest.skip("a GPU is present")
    lines = out.splitlines()
    assert lines[0].startswith("ERR1") and "no HIP device" in lines[0] and "no CPU path" in lines[0], out
    assert lines[1] == "NULL" and lines[2].startswith("ERR2") and "-M" in lines[2], out


def test_missing_library_is_an_error(tmp_path):
    with pytest.raises(E.EngineError, match="no CPU fallback"):
        E.load_library(str(tmp_path / "nope.so"))


def test_ibd0_ibd1_twins_match_the_oracle_and_the_reference_grid_bitwise(oracle):
    """ibdg_pdg_ibd0 / ibdg_pdg_ibd1 = find_pDgf / find_pDgIBD1 (src/ibd-math.c:84-142): the oracle's
    values (itself pinned to the reference's own calls over the grid, tests/test_oracle_golden.py) bit
    for bit over read counts x genotypes x frequencies, the untouched 1.0 for alleles other than 0/1, the
    r+a=0 short cut and the DBL_MIN clamp."""
    lib = E.load_library()
    C = oracle.lib
    fs = [0.0, 1e-3, 0.0137, 0.25, 1 / 3, 0.5, 0.731, 0.999, 1.0, 3 / 5008, 2504 / 5008]
    for eps, M in ((0.02, 20), (1e-3, 8), (0.3, 5)):
        for r in range(0, M + 1, max(1, M // 6)):
            for a in range(0, M + 1 - r, max(1, M // 6)):
                p = oracle.pdg(eps, M, r, a)
                for f in fs:
                    assert lib.ibdg_pdg_ibd0(f, *p).hex() == C.orc_pDgf(f, *p).hex(), (eps, r, a, f)
                    for a0, a1 in ((0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (0, 7)):
                        assert lib.ibdg_pdg_ibd1(a0, a1, f, *p).hex() == C.orc_pDgIBD1(a0, a1, f, *p).hex()
    assert lib.ibdg_pdg_ibd0(0.3, 1.0, 1.0, 1.0) == 1.0
    assert lib.ibdg_pdg_ibd0(0.3, 0.0, 0.0, 0.0) == 2.2250738585072014e-308
    assert lib.ibdg_pdg_ibd1(1, 1, 1.0, 0.0, 0.0, 0.0) == 2.2250738585072014e-308
    assert lib.ibdg_pdg_ibd1(3, 0, 0.3, 0.1, 0.2, 0.3) == 1.0


def test_bounded_poll_of_the_site_preparation():
    """ibdg_upload_sites waits for the device's hand-over by polling one host-mapped word; the poll ends when the word
    arrives, when the stream ends without it, when the stream fails, and -- a wedged stream -- when its wall-clock bound
    runs out (it used to spin for ever).  Exercised without a device through the library's self check."""
    lib = E.load_library()
    assert lib.ibdg_selftest(b"wait_info") == 0
    assert lib.ibdg_selftest(b"no such check") == 1
