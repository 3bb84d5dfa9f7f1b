"""Genotype ingest of the host program: the block-wise, multi-threaded
.hap reader of ibdgem_amd/host/ingest.c and its packed-panel cache.

Bars: the packed rows equal a plain numpy packing of the same text bit for bit, for any thread
count, for plain and gzip input; the "row is clean" flags and the all-zero rows follow
ibdg_pack_hap_text (short rows, characters other than 0/1 at an allele offset, empty lines, a last
line without newline); the cache file reproduces the same panel and is dropped when the .hap file
changes.  No device is needed (`--plan`)."""
import gzip
import os
import subprocess
import time

import numpy as np
import pytest

from test_host_cli import _exe


def write_inputs(tmp, N, L, seed, gz=False, mutate=True):
    rng = np.random.default_rng(seed)
    alle = (rng.random((L, 2 * N)) < 0.3).astype(np.uint8)
    lines = [" ".join("01"[b] for b in row) for row in alle]
    ok = np.ones(L, dtype=bool)
    if mutate:
        lines[3] = lines[3][: 4 * N - 2]                      # one character short
        ok[3] = False
        c = 4 * (N // 2)
        lines[7] = lines[7][:c] + "2" + lines[7][c + 1:]      # not 0/1 at an allele offset
        ok[7] = False
        lines[9] = lines[9][:c + 1] + "x" + lines[9][c + 2:]  # separator offsets are not looked at
        lines[11] = ""                                        # empty line
        ok[11] = False
        lines[13] = lines[13] + " trailing"                   # longer than needed: fine
        lines[15] = lines[15][:4 * (N - 1) + 2] + "?"         # bad last allele (scalar tail when N % 8)
        ok[15] = False
    text = "\n".join(lines)                                   # last line without newline
    hap = os.path.join(tmp, "p.hap" + (".gz" if gz else ""))
    with (gzip.open(hap, "wt") if gz else open(hap, "w")) as fh:
        fh.write(text)
    with open(os.path.join(tmp, "p.legend"), "w") as fh:
        fh.write("id position a0 a1\n")
        for i in range(L):
            fh.write(f"rs{i} {100 + 10 * i} A G\n")
    with open(os.path.join(tmp, "p.indv"), "w") as fh:
        for n in range(N):
            fh.write(f"ind{n}\n")
    with open(os.path.join(tmp, "p.pileup"), "w") as fh:
        for i in range(0, L, 3):
            fh.write(f"chr1\t{100 + 10 * i}\tN\t2\tAG\tII\t]]\n")
    return hap, alle, ok


def numpy_pack(alle, ok, N):
    L = alle.shape[0]
    words = 2 * ((N + 63) // 64)
    out = np.zeros((L, words), dtype=np.uint64)
    for n in range(N):
        w, bit = 2 * (n // 64), np.uint64(1) << np.uint64(n % 64)
        out[:, w] |= np.where(alle[:, 2 * n] == 1, bit, np.uint64(0))
        out[:, w + 1] |= np.where(alle[:, 2 * n + 1] == 1, bit, np.uint64(0))
    out[~ok] = 0
    return out


def run_dump(tmp, hap, threads, extra=()):
    dump = os.path.join(tmp, f"dump{threads}.bin")
    res = subprocess.run([_exe(), "-H", hap, "-L", "p.legend", "-I", "p.indv", "-P", "p.pileup", "-s", "ind0", "--plan",
                          "--threads", str(threads), "--dump-panel", dump, *extra], cwd=tmp, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return open(dump, "rb").read(), res.stdout


def split_dump(raw, L, N):
    words = 2 * ((N + 63) // 64)
    flags = np.frombuffer(raw[:L], dtype=np.uint8).astype(bool)
    rows = np.frombuffer(raw[L:], dtype=np.uint64).reshape(L, words)
    return flags, rows


@pytest.mark.parametrize("N,gz", [(131, False), (64, False), (200, True), (5, False)])
def test_packed_panel_equals_numpy_packing_for_any_thread_count(tmp_path, N, gz):
    L = 3000
    hap, alle, ok = write_inputs(str(tmp_path), N, L, seed=N, gz=gz)
    want = numpy_pack(alle, ok, N)
    ref_out = None
    for threads in (1, 3, 8):
        raw, out = run_dump(str(tmp_path), hap, threads)
        flags, rows = split_dump(raw, L, N)
        assert (flags == ok).all(), np.nonzero(flags != ok)
        assert (rows == want).all()
        ref_out = ref_out or out
        assert out == ref_out                                  # the plan (filter chain) does not depend on it


def test_rows_follow_the_library_row_packer(tmp_path):
    """Same flags and, for clean rows, the same words as ibdg_pack_hap_text (the per-row ABI function)."""
    import ctypes
    from ibdgem_amd import engine as E
    N, L = 77, 400
    hap, alle, ok = write_inputs(str(tmp_path), N, L, seed=5)
    raw, _ = run_dump(str(tmp_path), hap, 4)
    flags, rows = split_dump(raw, L, N)
    lib = E.load_library()
    words = lib.ibdg_row_words(N)
    for r, line in enumerate(open(hap).read().split("\n")):
        buf = (ctypes.c_uint64 * words)()
        rc = lib.ibdg_pack_hap_text(line.encode(), N, buf)
        assert (rc == 0) == bool(flags[r]), r
        if rc == 0:
            assert list(buf) == list(rows[r]), r


def test_more_hap_rows_than_legend_rows_and_the_reverse(tmp_path):
    N, L = 20, 50
    hap, alle, ok = write_inputs(str(tmp_path), N, L, seed=9, mutate=False)
    for keep in (50, 30):
        lines = open(os.path.join(tmp_path, "p.legend")).read().split("\n")
        with open(os.path.join(tmp_path, "p.legend"), "w") as fh:
            fh.write("\n".join(lines[: keep + 1]) + "\n")
        raw, out = run_dump(str(tmp_path), hap, 2)
        assert len(raw) == keep * (1 + 8 * 2)                   # rows = the shorter of the two files
    with open(hap, "w") as fh:                                  # now the .hap file is the shorter one
        fh.write("\n".join(" ".join("01"[b] for b in row) for row in alle[:12]) + "\n")
    raw, out = run_dump(str(tmp_path), hap, 2)
    assert len(raw) == 12 * (1 + 8 * 2)


def test_panel_cache_round_trip_and_invalidation(tmp_path):
    N, L = 131, 2000
    hap, alle, ok = write_inputs(str(tmp_path), N, L, seed=21)
    cache = os.path.join(tmp_path, "panel.cache")
    raw0, out0 = run_dump(str(tmp_path), hap, 4)
    raw1, out1 = run_dump(str(tmp_path), hap, 4, ("--panel-cache", cache))      # builds the cache
    assert os.path.exists(cache) and raw1 == raw0 and out1 == out0
    stamp = os.stat(cache).st_mtime_ns
    raw2, out2 = run_dump(str(tmp_path), hap, 4, ("--panel-cache", cache))      # reads it
    assert raw2 == raw0 and out2 == out0 and os.stat(cache).st_mtime_ns == stamp
    # a changed .hap file (other content, other mtime) must not be served from the old cache
    time.sleep(0.01)
    hap, alle2, ok2 = write_inputs(str(tmp_path), N, L, seed=22)
    raw3, _ = run_dump(str(tmp_path), hap, 4, ("--panel-cache", cache))
    flags, rows = split_dump(raw3, L, N)
    assert (rows == numpy_pack(alle2, ok2, N)).all() and os.stat(cache).st_mtime_ns != stamp
    # a cache for another panel width is ignored as well
    with open(os.path.join(tmp_path, "p.indv"), "a") as fh:
        fh.write("extra\n")
    res = subprocess.run([_exe(), "-H", hap, "-L", "p.legend", "-I", "p.indv", "-P", "p.pileup", "-s", "ind0", "--plan",
                          "--panel-cache", cache], cwd=tmp_path, capture_output=True, text=True)
    assert res.returncode == 0


def test_row_formatter_output_does_not_depend_on_the_thread_count(tmp_path):
    """Output formatter (§8(f) rank 2): the per-site rows are formatted by a team of threads into
    buffers written in order -- the text is the same for any team size (here: the `--plan` rows;
    the full files are compared byte for byte with the reference's in test_host_cli.py on a GPU)."""
    N, L = 12, 30000
    hap, alle, ok = write_inputs(str(tmp_path), N, L, seed=31, mutate=False)
    with open(os.path.join(tmp_path, "p.pileup"), "w") as fh:
        for i in range(L):
            fh.write(f"chr1\t{100 + 10 * i}\tN\t3\tAGA\tIII\t]]]\n")
    outs = [run_dump(str(tmp_path), hap, th)[1] for th in (1, 5, 16)]
    assert outs[0].count("\n") > L and outs[0] == outs[1] == outs[2]
