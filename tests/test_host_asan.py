"""The host C programs under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers on
the host build; the GPU pool offers none for device code).

`make -C ibdgem_amd/host asan` builds ibdgem_asan / hiddengem_asan; the CPU-tier host tests (option
handling, the integer plan of every golden case, genotype ingest for every thread count, hiddengem's
golden tables) are then rerun against those binaries in a child pytest.  Any sanitizer report aborts
the program, which fails the test that ran it."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(REPO, "ibdgem_amd", "host")


@pytest.mark.skipif(os.environ.get("IBDGEM_EXE") is not None, reason="already inside the sanitizer rerun")
def test_host_tests_pass_under_asan_and_ubsan():
    subprocess.run(["make", "-C", os.path.join(REPO, "ibdgem_amd", "csrc")], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", HOST, "asan"], check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ,
               IBDGEM_EXE=os.path.join(HOST, "ibdgem_asan"), HIDDENGEM_EXE=os.path.join(HOST, "hiddengem_asan"),
               # the HIP runtime the engine library pulls in keeps allocations until exit: leak checking off
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        "tests/test_host_cli.py", "tests/test_host_ingest.py", "tests/test_hiddengem.py"],
                       cwd=REPO, env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_threaded_host_paths_are_race_free_under_tsan():
    """ThreadSanitizer build of the host program: golden cases through the threaded pileup / legend readers,
    the threaded row filter chain and the threaded .hap reader (IBDGEM_MT_MIN_BYTES=1 sends small files that
    way), several team sizes: no report, same stdout as the plain build."""
    import golden_io as G
    subprocess.run(["make", "-C", os.path.join(REPO, "ibdgem_amd", "csrc")], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", HOST, "ibdgem_tsan", "ibdgem"], check=True, stdout=subprocess.DEVNULL)
    meta = G.cases("synA")
    cwd = os.path.join(G.GOLD, "synA", "input")
    env = dict(os.environ, IBDGEM_MT_MIN_BYTES="1", TSAN_OPTIONS="halt_on_error=1:exitcode=66")
    for case in list(meta["cases"])[:6]:
        for th in ("3", "8"):
            args = meta["base_args"] + meta["cases"][case] + ["--plan", "--threads", th]
            want = subprocess.run([os.path.join(HOST, "ibdgem")] + args, cwd=cwd, capture_output=True, text=True)
            got = subprocess.run([os.path.join(HOST, "ibdgem_tsan")] + args, cwd=cwd, capture_output=True, text=True, env=env)
            assert "ThreadSanitizer" not in got.stderr, got.stderr[-2000:]
            assert got.returncode == want.returncode == 0
            assert got.stdout == want.stdout


def test_output_files_of_several_individuals_written_side_by_side_under_tsan(tmp_path):
    """The per-site tables of up to four comparison individuals are written beside the main thread's work on the ones
    after them, each through the ordered formatter pipeline (IBDGEM_MT_MIN_BYTES=1: chunks of 37 rows, short ring):
    no ThreadSanitizer report on the reference's own fixture (3 individuals, no device: non-LD) and on a synthetic case
    with more of them, and the files are the golden ones."""
    import golden_io as G
    import gzip
    subprocess.run(["make", "-C", os.path.join(REPO, "ibdgem_amd", "csrc")], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", HOST, "ibdgem_tsan"], check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ, IBDGEM_MT_MIN_BYTES="1", TSAN_OPTIONS="halt_on_error=1:exitcode=66", HIP_VISIBLE_DEVICES="",
               ROCR_VISIBLE_DEVICES="", IBDGEM_KEEP_TEARDOWN="1")
    fix_in = os.path.join(G.GOLD, "ibdgem-test", "input")
    args = ["-H", "test.hap", "-L", "test.legend", "-I", "test.indv", "-P", "test1.pileup", "-N", "sample1", "--threads", "4",
            "-O", str(tmp_path)]
    for slots in ("3", "2", "1"):                    # fewer slots: an individual waits for the files of the one before it
        got = subprocess.run([os.path.join(HOST, "ibdgem_tsan")] + args, cwd=fix_in, capture_output=True, text=True,
                             env=dict(env, IBDGEM_OUT_SLOTS=slots))
        assert "ThreadSanitizer" not in got.stderr, got.stderr[-2000:]
        assert got.returncode == 0, got.stderr[-2000:]
        for t in (1, 2, 3):
            for kind in ("tab", "summary"):
                fn = f"sample1.sample{t}.{kind}.txt"
                a = open(tmp_path / fn).read().splitlines()
                b = open(os.path.join(G.GOLD, "ibdgem-test", "output", fn)).read().splitlines()
                assert a[1:] == b[1:], fn
                os.remove(tmp_path / fn)
    meta = G.cases("synA")
    case = "nonld_all_targets_w2"
    out2 = tmp_path / "syn"
    out2.mkdir()
    got = subprocess.run([os.path.join(HOST, "ibdgem_tsan")] + meta["base_args"] + meta["cases"][case] +
                         ["--threads", "5", "-O", str(out2)], cwd=os.path.join(G.GOLD, "synA", "input"), capture_output=True,
                         text=True, env=env)
    assert "ThreadSanitizer" not in got.stderr, got.stderr[-2000:]
    assert got.returncode == 0, got.stderr[-2000:]
    ref = os.path.join(G.GOLD, "synA", case, "ref7")
    n_tab = 0
    for fn in sorted(os.listdir(ref)):
        want = gzip.open(os.path.join(ref, fn), "rt").read().splitlines()
        have = open(out2 / fn[:-3]).read().splitlines()
        if fn.endswith(".tab.txt.gz"):
            have = have[1:]
            n_tab += 1
        assert have == want, fn
    assert n_tab >= 2
