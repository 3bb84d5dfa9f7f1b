"""The C host program (ibdgem_amd/host/ibdgem): reference CLI, parsers, row filter chain,
window boundaries and output files.

CPU part (`--plan`, no device): every integer column, the AF text, the processed/skipped
counters, the coverage histograms and the window boundaries must equal the reference's
files for all golden cases (flags -v -D -A -p -c -M -F -f -e -w -B -S -s -N exercised).
GPU part: the full output files must be byte-identical to what the unmodified reference
wrote (fixtures: its 18 shipped files; synthetic: tests/golden/syn*/<case>/ref7)."""
import gzip
import os
import subprocess

import pytest

import golden_io as G

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# IBDGEM_EXE: another build of the same program (tests/test_host_asan.py points it at the sanitizer build)
EXE = os.environ.get("IBDGEM_EXE") or os.path.join(REPO, "ibdgem_amd", "host", "ibdgem")


def _exe():
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.join(REPO, "ibdgem_amd", "csrc")], check=True, stdout=subprocess.DEVNULL)
        subprocess.run(["make", "-C", os.path.join(REPO, "ibdgem_amd", "host")], check=True, stdout=subprocess.DEVNULL)
    return EXE


def parse_plan(text):
    """stdout of --plan -> {target: dict(rows, comments, windows, processed, skipped)}"""
    out, cur = {}, None
    for line in text.splitlines():
        if line.startswith("## PLAN "):
            f = line.split()
            cur = dict(rows=[], comments=[], windows=[], processed=int(f[4].split("=")[1]),
                       skipped=int(f[5].split("=")[1]))
            out[f[3]] = cur
        elif line.startswith("## WINDOW "):
            cur["windows"].append(line.split("\t")[1:])
        elif line.startswith("#"):
            cur["comments"].append(line)
        elif line:
            cur["rows"].append(line.split("\t"))
    return out


def check_plan_against(plan, tab, summ):
    assert plan["processed"] == tab.processed and plan["skipped"] == tab.skipped
    assert len(plan["rows"]) == len(tab.rows)
    for got, want in zip(plan["rows"], tab.rows):
        assert got == want[:11], (got, want)          # CHR rsID POS REF ALT AF DP SQ_NREF SQ_NALT GT_A0 GT_A1
    want_c = [c for c in tab.comments if not c.startswith("# Entered command")]
    assert plan["comments"] == want_c                  # histograms, mean depth, cull ratio, counters
    assert len(plan["windows"]) == len(summ.start)
    for w, (s, e, n) in enumerate(plan["windows"]):
        assert (int(s), int(e), int(n)) == (summ.start[w], summ.end[w], summ.nsites[w])


FIX_IN = os.path.join(G.GOLD, "ibdgem-test", "input")


@pytest.mark.parametrize("k", [1, 2, 3])
def test_plan_matches_reference_fixture_files(k):
    res = subprocess.run([_exe(), "-H", "test.hap", "-L", "test.legend", "-I", "test.indv", "-P", f"test{k}.pileup",
                          "-N", f"sample{k}", "--plan"], cwd=FIX_IN, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    plan = parse_plan(res.stdout)
    assert sorted(plan) == ["sample1", "sample2", "sample3"]
    for t in plan:
        tab, summ = G.fixture_outputs(f"sample{k}", t)
        check_plan_against(plan[t], tab, summ)
    assert "Running sample%d-vs-sample1 comparison..." % k in res.stderr and "Run time:" in res.stderr


def _syn_cases():
    return [(tag, case) for tag in ("synA", "synB") for case in G.cases(tag)["cases"]]


@pytest.mark.parametrize("tag,case", _syn_cases())
def test_plan_matches_reference_on_synthetic_cases(tag, case):
    meta = G.cases(tag)
    args = meta["base_args"] + meta["cases"][case]
    res = subprocess.run([_exe()] + args + ["--plan"], cwd=os.path.join(G.GOLD, tag, "input"), capture_output=True,
                         text=True)
    assert res.returncode == 0, res.stderr
    plan = parse_plan(res.stdout)
    flags = G.parse_flags(meta["cases"][case])
    files = sorted(f for f in os.listdir(os.path.join(G.GOLD, tag, case)) if f.endswith(".tab.txt.gz"))
    assert len(files) == len(plan) > 0
    for fn in files:
        name = fn.split(".")[1]
        tab, summ = G.syn_outputs(tag, case, flags["sq"], name)
        check_plan_against(plan[name], tab, summ)


@pytest.mark.parametrize("case", ["vcf_ld", "vcf_nonld_q30", "vcf_ld_varsites_w50"])
def test_plan_matches_reference_on_vcf_input(case):
    meta = G.cases("synV")
    res = subprocess.run([_exe()] + meta["base_args"] + meta["cases"][case] + ["--plan"],
                         cwd=os.path.join(G.GOLD, "synV", "input"), capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    plan = parse_plan(res.stdout)
    flags = G.parse_flags(meta["cases"][case])
    (name,) = plan
    tab, summ = G.syn_outputs("synV", case, flags["sq"], name)
    check_plan_against(plan[name], tab, summ)
    assert "Failed to parse genotype fields at" in res.stderr


def test_option_errors_match_the_reference():
    exe = _exe()

    def run(*a):
        return subprocess.run([exe, *a], cwd=FIX_IN, capture_output=True, text=True)
    base = ["-H", "test.hap", "-L", "test.legend", "-I", "test.indv", "-P", "test1.pileup"]
    r = run(*base, "-w", "1")
    assert r.returncode == 0 and "[::] ERROR: Invalid window size (-w) of 1 (must be >= 2)." in r.stderr
    r = run(*base, "-M", "0")
    assert r.returncode == 0 and "(-M) of 0 (must be >= 1)" in r.stderr
    r = run(*base, "-F", "1.5")
    assert r.returncode == 0 and "(-F) of 1.50 (must be <= 1)" in r.stderr
    r = run(*base, "-D", "-1")
    assert r.returncode == 0 and "(-D) of -1.00 (must be > 0)" in r.stderr
    r = run(*base, "-w")
    assert r.returncode == 0 and "Option -w missing required argument." in r.stderr
    r = run("-P", "test1.pileup")
    assert r.returncode == 1 and "[::] ERROR: Missing genotype files." in r.stderr
    r = run(*base, "-V", "x.vcf")
    assert r.returncode == 1 and "2 types of genotype inputs detected" in r.stderr
    r = run("-V", "test.legend", "-P", "test1.pileup")
    assert r.returncode == 1 and "[::] ERROR parsing VCF header." in r.stderr
    r = run(*base, "-s", "nobody", "--plan")
    assert r.returncode == 1 and "Sample nobody not found in reference panel." in r.stderr
    r = run("-H", "test.hap", "-L", "test.legend", "-I", "test.indv", "-P", "missing.pileup")
    assert r.returncode == 1 and "ERROR parsing Pileup data" in r.stderr
    r = run()
    assert r.returncode == 0 and "Usage:" in r.stderr


def test_background_list_naming_someone_256_times_is_refused(tmp_path):
    """The engine's background multiplicities are one byte each (include/ibdgem_hip.h): 255 listings of
    one individual pass, 256 are an error -- never a silent truncation."""
    base = ["-H", "test.hap", "-L", "test.legend", "-I", "test.indv", "-P", "test1.pileup", "--LD", "--plan"]
    for n, ok in ((255, True), (256, False)):
        lst = tmp_path / f"bg{n}.txt"
        lst.write_text("sample2\n" + "sample3\n" * n)
        r = subprocess.run([_exe(), *base, "-B", str(lst)], cwd=FIX_IN, capture_output=True, text=True)
        if ok:
            assert r.returncode == 0, r.stderr
        else:
            assert r.returncode == 1 and "sample3 is listed more than 255 times" in r.stderr


def test_read_thinning_stream_is_glibc_rand():
    """-D thins reads with the reference's unseeded rand() stream (src/ibdgem.c:132); the host
    restates glibc's generator so that the HIP runtime in the same process cannot disturb it."""
    import ctypes
    n = 20000
    got = subprocess.run([_exe(), "--rand-stream", str(n)], capture_output=True, text=True).stdout.split()
    code = ("import ctypes\nl = ctypes.CDLL(None)\nprint(' '.join(str(l.rand()) for _ in range(%d)))" % n)
    want = subprocess.run(["python3", "-c", code], capture_output=True, text=True).stdout.split()   # fresh process
    assert len(got) == n and got == want


def test_pileup_parser_edge_cases(tmp_path):
    """The byte-matching rule for read counts (reference src/pileup.c:253-415, src/ibdgem.c:620-621)."""
    d = tmp_path
    (d / "p.indv").write_text("a\nb\n")
    (d / "p.legend").write_text("ID pos allele0 allele1\n" + "".join(f"r{i} {10 * i} A C\n" for i in range(1, 9)))
    (d / "p.hap").write_text("0 1 1 0\n" * 8)
    (d / "p.pileup").write_text(
        "1\t10\tA\t4\t.,Cc\tIIII\tIIII\n"           # '.' ',' take the pileup's REF letter A -> 2 ref, 2 alt
        "1\t20\ta\t3\t..C\tIII\tIII\n"               # lowercase REF column: '.' never matches 'A'
        "1\t30\tN\t3\tA^]A$+2ACc-1g\tIII\tIII\n"    # read start/end markers and indels are skipped
        "1\t40\tN\t2\tA*\tII\tII\n"                  # '*' counts towards DP only
        "1\t50\tN\t2\tAA\tI\tI\n"                    # both quality strings of the wrong length: line dropped
        "1\t60\tN\t3\tAA\tIII\tIII\n"                # fewer bases than DP: dropped
        "1\t70\tN\t1\tA\tI\n"                        # six columns only: dropped
        "1\t80\tN\t0\t*\t*\t*\n")
    res = subprocess.run([_exe(), "-H", "p.hap", "-L", "p.legend", "-I", "p.indv", "-P", "p.pileup", "-s", "a", "--plan"],
                         cwd=d, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    rows = {int(r[2]): r for r in parse_plan(res.stdout)["a"]["rows"]}
    assert sorted(rows) == [10, 20, 30, 40, 80]
    assert rows[10][6:9] == ["4", "2", "2"]
    assert rows[20][6:9] == ["3", "0", "1"]
    assert rows[30][6:9] == ["3", "2", "1"]
    assert rows[40][6:9] == ["2", "1", "0"]
    assert rows[80][6:9] == ["0", "0", "0"]
    assert "Incorrect number of bases read in" in res.stderr


# ------------------------------------------------------------------------------------------- no device (configs[0])
def _run_no_device(args, cwd, out, expect_ok=True):
    """The host program on a machine without a HIP device (none visible): non-LD runs take the per-row values and
    window products from the library's host twins of the reference's math seam (reference src/ibd-math.h:14-63)."""
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", IBDGEM_KEEP_TEARDOWN="1")
    res = subprocess.run([_exe()] + args + ["-O", str(out)], cwd=cwd, capture_output=True, text=True, env=env)
    if expect_ok:
        assert res.returncode == 0, res.stderr
    return res


@pytest.mark.parametrize("k", [1, 2, 3])
def test_no_device_run_reproduces_the_reference_fixture_files(k, tmp_path):
    """BASELINE configs[0]: supplementary/ibdgem-test, non-LD, IMPUTE input, without a GPU -- all 18 files
    byte for byte (line 1 of the tab files, the echoed command, aside)."""
    res = _run_no_device(["-H", "test.hap", "-L", "test.legend", "-I", "test.indv", "-P", f"test{k}.pileup", "-N", f"sample{k}"],
                         FIX_IN, tmp_path)
    assert "computed on the host" in res.stderr
    for t in (1, 2, 3):
        for kind in ("tab", "summary"):
            fn = f"sample{k}.sample{t}.{kind}.txt"
            got = _read(str(tmp_path / fn))
            want = _read(os.path.join(G.GOLD, "ibdgem-test", "output", fn))
            if kind == "tab":
                assert got[0].startswith("# Entered command: ") and got[0].endswith(" ")
                got, want = got[1:], want[1:]
            assert got == want, fn


@pytest.mark.parametrize("tag,case", _syn_cases() + [("synV", c) for c in sorted(G.cases("synV")["cases"])])
def test_no_device_run_on_synthetic_cases(tag, case, tmp_path):
    """Non-LD cases: every output file equals the reference's.  --LD cases run here WITHOUT --LD: the per-site
    table of the reference does not depend on --LD (src/ibdgem.c:731-733 prints ibd0/1/2 of the row), so their tab
    files must still match; with --LD and no device the program stops with the engine's message."""
    meta = G.cases(tag)
    args = meta["base_args"] + meta["cases"][case]
    ld = "--LD" in args
    inp = os.path.join(G.GOLD, tag, "input")
    if ld:
        res = _run_no_device(args, inp, tmp_path, expect_ok=False)
        assert res.returncode != 0 and "no HIP device" in res.stderr
        args = [a for a in args if a != "--LD"]
    _run_no_device(args, inp, tmp_path)
    ref = os.path.join(G.GOLD, tag, case, "ref7")
    for fn in sorted(os.listdir(ref)):
        if ld and not fn.endswith(".tab.txt.gz"):
            continue
        got = _read(str(tmp_path / fn[:-3]))
        want = _read(os.path.join(ref, fn))
        if fn.endswith(".tab.txt.gz"):
            got = got[1:]
        assert got == want, f"{tag}/{case}/{fn}"


def test_no_device_summary_only_and_threads(tmp_path):
    meta = G.cases("synA")
    args = meta["base_args"] + meta["cases"]["nonld_all_targets_w2"]
    inp = os.path.join(G.GOLD, "synA", "input")
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    _run_no_device(args, inp, tmp_path / "a")
    _run_no_device(args + ["--summary-only", "--threads", "3"], inp, tmp_path / "b")
    names = sorted(os.listdir(tmp_path / "a"))
    assert sorted(os.listdir(tmp_path / "b")) == [n for n in names if n.endswith(".summary.txt")]
    for n in os.listdir(tmp_path / "b"):
        assert _read(str(tmp_path / "a" / n)) == _read(str(tmp_path / "b" / n))


def test_a_failing_run_leaves_no_stale_tables_behind(tmp_path):
    """The output files are opened without truncation and emptied by whoever writes them (src/ibdgem.c:529-530 truncates at
    fopen).  A run that fails after it opened an individual's files must not leave an earlier run's complete-looking table
    in place: the exit path empties what was opened and not yet written, after waiting for every thread it started."""
    fix = ["-H", "test.hap", "-L", "test.legend", "-I", "test.indv", "-P", "test1.pileup", "-N", "sample1"]
    _run_no_device(fix, FIX_IN, tmp_path)                                  # a complete earlier run
    tab2 = tmp_path / "sample1.sample2.tab.txt"
    assert tab2.stat().st_size > 1000
    first = _read(str(tmp_path / "sample1.sample1.summary.txt"))
    os.remove(tmp_path / "sample1.sample2.summary.txt")
    os.mkdir(tmp_path / "sample1.sample2.summary.txt")                     # the second individual's summary cannot be opened
    res = _run_no_device(fix, FIX_IN, tmp_path, expect_ok=False)
    assert res.returncode == 1 and "Cannot open" in res.stderr
    assert tab2.stat().st_size == 0, "the stale table of the individual whose files could not be opened survived"
    assert _read(str(tmp_path / "sample1.sample1.summary.txt")) == first     # the individual before it is complete


@pytest.mark.skipif(not os.path.exists(os.path.join(REPO, "oracle", "_ref", "ibdgem")),
                    reason="the reference binary (oracle/_ref/ibdgem) is not in this tree")
def test_no_device_runs_against_the_reference_binary_on_random_inputs():
    """The device-less non-LD run on random panels, pileups and flags (-v -D -M -F -f -w -e -c -p -A -N ...): every
    output file of the host program byte for byte the reference binary's (tools/fuzz_cli_full.py --no-device)."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "fuzz_cli_full.py"), "80", "7", "--no-device"], cwd=REPO,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "full CLI fuzz: 80 cases" in r.stdout and "0 failures" in r.stdout.splitlines()[-1], r.stdout[-2000:]


# ------------------------------------------------------------------------------------------- GPU
def _run_full(args, cwd, out):
    res = subprocess.run([_exe()] + args + ["-O", str(out)], cwd=cwd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return res


def _read(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as fh:
        return fh.read().splitlines()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 3])
def test_cli_reproduces_the_reference_fixture_files(k, tmp_path):
    _run_full(["-H", "test.hap", "-L", "test.legend", "-I", "test.indv", "-P", f"test{k}.pileup", "-N", f"sample{k}"],
              FIX_IN, tmp_path)
    for t in (1, 2, 3):
        for kind in ("tab", "summary"):
            fn = f"sample{k}.sample{t}.{kind}.txt"
            got = _read(str(tmp_path / fn))
            want = _read(os.path.join(G.GOLD, "ibdgem-test", "output", fn))
            if kind == "tab":
                assert got[0].startswith("# Entered command: ") and got[0].endswith(" ")
                got, want = got[1:], want[1:]
            assert got == want, fn


@pytest.mark.gpu
@pytest.mark.parametrize("tag,case", _syn_cases())
def test_cli_reproduces_the_reference_on_synthetic_cases(tag, case, tmp_path):
    meta = G.cases(tag)
    _run_full(meta["base_args"] + meta["cases"][case], os.path.join(G.GOLD, tag, "input"), tmp_path)
    ref = os.path.join(G.GOLD, tag, case, "ref7")
    files = sorted(os.listdir(ref))
    assert files and sorted(os.listdir(tmp_path)) == [f[:-3] for f in files]
    for fn in files:
        got = _read(str(tmp_path / fn[:-3]))
        want = _read(os.path.join(ref, fn))
        if fn.endswith(".tab.txt.gz"):
            got = got[1:]
        assert got == want, f"{tag}/{case}/{fn}"


@pytest.mark.gpu
@pytest.mark.parametrize("tag,case,extra", [("synA", "ld_default", []), ("synA", "ld_varsites", []), ("synB", "ld_w37", ["--devices", "0,0,0"]),
                                            ("synA", "nonld_all_targets_w2", []), ("synA", "ld_default", ["--summary-only"])])
def test_cli_with_the_panel_cache_as_a_file(tag, case, extra, tmp_path):
    """--panel-cache: the first run packs the .hap text and writes the cache, the second hands the engine the cache FILE
    (ibdg_upload_panel_fd: its staging threads read the rows themselves, nothing is mapped on the host side unless a row's
    alleles are asked for -- the per-site table, -v) -- with --devices every context its byte range of the file.  Both runs
    write the reference's files byte for byte."""
    meta = G.cases(tag)
    cache = str(tmp_path / "panel.cache")
    ref = os.path.join(G.GOLD, tag, case, "ref7")
    files = sorted(f for f in os.listdir(ref) if "--summary-only" not in extra or f.endswith(".summary.txt.gz"))
    for turn in ("cache written", "cache read"):
        out = tmp_path / turn.replace(" ", "_")
        out.mkdir()
        _run_full(meta["base_args"] + meta["cases"][case] + extra + ["--panel-cache", cache], os.path.join(G.GOLD, tag, "input"), out)
        assert os.path.getsize(cache) > 0
        assert files and sorted(os.listdir(out)) == [f[:-3] for f in files], turn
        for fn in files:
            got = _read(str(out / fn[:-3]))
            want = _read(os.path.join(ref, fn))
            if fn.endswith(".tab.txt.gz"):
                got = got[1:]
            assert got == want, f"{turn}: {tag}/{case}/{fn}"


@pytest.mark.gpu
@pytest.mark.parametrize("tag,case,devices", [("synA", "ld_default", "0,0"), ("synA", "ld_downsample", "0,0,0"),
                                              ("synA", "ld_varsites", "0,0,0,0,0"), ("synA", "nonld_all_targets_w2", "0,0,0"),
                                              ("synA", "ld_bg20_w64", "0,0,0,0"), ("synA", "ld_pu_in_panel", "0,0,0"),
                                              ("synA", "ld_af_file", "0,0"), ("synA", "ld_positions", "0,0,0,0,0,0"),
                                              ("synB", "ld_w37", "0,0,0,0,0,0,0"), ("synB", "ld_pu_named", "0,0")])
def test_cli_window_sharding_over_several_contexts(tag, case, devices, tmp_path):
    """--devices: the windows of each comparison are cut into contiguous ranges, one per engine
    context (here several contexts on the one GPU of the test box), evaluated by one host thread
    each and gathered on the host -- output files still byte-identical to the reference's.  Without -v / -D
    (one site list for all comparison individuals) a context holds only the panel rows of its window range
    (windows are independent, reference src/ibdgem.c:558-570); with them, the whole panel."""
    meta = G.cases(tag)
    args = meta["base_args"] + meta["cases"][case] + ["--devices", devices]
    res = subprocess.run([_exe()] + args + ["-O", str(tmp_path)], cwd=os.path.join(G.GOLD, tag, "input"),
                         capture_output=True, text=True, env=dict(os.environ, IBDGEM_TIMING="1"))
    assert res.returncode == 0, res.stderr
    slices = [l.split() for l in res.stderr.splitlines() if l.startswith("## panel slice of device")]
    n_dev = devices.count(",") + 1
    if "-v" in args:
        assert not slices
    elif "-D" not in args or slices:         # (-D thins the reads only when the depth is above the target)
        assert len(slices) == n_dev
        total = int(slices[0][-1])
        first = [int(x[7]) for x in slices]          # "## panel slice of device D: rows R0 + N of TOTAL"
        count = [int(x[9]) for x in slices]
        live = [(a, c) for a, c in zip(first, count) if c]
        # contiguous window ranges: ascending, and only the two rows at a cut may be shared... no row twice
        assert all(a + c <= b for (a, c), (b, _) in zip(live, live[1:]))
        assert sum(count) <= total and (n_dev == 1 or max(count) < total)
    ref = os.path.join(G.GOLD, tag, case, "ref7")
    for fn in sorted(os.listdir(ref)):
        got = _read(str(tmp_path / fn[:-3]))
        want = _read(os.path.join(ref, fn))
        if fn.endswith(".tab.txt.gz"):
            got = got[1:]
        assert got == want, f"{case}/{fn} with --devices {devices}"


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["vcf_ld", "vcf_nonld_q30", "vcf_ld_varsites_w50"])
def test_cli_reproduces_the_reference_on_vcf_input(case, tmp_path):
    meta = G.cases("synV")
    _run_full(meta["base_args"] + meta["cases"][case], os.path.join(G.GOLD, "synV", "input"), tmp_path)
    ref = os.path.join(G.GOLD, "synV", case, "ref7")
    files = sorted(os.listdir(ref))
    assert files and sorted(os.listdir(tmp_path)) == [f[:-3] for f in files]
    for fn in files:
        got = _read(str(tmp_path / fn[:-3]))
        want = _read(os.path.join(ref, fn))
        if fn.endswith(".tab.txt.gz"):
            got = got[1:]
        assert got == want, f"synV/{case}/{fn}"


@pytest.mark.gpu
def test_cli_vcf_with_several_comparison_individuals(tmp_path):
    """The reference crashes on the second individual of a VCF run; here the VCF path gives, row by
    row, what the IMPUTE path gives for the same genotypes (synV's VCF = synA's panel plus rows the
    VCF path must skip)."""
    metaV, metaA = G.cases("synV"), G.cases("synA")
    outV, outA = tmp_path / "v", tmp_path / "a"
    outV.mkdir()
    outA.mkdir()
    _run_full(metaV["base_args"] + ["--LD", "-s", "ind3,ind64,ind9"], os.path.join(G.GOLD, "synV", "input"), outV)
    _run_full(metaA["base_args"] + ["--LD", "-s", "ind3,ind64,ind9"], os.path.join(G.GOLD, "synA", "input"), outA)
    for name in ("ind3", "ind64", "ind9"):
        tv = {l.split("\t")[2]: l for l in _read(str(outV / f"UNKWN.{name}.tab.txt")) if l and l[0] != "#"}
        ta = {l.split("\t")[2]: l for l in _read(str(outA / f"UNKWN.{name}.tab.txt")) if l and l[0] != "#"}
        assert 900 < len(tv) < len(ta)
        for pos, l in tv.items():
            assert ta[pos] == l
        assert len(_read(str(outV / f"UNKWN.{name}.summary.txt"))) > 3


@pytest.mark.gpu
@pytest.mark.parametrize("devices", ["0", "0,0,0"])
def test_cli_batches_comparison_individuals_when_their_site_lists_coincide(devices, tmp_path):
    """No -v and no -D: the host hands the engine up to 16 comparison individuals per call (groups of
    four share a workgroup of the --LD kernel).  The files of the two individuals the reference was
    run for are byte-identical to its own, and every file equals the one of a run for that individual
    alone (after the command line)."""
    meta = G.cases("synA")
    cwd = os.path.join(G.GOLD, "synA", "input")
    names = ["ind3", "ind10", "ind11", "ind64", "ind20", "ind0", "ind5", "ind47", "ind31"]
    batch = tmp_path / "batch"
    batch.mkdir()
    _run_full(meta["base_args"] + ["--LD", "-s", ",".join(names), "--devices", devices], cwd, batch)
    ref = os.path.join(G.GOLD, "synA", "ld_default", "ref7")
    for fn in sorted(os.listdir(ref)):
        got, want = _read(str(batch / fn[:-3])), _read(os.path.join(ref, fn))
        assert (got[1:] if fn.endswith(".tab.txt.gz") else got) == want, fn
    for name in names[1:3] + names[-1:]:
        solo = tmp_path / name
        solo.mkdir()
        _run_full(meta["base_args"] + ["--LD", "-s", name], cwd, solo)
        for kind in ("tab", "summary"):
            fn = f"UNKWN.{name}.{kind}.txt"
            a, b = _read(str(batch / fn)), _read(str(solo / fn))
            assert a[1:] == b[1:], fn


@pytest.mark.gpu
def test_cli_summary_only_writes_the_same_summary_files_and_no_tab_files(tmp_path):
    """--summary-only (an addition to the reference's options, for all-against-all runs whose
    per-site tables would be hundreds of GB): the summary files are the reference's, byte for byte."""
    meta = G.cases("synA")
    _run_full(meta["base_args"] + meta["cases"]["ld_default"] + ["--summary-only"], os.path.join(G.GOLD, "synA", "input"),
              tmp_path)
    ref = os.path.join(G.GOLD, "synA", "ld_default", "ref7")
    want = sorted(f[:-3] for f in os.listdir(ref) if ".summary." in f)
    assert sorted(os.listdir(tmp_path)) == want
    for fn in want:
        assert _read(str(tmp_path / fn)) == _read(os.path.join(ref, fn + ".gz")), fn


@pytest.mark.gpu
def test_cli_summary_only_over_all_individuals_of_the_panel(tmp_path):
    """The whole-panel job (one pileup against every individual, src/ibdgem.c:522) with --summary-only: batches of 30, the
    next batch queued on the device while this one's summary files are written, twelve files at a time -- every summary
    file equals the one of a run that writes tables too (no look-ahead, four files at a time), and the two the reference
    was run for are its own, byte for byte."""
    meta = G.cases("synA")
    cwd = os.path.join(G.GOLD, "synA", "input")
    a, b = tmp_path / "summaries", tmp_path / "tables"
    a.mkdir()
    b.mkdir()
    _run_full(meta["base_args"] + ["--LD", "--summary-only"], cwd, a)          # no -s: every individual of the panel
    _run_full(meta["base_args"] + ["--LD"], cwd, b)
    want = sorted(f for f in os.listdir(b) if f.endswith(".summary.txt"))
    assert len(want) > 60 and sorted(os.listdir(a)) == want
    for fn in want:
        assert _read(str(a / fn)) == _read(str(b / fn)), fn
    ref = os.path.join(G.GOLD, "synA", "ld_default", "ref7")
    for fn in sorted(os.listdir(ref)):
        if ".summary." in fn:
            assert _read(str(a / fn[:-3])) == _read(os.path.join(ref, fn)), fn


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["ld_default", "ld_bg_dup", "ld_bg_self_nan", "ld_pu_in_panel"])
def test_cli_reference_order_mode(case, tmp_path):
    """--reference-order: the background sums are taken serially in the reference's order (list order of
    -B, duplicates kept), so the --LD columns are the reference's bits, not merely its 7 digits."""
    meta = G.cases("synA")
    _run_full(meta["base_args"] + meta["cases"][case] + ["--reference-order"], os.path.join(G.GOLD, "synA", "input"),
              tmp_path)
    ref = os.path.join(G.GOLD, "synA", case, "ref7")
    for fn in sorted(os.listdir(ref)):
        got, want = _read(str(tmp_path / fn[:-3])), _read(os.path.join(ref, fn))
        assert (got[1:] if fn.endswith(".tab.txt.gz") else got) == want, fn


@pytest.mark.parametrize("tag,case", [("synA", c) for c in G.cases("synA")["cases"]][:6] + [("synB", c) for c in G.cases("synB")["cases"]][:3])
def test_threaded_text_readers_give_the_same_plan_and_messages(tag, case):
    """Pileup and legend files above 1 MiB are parsed by a team of threads (byte ranges cut at line starts,
    tables and stderr text joined in file order).  IBDGEM_MT_MIN_BYTES=1 sends the small golden inputs
    through that path: stdout and stderr must equal the line-by-line run's, for any team size."""
    meta = G.cases(tag)
    args = meta["base_args"] + meta["cases"][case] + ["--plan"]
    cwd = os.path.join(G.GOLD, tag, "input")
    want = subprocess.run([_exe()] + args + ["--threads", "1"], cwd=cwd, capture_output=True, text=True)
    assert want.returncode == 0
    strip = lambda e: "\n".join(l for l in e.splitlines() if not l.startswith("Run time:"))
    for th in ("2", "3", "7"):
        got = subprocess.run([_exe()] + args + ["--threads", th], cwd=cwd, capture_output=True, text=True,
                             env=dict(os.environ, IBDGEM_MT_MIN_BYTES="1"))
        assert got.returncode == 0, got.stderr
        assert got.stdout == want.stdout
        assert strip(got.stderr) == strip(want.stderr)


def test_threaded_pileup_reader_on_messy_lines(tmp_path):
    """Unparsable starts, bad read fields, over-deep lines, a last line without newline and several
    chromosomes: the threaded reader reports and keeps what the line-by-line reader does."""
    import random
    rnd = random.Random(5)
    lines, pos = [], 100
    for i in range(4000):
        pos += rnd.randint(1, 9)
        kind = rnd.random()
        chrom = "chr1" if i < 2500 else "chr2" if i < 3300 else "chr1"
        if kind < 0.02:
            lines.append("garbage")
        elif kind < 0.04:
            lines.append(f"{chrom}\t{pos}\tA\t3\tAC?\tIII\tIII")
        elif kind < 0.06:
            lines.append(f"{chrom}\t{pos}\tA\t2\tACG\tII\tII")
        elif kind < 0.08:
            lines.append(f"{chrom}\t{pos}\tA\t200\t{'A' * 200}\t{'I' * 200}\t{'I' * 200}")
        else:
            c = rnd.randint(0, 6)
            bases = "".join(rnd.choice("ACGTacgt.,") for _ in range(c)) or "*"
            lines.append(f"{chrom}\t{pos}\tN\t{c}\t{bases}\t{'I' * max(c, 1)}\t{'I' * max(c, 1)}")
    pu = tmp_path / "m.pileup"
    pu.write_text("\n".join(lines))                       # no newline at the end
    base = ["-H", os.path.join(FIX_IN, "test.hap"), "-L", os.path.join(FIX_IN, "test.legend"), "-I",
            os.path.join(FIX_IN, "test.indv"), "-P", str(pu), "--plan"]
    want = subprocess.run([_exe()] + base + ["--threads", "1"], capture_output=True, text=True)
    strip = lambda e: "\n".join(l for l in e.splitlines() if not l.startswith("Run time:"))
    for th in ("2", "5", "16"):
        got = subprocess.run([_exe()] + base + ["--threads", th], capture_output=True, text=True,
                             env=dict(os.environ, IBDGEM_MT_MIN_BYTES="1"))
        assert got.returncode == want.returncode
        assert got.stdout == want.stdout and strip(got.stderr) == strip(want.stderr)
    assert "Problem parsing garbage" in want.stderr and "Cannot parse ? in reads field" in want.stderr


def test_host_sources_keep_clear_of_libc_calls_with_hidden_state():
    """The device contexts start on a thread of their own while the main thread parses its inputs, and the GPU runtime's
    start-up uses libc freely: strtok in the -s parser shared its hidden state with the runtime's strtok and now and then
    cut the list of comparison individuals short ("Sample 00000000 not found in reference panel.", exit code 0).  The
    program's sources use the re-entrant forms only (rand(): the program has its own copy of glibc's generator)."""
    import re
    host = os.path.join(REPO, "ibdgem_amd", "host")
    for fn in sorted(os.listdir(host)):
        if fn.endswith(".c"):
            text = open(os.path.join(host, fn)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            for call in ("strtok", "localtime", "gmtime", "asctime", "ctime", "strerror", "rand", "srand", "getpwnam", "readdir"):
                assert not re.search(r"(?<![A-Za-z0-9_])%s\s*\(" % call, text), f"{fn}: {call}()"


def test_number_conversions_equal_printf():
    """The per-site table is written with the program's own %e / %lf conversions (long double / 128-bit integer
    arithmetic, sprintf for whatever falls within 1e-6 of a rounding boundary): against sprintf on two million
    random doubles -- raw bit patterns, [0,1), likelihood-like magnitudes down to 1e-322, short decimals, %lf ties
    (k/2^7), allele frequencies -- there must be no difference."""
    r = subprocess.run([_exe(), "--fmt-check", "2000000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-500:]
    assert "2000000 values, 0 differences" in r.stdout
