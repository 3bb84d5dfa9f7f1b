"""Readers for the committed golden fixtures (tests/golden/, see make_golden.py).

The tab/summary files are the reference's own output formats
(reference src/ibdgem.c:547-548, :731-733, :751-768).
"""
import gzip
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _open(path):
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path)


def read_lines(path):
    with _open(path) as fh:
        return fh.read().splitlines()


class Panel:
    """IMPUTE .hap/.legend/.indv as arrays (test-side reader, no filtering)."""

    def __init__(self, hap, legend, indv):
        self.names = [l for l in read_lines(indv) if l != ""]
        leg = read_lines(legend)[1:]
        self.legend = [l.split() for l in leg]
        self.pos = np.array([int(x[1]) for x in self.legend], dtype=np.int64)
        rows = read_lines(hap)
        self.alleles = np.array([np.frombuffer(r.encode(), dtype=np.uint8)[::2] - ord("0") for r in rows],
                                dtype=np.uint8)
        assert self.alleles.shape == (len(self.pos), 2 * len(self.names))
        self.row_of_pos = {int(p): i for i, p in enumerate(self.pos)}

    def index(self, name):
        return self.names.index(name)


def load_syn_panel(tag):
    inp = os.path.join(GOLD, tag, "input")
    return Panel(os.path.join(inp, "panel.hap.gz"), os.path.join(inp, "panel.legend.gz"),
                 os.path.join(inp, "panel.indv"))


def load_fixture_panel():
    inp = os.path.join(GOLD, "ibdgem-test", "input")
    return Panel(os.path.join(inp, "test.hap"), os.path.join(inp, "test.legend"), os.path.join(inp, "test.indv"))


def cases(tag):
    with open(os.path.join(GOLD, tag, "cases.json")) as fh:
        return json.load(fh)


class TabFile:
    """Parsed *.tab.txt: data rows + header/footer statistics."""

    def __init__(self, path):
        self.rows = []
        self.comments = []
        for line in read_lines(path):
            if line.startswith("#"):
                self.comments.append(line)
                continue
            if line == "":
                continue
            c = line.split("\t")
            self.rows.append(c)
        r = self.rows
        self.chr = [x[0] for x in r]
        self.rsid = [x[1] for x in r]
        self.pos = np.array([int(x[2]) for x in r], dtype=np.int64)
        self.ref = [x[3] for x in r]
        self.alt = [x[4] for x in r]
        self.af_txt = [x[5] for x in r]
        self.dp = np.array([int(x[6]) for x in r], dtype=np.int64)
        self.n_ref = np.array([int(x[7]) for x in r], dtype=np.uint8)
        self.n_alt = np.array([int(x[8]) for x in r], dtype=np.uint8)
        self.a0 = np.array([int(x[9]) for x in r], dtype=np.uint8)
        self.a1 = np.array([int(x[10]) for x in r], dtype=np.uint8)
        self.ll = np.array([[float(x[11]), float(x[12]), float(x[13])] for x in r], dtype=np.float64).reshape(-1, 3)
        self.ll_txt = [x[11:14] for x in r]
        self.processed = self.skipped = None
        for c in self.comments:
            if c.startswith("## Number of sites processed:"):
                self.processed = int(c.split(":")[1])
            if c.startswith("## Number of sites skipped:"):
                self.skipped = int(c.split(":")[1])


class SummaryFile:
    def __init__(self, path):
        rows = [l.split("\t") for l in read_lines(path) if l and not l.startswith("#")]
        self.segment = np.array([int(x[0]) for x in rows], dtype=np.int64)
        self.start = np.array([int(x[1]) for x in rows], dtype=np.int64)
        self.end = np.array([int(x[2]) for x in rows], dtype=np.int64)
        self.ll = np.array([[float(x[3]), float(x[4]), float(x[5])] for x in rows], dtype=np.float64).reshape(-1, 3)
        self.ll_txt = [x[3:6] for x in rows]
        self.nsites = np.array([int(x[6]) for x in rows], dtype=np.int64)


def syn_outputs(tag, case, sq, target):
    d = os.path.join(GOLD, tag, case)
    return (TabFile(os.path.join(d, f"{sq}.{target}.tab.txt.gz")),
            SummaryFile(os.path.join(d, f"{sq}.{target}.summary.txt.gz")))


def fixture_outputs(sq, target):
    d = os.path.join(GOLD, "ibdgem-test", "output")
    return (TabFile(os.path.join(d, f"{sq}.{target}.tab.txt")),
            SummaryFile(os.path.join(d, f"{sq}.{target}.summary.txt")))


def parse_flags(args):
    """The subset of reference CLI flags the golden cases use -> dict."""
    out = dict(ld=False, window=100, eps=0.02, max_cov=20, sq="UNKWN", targets=None, targets_file=None,
               bg_file=None, af_file=None, varsites=False)
    it = iter(args)
    for a in it:
        if a == "--LD":
            out["ld"] = True
        elif a == "-v":
            out["varsites"] = True
        elif a == "-w":
            out["window"] = int(next(it))
        elif a == "-e":
            out["eps"] = float(next(it))
        elif a == "-M":
            out["max_cov"] = int(next(it))
        elif a == "-N":
            out["sq"] = next(it)
        elif a == "-s":
            out["targets"] = next(it).split(",")
        elif a == "-S":
            out["targets_file"] = next(it)
        elif a == "-B":
            out["bg_file"] = next(it)
        elif a == "-A":
            out["af_file"] = next(it)
        elif a in ("-F", "-f", "-c", "-D", "-p", "-H", "-L", "-I", "-P", "-O"):
            next(it)
    return out


def case_setup(tag, case):
    """Everything a parity test needs for one synthetic golden case.

    Returns (flags, panel, targets, refids, pu_id, per-target dict of
    (tab, summary, alleles, f_override))."""
    meta = cases(tag)
    flags = parse_flags(meta["cases"][case])
    panel = load_syn_panel(tag)
    inp = os.path.join(GOLD, tag, "input")
    if flags["targets_file"]:
        names = [l for l in read_lines(os.path.join(inp, flags["targets_file"])) if l in panel.names]
    else:
        names = [n for n in flags["targets"] if n in panel.names]
    refids = None
    if flags["bg_file"]:
        refids = [panel.index(l) for l in read_lines(os.path.join(inp, flags["bg_file"])) if l in panel.names]
    pu_id = panel.index(flags["sq"]) if flags["sq"] in panel.names else -1
    af = None
    if flags["af_file"]:
        af = {}
        for l in read_lines(os.path.join(inp, flags["af_file"])):
            c = l.split()
            af[int(c[1])] = float(c[2])
    per_target = {}
    for name in names:
        tab, summ = syn_outputs(tag, case, flags["sq"], name)
        rows = np.array([panel.row_of_pos[int(p)] for p in tab.pos], dtype=np.int64)
        alle = panel.alleles[rows] if len(rows) else np.zeros((0, panel.alleles.shape[1]), np.uint8)
        fo = None
        if af is not None:
            fo = np.array([af.get(int(p), np.nan) for p in tab.pos], dtype=np.float64)
        per_target[name] = (tab, summ, alle, fo, rows)
    return flags, panel, names, refids, pu_id, per_target
