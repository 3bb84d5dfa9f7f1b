"""Golden outputs of the reference's second program (src/hiddengem.c), produced by running the
UNMODIFIED reference binary built by `make -C oracle ref` (oracle/_ref/hiddengem) on summary files
that are already committed as data: the reference's own 9 fixture summaries, the reference-written
--LD summaries of the synthetic cases, and a few hand-made tables (NaN rows, a single window,
comment and malformed lines in the middle, custom penalties).  Only data is written:
tests/golden/hidden/cases.json + one <name>.out (stdout of the reference) and, for hand-made
inputs, <name>.summary.txt.

    python tests/golden/make_golden_hidden.py
"""
import glob
import gzip
import json
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(REPO, "oracle", "_ref", "hiddengem")
OUT = os.path.join(HERE, "hidden")

HAND = {
    "one_window": "# SEGMENT\tSTART\tEND\tLIBD0\tLIBD1\tLIBD2\tNUM_SITES\n1\t10\t90\t1e-30\t3e-29\t2e-40\t100\n",
    "nan_rows": ("# SEGMENT\tSTART\tEND\tLIBD0\tLIBD1\tLIBD2\tNUM_SITES\n"
                 "1\t10\t90\t-nan\t-nan\t1e-12\t100\n2\t100\t190\t1e-20\t1e-22\t1e-30\t100\n"
                 "3\t200\t290\t0\t0\t0\t100\n4\t300\t390\t1e-25\t1e-21\t1e-26\t57\n"),
    "noise_lines": ("# a comment\n# SEGMENT\tSTART\tEND\tLIBD0\tLIBD1\tLIBD2\tNUM_SITES\n"
                    "1\t10\t90\t1e-20\t1e-22\t1e-30\t100\n# a comment in the middle\nnot a row\n\n"
                    "2\t100\t190\t1e-28\t1e-22\t1e-21\t100\n3\t200\t290\t1e-28\t1e-22\t1e-21\t100\n"
                    "4\t300\t390\t2e-25\t1e-25\t1e-25\t100\n5 400 490 1e-3 1e-9 1e-9 12\n"),
    "switching": "# SEGMENT\tSTART\tEND\tLIBD0\tLIBD1\tLIBD2\tNUM_SITES\n" + "".join(
        f"{i + 1}\t{100 * i}\t{100 * i + 99}\t{l[0]:e}\t{l[1]:e}\t{l[2]:e}\t100\n" for i, l in enumerate(
            [(1e-20, 1e-26, 1e-40)] * 6 + [(1e-30, 1e-21, 1e-27)] * 7 + [(1e-45, 1e-30, 1e-22)] * 5 +
            [(1e-21, 1e-22, 1e-23)] * 3 + [(3e-300, 2e-300, 1e-300)] * 4)),
}


def main():
    assert os.path.exists(REF), "run `make -C oracle ref` first"
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    cases = []
    inputs = [(os.path.relpath(p, HERE), []) for p in sorted(glob.glob(os.path.join(HERE, "ibdgem-test", "output", "*.summary.txt")))]
    for tag in ("synA", "synB"):
        for p in sorted(glob.glob(os.path.join(HERE, tag, "ld_*", "ref7", "*.summary.txt.gz"))):
            inputs.append((os.path.relpath(p, HERE), []))
    pen = [[], ["--p01", "0.5", "--p02", "0.25", "--p12", "0.9"], ["--p01", "1e-9"], ["--p12", "0", "--p02", "1e-2"]]
    with tempfile.TemporaryDirectory() as tmp:
        for name, text in HAND.items():
            with open(os.path.join(OUT, name + ".summary.txt"), "w") as fh:
                fh.write(text)
            for k, extra in enumerate(pen if name == "switching" else pen[:1]):
                inputs.append((os.path.join("hidden", name + ".summary.txt"), extra))
        # the largest reference-written table again with other penalties
        big = max((i for i in inputs if i[0].endswith(".gz")), key=lambda i: os.path.getsize(os.path.join(HERE, i[0])))
        inputs.append((big[0], pen[1]))
        for n, (rel, extra) in enumerate(inputs):
            src = os.path.join(HERE, rel)
            if src.endswith(".gz"):                   # golden tables are stored gzipped; the reference
                plain = os.path.join(tmp, f"in{n}.summary.txt")    # reads .gz too, but keep its input plain
                with gzip.open(src, "rb") as fi, open(plain, "wb") as fo:
                    fo.write(fi.read())
                src = plain
            res = subprocess.run([REF, "-s", src, *extra], capture_output=True, check=True)
            name = f"case{n:03d}"
            with open(os.path.join(OUT, name + ".out"), "wb") as fh:
                fh.write(res.stdout)
            cases.append(dict(name=name, input=rel, args=extra))
    with open(os.path.join(OUT, "cases.json"), "w") as fh:
        json.dump(cases, fh, indent=1)
    print(len(cases), "cases written to", OUT)


if __name__ == "__main__":
    main()
