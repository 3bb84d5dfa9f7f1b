#!/usr/bin/env python3
"""Regenerate tests/golden/ (run in the build container only).

What this script commits under tests/golden/ is DATA:
  ibdgem-test/{input,output}/   the reference's own example inputs and its 18
                                expected output files (copied verbatim from
                                /root/reference/supplementary/ibdgem-test).
  syn*/input/                   small seeded synthetic IMPUTE panels + pileups
                                (gzip text, reference file formats).
  syn*/<case>/ref7/             the UNMODIFIED reference's output files for the same runs
                                (oracle/_ref/ibdgem, "%e" = 7 digits): byte-for-byte targets
                                for the host program.
  syn*/<case>/                  what the REFERENCE ITSELF printed for them: the
                                reference sources compiled where they lie by
                                oracle/Makefile into oracle/_ref/ibdgem_p17
                                (the "%e" conversions widened to "%.17e" so the
                                doubles round-trip) and run with the flags
                                recorded in cases.json.  Output files are stored
                                gzip-compressed with line 1 (the echoed command,
                                which contains container paths) dropped.
  math_grid.tsv.gz              find_pDgG/find_pDgf/find_pDgIBD1 of the
                                reference (oracle/_ref/libibdmath_ref.so) over
                                the full (n_ref,n_alt) grid at several eps, as
                                C99 hex floats.

Usage:  python tests/golden/make_golden.py        (needs /root/reference)
"""
import ctypes
import gzip
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
REFBIN = os.path.join(REPO, "oracle", "_ref", "ibdgem_p17")
REFBIN7 = os.path.join(REPO, "oracle", "_ref", "ibdgem")      # unmodified: "%e" outputs, byte-for-byte targets
REFMATH = os.path.join(REPO, "oracle", "_ref", "libibdmath_ref.so")
BASES = "ACGT"


def gz_write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with gzip.GzipFile(path, "wb", mtime=0) as fh:   # mtime=0: reproducible bytes
        fh.write(text.encode())


def synth(seed, n_ids, n_sites, truth_id, chrom="7"):
    """Seeded synthetic panel + pileup in the reference's text formats."""
    rng = np.random.default_rng(seed)
    gaps = rng.integers(1, 120, size=n_sites)
    pos = 1000 + np.cumsum(gaps)
    f = np.clip(rng.beta(0.3, 1.0, size=n_sites), 1e-3, 0.999)
    hap = (rng.random((n_sites, 2 * n_ids)) < f[:, None]).astype(np.uint8)
    ref = rng.integers(0, 4, size=n_sites)
    alt = (ref + rng.integers(1, 4, size=n_sites)) % 4
    legend = ["ID pos allele0 allele1"]
    kinds = rng.random(n_sites)
    for i in range(n_sites):
        r, a = BASES[ref[i]], BASES[alt[i]]
        if kinds[i] < 0.02:
            a = a + "T"            # indel row: fails is_snp
        elif kinds[i] < 0.03:
            r = "-"                # not in ACGT
        legend.append(f"rs{i} {pos[i]} {r} {a}")
    hap_lines = [" ".join(map(str, row)) for row in hap]
    indv = [f"ind{n}" for n in range(n_ids)]

    pile = []
    cov = rng.poisson(2.0, size=n_sites)
    burst = rng.random(n_sites)
    cov = np.where(burst < 0.01, 25, cov)       # n_ref+n_alt > max-cov -> skipped
    cov = np.where((burst >= 0.01) & (burst < 0.02), 130, cov)   # cov >= 128 -> line dropped
    has_line = rng.random(n_sites) > 0.05       # ~5% of SNPs have no pileup line
    err = 0.02
    extra_at = set(rng.choice(n_sites, size=n_sites // 10, replace=False).tolist())
    for i in range(n_sites):
        if i in extra_at:                       # a pileup line at a non-SNP position
            pile.append(f"{chrom}\t{pos[i] - 1 if gaps[i] > 1 else pos[i]}\tN\t1\tA\tI\t]")
            if gaps[i] <= 1:
                continue
        if not has_line[i]:
            continue
        c = int(cov[i])
        if c == 0:
            pile.append(f"{chrom}\t{pos[i]}\tN\t0\t*\t*\t*")
            continue
        g0, g1 = hap[i, 2 * truth_id], hap[i, 2 * truth_id + 1]
        seq = []
        for _ in range(c):
            allele = g0 if rng.random() < 0.5 else g1
            b = alt[i] if allele else ref[i]
            u = rng.random()
            if u < err:
                b = (b + rng.integers(1, 4)) % 4
            ch = BASES[b]
            if rng.random() < 0.5:
                ch = ch.lower()
            v = rng.random()
            if v < 0.03:
                ch = "^]" + ch              # read start marker + mapq char
            elif v < 0.06:
                ch = ch + "$"
            elif v < 0.08:
                ch = ch + "+2AC"
            elif v < 0.10:
                ch = ch + "-1g"
            elif v < 0.11:
                ch = "."                    # pileup REF column is N: matches nothing
            elif v < 0.12:
                ch = "*"
            seq.append(ch)
        q = "I" * c
        pile.append(f"{chrom}\t{pos[i]}\tN\t{c}\t{''.join(seq)}\t{q}\t{q}")
    return dict(pos=pos, hap=hap, f=f, legend="\n".join(legend) + "\n",
                hap_txt="\n".join(hap_lines) + "\n", indv="\n".join(indv) + "\n",
                pileup="\n".join(pile) + "\n")


def run_case(workdir, indir, name, args, outroot):
    for exe, sub in ((REFBIN, ""), (REFBIN7, "ref7")):
        out = os.path.join(workdir, "out_" + name + sub)
        os.makedirs(out)
        cmd = [exe] + args + ["-O", out]
        res = subprocess.run(cmd, cwd=indir, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        if res.returncode != 0:
            raise SystemExit(f"reference failed for {name}: {res.stderr}")
        for fn in sorted(os.listdir(out)):
            with open(os.path.join(out, fn)) as fh:
                lines = fh.readlines()
            if fn.endswith(".tab.txt"):
                lines = lines[1:]               # drop '# Entered command:' (container paths)
            gz_write(os.path.join(outroot, name, sub, fn + ".gz"), "".join(lines))


def build_syn(tag, seed, n_ids, n_sites, truth_id, cases, extra_files):
    d = synth(seed, n_ids, n_sites, truth_id)
    root = os.path.join(HERE, tag)
    shutil.rmtree(root, ignore_errors=True)
    inp = os.path.join(root, "input")
    gz_write(os.path.join(inp, "panel.hap.gz"), d["hap_txt"])
    gz_write(os.path.join(inp, "panel.legend.gz"), d["legend"])
    gz_write(os.path.join(inp, "reads.pileup.gz"), d["pileup"])
    os.makedirs(inp, exist_ok=True)
    with open(os.path.join(inp, "panel.indv"), "w") as fh:
        fh.write(d["indv"])
    for fn, text in extra_files(d).items():
        with open(os.path.join(inp, fn), "w") as fh:
            fh.write(text)
    base = ["-H", "panel.hap.gz", "-L", "panel.legend.gz", "-I", "panel.indv", "-P", "reads.pileup.gz"]
    with tempfile.TemporaryDirectory() as wd:
        for name, extra in cases.items():
            run_case(wd, inp, name, base + extra, root)
    with open(os.path.join(root, "cases.json"), "w") as fh:
        json.dump({"seed": seed, "n_ids": n_ids, "n_sites": n_sites, "truth_id": truth_id,
                   "base_args": base, "cases": cases}, fh, indent=1)
        fh.write("\n")


def build_vcf_case():
    """synA's panel as a VCF (plus rows the VCF path must skip) through the reference's -V path.
    One comparison individual per run: the reference frees its GT regex inside the per-target loop
    (src/ibdgem.c:472) and crashes on the second target."""
    root = os.path.join(HERE, "synV")
    shutil.rmtree(root, ignore_errors=True)
    inp = os.path.join(root, "input")
    os.makedirs(inp)
    src = os.path.join(HERE, "synA", "input")
    with gzip.open(os.path.join(src, "panel.hap.gz"), "rt") as fh:
        hap = [l.split() for l in fh.read().splitlines()]
    with gzip.open(os.path.join(src, "panel.legend.gz"), "rt") as fh:
        leg = [l.split() for l in fh.read().splitlines()[1:]]
    names = open(os.path.join(src, "panel.indv")).read().split()
    rng = np.random.default_rng(7)
    out = ["##fileformat=VCFv4.2", "##source=make_golden.py",
           "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names)]
    for (rid, pos, ref, alt), row in zip(leg, hap):
        u = rng.random()
        qual = "50" if u > 0.1 else "12.5"
        gts = [f"{row[2 * i]}|{row[2 * i + 1]}" for i in range(len(names))]
        if u > 0.97:
            alt = alt + ",G"                       # multi-allelic: skipped
        elif u > 0.95:
            gts[5] = "./."                         # unparsable genotype: row skipped with a message
        elif u > 0.93:
            gts[2] = gts[2].replace("|", "/") + ":35"   # unphased separator and extra sub-fields are fine
        out.append("\t".join(["7", pos, rid, ref, alt, qual, "PASS", "AC=1", "GT"] + gts))
    gz_write(os.path.join(inp, "panel.vcf.gz"), "\n".join(out) + "\n")
    shutil.copy(os.path.join(src, "reads.pileup.gz"), os.path.join(inp, "reads.pileup.gz"))
    base = ["-V", "panel.vcf.gz", "-P", "reads.pileup.gz"]
    cases = {"vcf_ld": ["--LD", "-s", "ind3"], "vcf_nonld_q30": ["-s", "ind64", "-q", "30"],
             "vcf_ld_varsites_w50": ["--LD", "-v", "-w", "50", "-s", "ind9", "-N", "ind5"]}
    with tempfile.TemporaryDirectory() as wd:
        for name, extra in cases.items():
            run_case(wd, inp, name, base + extra, root)
    with open(os.path.join(root, "cases.json"), "w") as fh:
        json.dump({"base_args": base, "cases": cases}, fh, indent=1)
        fh.write("\n")


def math_grid():
    lib = ctypes.CDLL(REFMATH)
    lib.init_nCk.restype = ctypes.c_void_p
    lib.init_nCk.argtypes = [ctypes.c_uint]
    lib.find_pDgG.restype = ctypes.c_double
    lib.find_pDgG.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_ushort, ctypes.c_ushort,
                              ctypes.c_uint, ctypes.c_uint]
    lib.find_pDgf.restype = ctypes.c_double
    lib.find_pDgf.argtypes = [ctypes.c_double] * 4
    lib.find_pDgIBD1.restype = ctypes.c_double
    lib.find_pDgIBD1.argtypes = [ctypes.c_ushort, ctypes.c_ushort] + [ctypes.c_double] * 4
    rows = ["# eps\tmax_cov\tn_ref\tn_alt\tf\tp00\tp01\tp11\tibd0\tibd1_g0\tibd1_g1\tibd1_g2"]
    fs = [0.0, 1.0, 0.5, 1.0 / 3.0, 0.001, 0.999, 37.0 / 5008.0, 4999.0 / 5008.0, 0.123456789]
    for eps, M in [(0.02, 20), (0.05, 6), (1e-3, 40), (1e-30, 20)]:
        nck = lib.init_nCk(M)
        for r in range(M + 1):
            for a in range(M + 1 - r):
                p = [lib.find_pDgG(nck, eps, x, y, r, a) for x, y in ((0, 0), (0, 1), (1, 1))]
                for f in fs:
                    v = [lib.find_pDgf(f, *p)] + [lib.find_pDgIBD1(x, y, f, *p)
                                                 for x, y in ((0, 0), (1, 0), (1, 1))]
                    rows.append("\t".join([float(eps).hex(), str(M), str(r), str(a), float(f).hex()] +
                                          [float(x).hex() for x in p + v]))
    gz_write(os.path.join(HERE, "math_grid.tsv.gz"), "\n".join(rows) + "\n")


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (build container only)")
    subprocess.run(["make", "-C", os.path.join(REPO, "oracle"), "all", "ref"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    # 1. the reference's own fixture data
    dst = os.path.join(HERE, "ibdgem-test")
    shutil.rmtree(dst, ignore_errors=True)
    shutil.copytree(os.path.join(REF, "supplementary", "ibdgem-test"), dst)
    # 2. reference math over the grid
    math_grid()
    # 3. synthetic cases through the reference binary
    def extras_a(d):
        pos = d["pos"]
        rng = np.random.default_rng(99)
        af_rows = [f"7\t{p}\t{rng.random():.6f}" for p in pos[::3]]
        return {
            "bg20.txt": "".join(f"ind{n}\n" for n in range(0, 40, 2)),
            "bg_dup.txt": "ind10\nind11\nind10\nind12\nnosuch\n",
            "bg_self.txt": "ind3\n",
            "targets.txt": "ind3\nind64\nind69\n",
            "af.txt": "\n".join(af_rows) + "\n",
            "pos.txt": "".join(f"7\t{p}\n" for p in pos[::2]),
        }
    build_syn("synA", 20241008, 70, 1200, 3, {
        "ld_default": ["--LD", "-s", "ind3,ind64"],
        "ld_pu_in_panel": ["--LD", "-N", "ind5", "-s", "ind5,ind9"],
        "ld_bg20_w64": ["--LD", "-B", "bg20.txt", "-s", "ind3,ind4", "-w", "64"],
        "ld_bg_dup": ["--LD", "-B", "bg_dup.txt", "-s", "ind3,ind10"],
        "ld_bg_self_nan": ["--LD", "-B", "bg_self.txt", "-s", "ind3"],
        "ld_varsites": ["--LD", "-v", "-S", "targets.txt"],
        "ld_downsample": ["--LD", "-D", "1.0", "-s", "ind3,ind4"],
        "ld_af_file": ["--LD", "-A", "af.txt", "-s", "ind3"],
        "ld_positions": ["--LD", "-p", "pos.txt", "-s", "ind3", "-w", "30"],
        "nonld_flags": ["-s", "ind3", "-e", "0.05", "-M", "6", "-F", "0.9", "-f", "0.05", "-c", "7"],
        "nonld_all_targets_w2": ["-w", "2", "-s", "ind0,ind69"],
    }, extras_a)
    build_syn("synB", 20241009, 130, 500, 129, {
        "ld_w37": ["--LD", "-w", "37", "-s", "ind129,ind0,ind63,ind64"],
        "ld_pu_named": ["--LD", "-N", "ind129", "-s", "ind129,ind1"],
    }, lambda d: {})
    build_vcf_case()
    print("golden vectors written under", HERE)


if __name__ == "__main__":
    sys.exit(main())
