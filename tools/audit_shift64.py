"""Audit of the kernels' ISA for the 64-bit-shift hazard found in round 4 (docs/DESIGN_rounds_1-4.md s4.4, tools/ubench/shift64_top_vgpr.hip):
on gfx950 a v_lshlrev_b64 / v_lshrrev_b64 / v_ashrrev_i64 whose shift AMOUNT sits in the last VGPR of the wave's
allocation reads it wrongly in ~7 % of executions (the amount comes from v0 instead) -- LLVM knows this as the
"Shift64HighRegBug" of gfx11 and works around it there; hipcc does not for gfx950.  Like LLVM's workaround this audit is
conservative: it flags every such shift whose amount register has index 7 mod 8 (the allocation granule is 8).

    python tools/audit_shift64.py [file.s ...]        (default: ibdgem_amd/csrc/*.s, made by `make -C ibdgem_amd/csrc isa`)
Exit code 1 if anything is flagged."""
import glob, os, re, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def audit(path):
    flagged, kernel, n_shifts = [], None, 0
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z[\w$.]+):\s*$", line)
        if m:
            kernel = m.group(1)
            continue
        m = re.match(r"\s*(v_lshlrev_b64|v_lshrrev_b64|v_ashrrev_i64)(?:_e64)?\s+(\S+),\s*(\S+),\s*(\S+)", line)
        if not m:
            continue
        n_shifts += 1
        amount = m.group(3)
        r = re.fullmatch(r"v(\d+)", amount)
        if r and int(r.group(1)) % 8 == 7:
            flagged.append((path, ln, kernel, line.strip()))
    return flagged, n_shifts


def main(paths):
    if not paths:
        paths = sorted(glob.glob(os.path.join(REPO, "ibdgem_amd", "csrc", "*.s")))
    if not paths:
        print("no .s files: run `make -C ibdgem_amd/csrc isa` first")
        return 2
    bad, total = [], 0
    for p in paths:
        f, n = audit(p)
        bad += f
        total += n
    for path, ln, kernel, text in bad:
        print(f"{os.path.basename(path)}:{ln}: {kernel}: {text}")
    print(f"audit_shift64: {total} 64-bit vector shifts in {len(paths)} files, {len(bad)} with the amount in a VGPR of index 7 mod 8")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
