"""From a rocprofv3 kernel trace of bench.py --timed-only: for the last 20 launches of the dominant --LD kernel, its own
duration, the gap to the previous launch of it (end -> start), and which other kernels ran in between / beside it.
python tools/step_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
name_key = "Kernel_Name" if "Kernel_Name" in rows[0] else "Kernel Name"
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[name_key].split("(")[0]) for r in rows))
dom = [i for i, e in enumerate(ev) if "k_ld_popcount<" in e[2]]
print(f"{len(ev)} dispatches, {len(dom)} of the dominant kernel")
last = dom[-21:]
for a, b in zip(last[:-1], last[1:]):
    s0, e0, _ = ev[a]
    s1, e1, _ = ev[b]
    between = [(e[2].replace("ibdg::", "")[:28], (e[0] - e0) / 1e3, (e[1] - e[0]) / 1e3) for e in ev[a + 1:b]]
    print(f"kernel {(e1 - s1) / 1e3:8.1f} us   gap before it {(s1 - e0) / 1e3:6.1f} us   start-to-start {(s1 - s0) / 1e3:8.1f}   "
          + "  ".join(f"{n} @+{t:.1f} for {d:.1f}" for n, t, d in between))
