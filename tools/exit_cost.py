"""Wall clock of a minimal HIP process against the time it spends inside main: what the END of a process that has touched
the GPU costs by itself (tools/ubench/exit_cost.hip).  python tools/exit_cost.py   (on a GPU box)"""
import os, subprocess, sys, time
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "exit_cost.hip")
exe = "/tmp/exit_cost"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", exe, src], check=True)
for args in (["1"], ["1", "x"], ["8000"], ["8000", "x"]):
    for rep in range(4):
        t0 = time.perf_counter()
        r = subprocess.run([exe] + args, capture_output=True, text=True)
        wall = time.perf_counter() - t0
        inside = float(r.stdout.split()[0])
        print(f"{' '.join(args):8s} {'_exit' if len(args) > 1 else 'return':6s}  wall {wall:.3f} s, inside main {inside:.3f} s, outside {wall - inside:.3f} s")
