#!/usr/bin/env python3
"""Full-size golden of the headline workload through the UNMODIFIED program (VERDICT r03 #5b): the 150^3-node linear-elastic cube of
bench.py as a `!SOLUTION, TYPE=STATIC` deck (scripts/fistr1_cube_deck.py N --linear, ITERLOG switched on), run by
oracle/_ref/fistr1_ref (the reference's own main program, OpenMP build: the multicolour SSOR the GPU path reproduces).
Stored: iteration count, the first 50 ITERLOG lines ('(i7,1pe16.6)', hecmw_solver_CG.f90:245), the last line, and the
displacement / strain / stress extrema of 0.log -> tests/golden/cube<N>_linear_fistr1_ref.json.
The deck itself is not stored (330 MB at N = 149): the test regenerates it with the same script.
usage: make_cube_fullsize_golden.py [N=149] [THREADS=8]     (N = 149: ~25 GB of host memory, ~10 minutes on 8 cores)"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fistr1_run as f1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = tempfile.mkdtemp(prefix="cube%d_" % n, dir=os.environ.get("TMPDIR", "/tmp"))
subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "fistr1_cube_deck.py"), d, str(n), "--linear"], check=True)
cnt = os.path.join(d, "cube.cnt")
s = open(cnt).read().replace("ITERLOG=NO", "ITERLOG=YES")
open(cnt, "w").write(s)
t0 = time.time()
r = f1.run("fistr1_ref", d, threads=threads, timeout=7200)
wall = time.time() - t0
assert r["returncode"] == 0 and "FrontISTR Completed !!" in r["stdout"], r["stdout"][-3000:]
hist = [float(m.group(2)) for m in (re.match(r"^\s*(\d+)\s+(\d\.\d{6}E[-+]\d\d)\s*$", l) for l in r["stdout"].split("\n")) if m]
m = re.search(r"^\s*(\d+) iterations\s+([0-9.E+-]+)", r["stdout"], re.M)
out = {"n_elem_edge": n, "ndof": 3 * (n + 1) ** 3, "iterations": int(m.group(1)), "final_resid": float(m.group(2)),
       "history_head": hist[:50], "history_last": hist[-1], "n_history_lines": len(hist),
       "log_last_step": r["log"][-1], "threads": threads, "wall_s": round(wall, 1),
       "program": "oracle/_ref/fistr1_ref (unmodified reference, flang -fopenmp)", "deck": "scripts/fistr1_cube_deck.py DIR %d --linear, ITERLOG=YES" % n}
path = os.path.join(ROOT, "tests", "golden", "cube%d_linear_fistr1_ref.json" % n)
with open(path, "w") as fh:
    json.dump(out, fh, indent=1)
print("wrote", path, "iterations", out["iterations"], "wall %.1f s" % wall)
subprocess.run(["rm", "-rf", d])
