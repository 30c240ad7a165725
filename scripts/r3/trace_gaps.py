#!/usr/bin/env python3
"""Timeline of the last full Krylov iterations in a rocprofv3 kernel trace: kernel, duration, gap to the previous kernel's end."""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "")
names = [short(r[2]) for r in rows]
# an iteration starts at k_scalar<1> (OP_CG_RHO) or k_scalar<...>; find the timed region = the longest run without set-up kernels
marks = [i for i, n in enumerate(names) if n.startswith("k_scalar<1>") or n.startswith("k_scalar<5>")]
if len(marks) < 6:
    print("no iterations found"); sys.exit(0)
# take iterations 6..9 from the end of the krylov loop (before the roofline timing loops that follow)
# the krylov loop = consecutive marks with similar spacing; use marks[4:8]
sel = marks[5:8]
for a, b in zip(sel[:-1], sel[1:]):
    tot = rows[b][0] - rows[a][0]
    busy = sum(rows[i][1] - rows[i][0] for i in range(a, b))
    print("---- iteration: %d launches, wall %.1f us, kernel time %.1f us, gaps %.1f us" % (b - a, tot / 1e3, busy / 1e3, (tot - busy) / 1e3))
    for i in range(a, b):
        gap = rows[i][0] - rows[i - 1][1]
        print("%8.1f us  gap %6.1f  %s" % ((rows[i][1] - rows[i][0]) / 1e3, gap / 1e3, names[i]))
# aggregate over iterations marks[3]..marks[-1] region until the loop ends
