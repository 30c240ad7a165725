"""tests/golden/exI_A361_expect.npz: the reference's own known answers for its geometrically nonlinear example
examples/static/exI (A361.msh == exA's mesh, I300.cnt: !STATIC TYPE=NLGEOM, elastic E=4000 nu=0.3, CLOAD -1.0,
10 substeps): displacement extrema per step from exI/A361_correct.log (data of the reference's test harness,
compared at 1e-4 absolute by examples/test_FrontISTR.rb).  Run in the build container only."""
import os
import re

import numpy as np

LOG = "/root/reference/examples/static/exI/A361_correct.log"
steps, cur, in_global = {}, None, False
for line in open(LOG):
    m = re.match(r"#### Result step=\s*(\d+)", line)
    if m:
        cur = int(m.group(1)); steps[cur] = {}; in_global = False
        continue
    if "Global Summary" in line:
        in_global = True
        continue
    m = re.match(r"\s*//(U[123])\s+([-0-9.E+]+)\s+([-0-9.E+]+)\s*$", line)
    if m and in_global and cur is not None:
        steps[cur][m.group(1)] = (float(m.group(2)), float(m.group(3)))
n = max(steps)
out = np.zeros((n, 3, 2))
for s in range(1, n + 1):
    for c, k in enumerate(("U1", "U2", "U3")):
        out[s - 1, c] = steps[s][k]
print(out[:, 2])
np.savez(os.path.join(os.path.dirname(os.path.abspath(__file__)), "exI_A361_expect.npz"), extrema=out,
         substeps=n, converg=1.0e-3, max_iter=50, E=4000.0, nu=0.3)
