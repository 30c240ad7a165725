"""fstr_UpdateNewton of a linear static deck on the device (fx_update_c3d8_linear) on the n^3-element cube: kernel ms and wall time of the
whole call (uploads, kernel, 2 x 48 doubles per element back through the pinned staging) for ELEMOPT361 = IC / B-bar / FI.
Usage: python scripts/r4/bench_update_linear.py [n]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np                              # noqa: E402
from frontistr_amd import hecmw as hip          # noqa: E402
from frontistr_amd.mesh import CubeMesh          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
mesh = CubeMesh(n)
u = 1e-3 * np.sin(0.37 * np.arange(3 * mesh.n_node) + 0.1)
ctx = hip.SolverContext()
out = {"n_elem": int(mesh.conn.shape[0]), "dof": 3 * mesh.n_node,
       "bytes_out": 2 * 48 * 8 * int(mesh.conn.shape[0]) + 24 * mesh.n_node, "note": "first call of a process also pins the 2.5 GB host staging"}
for eo, name in ((1, "ic"), (2, "bbar"), (3, "fi")):
    ms, wall = [], []
    for _ in range(3):
        t0 = time.perf_counter()
        s, t, q, k = ctx.update_c3d8_linear(mesh.coord, mesh.conn, 210000.0, 0.3, u, elemopt=eo)
        wall.append(round(time.perf_counter() - t0, 3))
        ms.append(round(k, 3))
    out[name + "_kernel_ms"] = ms
    out[name + "_call_s"] = wall
    out[name + "_kernel_GBs"] = round(out["bytes_out"] / (min(ms) * 1e-3) / 1e9, 1)
print(json.dumps(out))
