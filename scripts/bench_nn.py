"""Generic block sizes at scale: SpMV rate and CG it/s of the NDOF != 3 path on a structured cube.
A = (graph Laplacian of the 27-point hex8 stencil + I) (x) I_NDOF plus an SPD perturbation of every diagonal block: SPD,
symmetric, the reference's D / AL / AU layout.  Usage: python scripts/bench_nn.py NDOF N_NODES_PER_EDGE [--steps K]"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frontistr_amd import hecmw as hip          # noqa: E402
from frontistr_amd.mesh import CubeMesh          # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("ndof", type=int)
ap.add_argument("n", type=int)
ap.add_argument("--steps", type=int, default=50)
a = ap.parse_args()
nd, nd2 = a.ndof, a.ndof * a.ndof
mesh = CubeMesh(a.n - 1)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
t0 = time.time()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
NP = m.NP
rng = np.random.default_rng(nd)
eye = np.eye(nd).ravel()
deg = (np.diff(m.indexL) + np.diff(m.indexU)).astype(np.float64)
G = rng.standard_normal((nd, nd))
pert = (G @ G.T / nd).ravel()
m.NDOF = nd
m.D = ((deg + 1.0)[:, None] * eye[None, :] + pert[None, :]).ravel()
m.AL = np.tile(-eye, m.NPL)
m.AU = np.tile(-eye, m.NPU)
m.B = rng.standard_normal(nd * NP)
m.X = np.zeros(nd * NP)
t_gen = time.time() - t0
out = {"ndof": nd, "nodes": NP, "dof": nd * NP, "blocks": int(NP + m.NPL + m.NPU), "t_generate_s": round(t_gen, 2)}
ctx = hip.SolverContext()
for name, meth, pc in (("cg_diag", 1, 3), ("cg_ssor", 1, 1), ("bicgstab_ssor", 2, 1)):
    m.Iarray[0], m.Iarray[1], m.Iarray[2] = a.steps, meth, pc
    m.Rarray[0] = 1e-30
    m.Iarray[96] = 1
    m.X[:] = 0.0
    t0 = time.time()
    code = hip.hecmw_solve(None, m, ctx=ctx)
    wall = time.time() - t0
    out[name] = {"code": int(code), "iters": int(ctx.info.iterations) - 1, "it_per_s": round((ctx.info.iterations - 1) / ctx.info.time_sol, 1),
                 "t_setup_s": round(ctx.info.time_setup, 2), "wall_s": round(wall, 2), "last_resid": float(ctx.history[-1]),
                 "ncolor": int(ctx.info.ncolor)}
ms = C.c_float(0)
st = (C.c_int64 * 4)()
hip._chk(hip.lib().fx_nn_matvec_resident(ctx.h, 20, C.byref(ms), st))
nb = st[3]
alg = nb * (nd2 * 8 + 4) + 2 * 4 * (NP + 1) + 2 * nd * 8 * NP
out["spmv"] = {"ms": round(ms.value, 4), "algorithmic_GB": round(alg / 1e9, 4), "GBps": round(alg / ms.value / 1e6, 1),
               "frac_of_8TBps": round(alg / ms.value / 1e6 / 8000.0, 3), "padded_blocks": int(st[2]),
               "padding": round(st[2] / nb - 1.0, 4)}
print(json.dumps(out))
