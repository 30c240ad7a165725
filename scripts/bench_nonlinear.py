"""Timing of one Newton iteration of the nonlinear static loop (configs[4] shape: elastoplastic,
updated Lagrange, B-bar) on the synthetic cube: tangent assembly, linear solve, stress update.
usage: python scripts/bench_nonlinear.py [n_elem_per_edge] [method] [precond] [newton_iters]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from frontistr_amd import fstr, hecmw as hip              # noqa: E402
from frontistr_amd.mesh import CubeMesh                   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
method = int(sys.argv[2]) if len(sys.argv) > 2 else 1
precond = int(sys.argv[3]) if len(sys.argv) > 3 else 1
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
TABLE = [[450, 0], [608, 0.05], [679, 0.1], [732, 0.2], [752, 0.3], [766, 0.4], [780, 0.5]]
m = CubeMesh(n)
mat = fstr.tMaterial(206900.0, 0.29, plastic=True, harden=fstr.MULTILINEAR, table=TABLE, nlgeom_flag=fstr.UPDATELAG)
hm = hip.hecmwST_local_mesh(n_node=m.n_node)
hm.elem_node_item = m.conn.ravel()
t0 = time.time()
hecMAT = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
t_con = time.time() - t0
ctx = hip.SolverContext()
ctx.upload(hecMAT, what=hip.FX_UP_PROFILE)
solid = fstr.fstr_solid(ctx, m.coord, m.conn, mat)
bn, bd, bv = m.dirichlet()
tn = m.top_nodes.astype(np.int32)
bc = (np.concatenate([bn, tn]), np.concatenate([bd, np.full(tn.size, 3, dtype=np.int32)]),
      np.concatenate([bv, np.full(tn.size, 0.004 * n)]))           # 0.4 % stretch per substep: plastic from the start
hecMAT.Iarray[0] = 20000; hecMAT.Iarray[1] = method; hecMAT.Iarray[2] = precond
hecMAT.Iarray[20] = 0; hecMAT.Iarray[21] = 0
hecMAT.Rarray[0] = 1e-6
from frontistr_amd.hecmw import lib, _chk                 # noqa: E402
_chk(lib().fx_nl_begin_substep(ctx.h, None))
rows = []
for it in range(1, iters + 1):
    ms_k = fstr.fstr_StiffMatrix(solid, (bc[0], bc[1], bc[2] if it == 1 else np.zeros_like(bc[2])))
    hecMAT.Iarray[96] = 2 if it == 1 else 1
    hecMAT.X[:] = 0.0
    ctx.upload(hecMAT, what=hip.FX_UP_X)
    t0 = time.time()
    code = ctx.solve_resident(hecMAT, want_history=False)
    t_solve = time.time() - t0
    (res, xn, qn, dun), ms_u = fstr.fstr_UpdateNewton(solid)
    rows.append(dict(iter=it, tangent_ms=ms_k, update_ms=ms_u, solve_s=t_solve, solver_iters=int(ctx.info.iterations),
                     solver_code=int(code), setup_s=float(ctx.info.time_setup), rres=res / max(qn, 1e-300), rx=xn / max(dun, 1e-300)))
    print(rows[-1], flush=True)
st = solid.get_state(("plstrain", "istat"))
out = dict(n_elem=m.n_elem, dof=3 * m.n_node, mat_con_s=t_con, method=method, precond=precond, iterations=rows,
           plastic_points=int(st["istat"].sum()), max_plstrain=float(st["plstrain"].max()))
print(json.dumps(out))
