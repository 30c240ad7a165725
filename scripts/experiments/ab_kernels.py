"""A/B timing of the SpMV and SSOR-apply kernels under env switches (one process per variant
is required because the switches are read at context creation; the GPU box is otherwise idle)."""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
mesh = CubeMesh(n)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
res = {}
for ps, pp in [(0, 0), (1, 1), (0, 0), (1, 1)]:
    os.environ["FX_PIPE_SPMV"] = str(ps)
    os.environ["FX_PIPE_SSOR"] = str(pp)
    ctx = hip.SolverContext()
    ctx.upload(m, what=hip.FX_UP_PROFILE)
    ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
    m.Iarray[2] = 1
    ctx.precond_setup(m)
    ctx.matvec_resident_ms(5); ctx.precond_apply_ms(3)
    a = [ctx.matvec_resident_ms(20) for _ in range(3)]
    b = [ctx.precond_apply_ms(10) for _ in range(3)]
    st = ctx.stats()
    print("pipe_spmv=%d pipe_ssor=%d  spmv ms %s  ssor ms %s" % (ps, pp, ["%.4f" % x for x in a], ["%.4f" % x for x in b]), flush=True)
    ctx.close()
print("stats", st)
print("L padding %.3f  U padding %.3f  M padding %.3f" % (2 * st["L_pairs"] * 64 / st["L_blocks"] - 1, 2 * st["U_pairs"] * 64 / st["U_blocks"] - 1, 2 * st["M_pairs"] * 64 / st["M_blocks"] - 1))
