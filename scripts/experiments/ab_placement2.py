import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
mesh = CubeMesh(149)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node); hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
os.environ["FX_SSOR_MODE"] = "0"
for first in (0, 1, 0, 1):
    ctx = hip.SolverContext()
    ctx.upload(m, what=hip.FX_UP_PROFILE)
    ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
    m.Iarray[2] = 1
    if first:
        ctx.matvec_resident_ms(2)      # builds M before the SSOR structures
    ctx.precond_setup(m)
    ctx.matvec_resident_ms(5)
    a = [ctx.matvec_resident_ms(20) for _ in range(3)]
    b = [ctx.precond_apply_ms(10) for _ in range(2)]
    print("M first=%d  spmv ms %s  ssor ms %s" % (first, ["%.4f" % x for x in a], ["%.4f" % x for x in b]), flush=True)
    ctx.close()
