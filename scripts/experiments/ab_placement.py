"""Does the placement of the value stream change the SpMV time? (one process, natural order, DIAG)"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
mesh = CubeMesh(149)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node); hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
for pad in (0, 0, 4096, 65536, 1 << 20, 3 << 20, 0, (1 << 21) + 4096, 0):
    os.environ["FX_VAL2_PAD"] = str(pad)
    ctx = hip.SolverContext()
    ctx.upload(m, what=hip.FX_UP_PROFILE)
    ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
    m.Iarray[2] = 3
    ctx.precond_setup(m)
    ctx.matvec_resident_ms(5)
    a = [ctx.matvec_resident_ms(20) for _ in range(3)]
    print("pad %9d  spmv ms %s" % (pad, ["%.4f" % x for x in a]), flush=True)
    ctx.close()
