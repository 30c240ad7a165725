import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
mesh = CubeMesh(149)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node); hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
for rnd in range(2):
    for thr in (1000000, 4096, 0, 2048):
        os.environ["FX_PIPE_MAX_SLICES"] = str(thr)
        ctx = hip.SolverContext()
        ctx.upload(m, what=hip.FX_UP_PROFILE)
        ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
        m.Iarray[0] = 400; m.Iarray[1] = 1; m.Iarray[2] = 1; m.Rarray[0] = 1e-30
        ctx.precond_setup(m)
        ctx.precond_apply_ms(3)
        b = [ctx.precond_apply_ms(10) for _ in range(3)]
        print("pipe_max_slices %8d  ssor ms %s" % (thr, ["%.4f" % x for x in b]), flush=True)
        ctx.close()
