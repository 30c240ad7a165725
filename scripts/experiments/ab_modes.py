"""In-process A/B of the SSOR numbering modes (and SpMV) at 10M DOF: alternating contexts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
mesh = CubeMesh(n)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node); hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
for rnd in range(2):
    for mode in (0, 1):
        os.environ["FX_SSOR_MODE"] = str(mode)
        ctx = hip.SolverContext()
        ctx.upload(m, what=hip.FX_UP_PROFILE)
        ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
        m.Iarray[0] = 400; m.Iarray[1] = 1; m.Iarray[2] = 1; m.Rarray[0] = 1e-30
        ctx.precond_setup(m)
        ctx.krylov_begin(m); ctx.krylov_steps(10); ctx.synchronize()
        t0 = time.perf_counter(); it, st, rs = ctx.krylov_steps(100); ctx.synchronize(); dt = time.perf_counter() - t0
        sp = ctx.matvec_resident_ms(20); pr = ctx.precond_apply_ms(10)
        print("mode %d: %.1f it/s (%.3f ms/it)  spmv %.4f ms  ssor %.4f ms  resid %.6e" % (mode, 100 / dt, 10 * dt, sp, pr, rs), flush=True)
        ctx.close()
