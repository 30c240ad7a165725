"""PCIe-inclusive cost of the host-array boundary (fx_solve / fx_upload) at 10.1M DOF: upload of D/AL/AU
(6.9 GB) from the caller's (pageable) arrays, first and repeated, and the whole fx_solve vs fx_solve_resident."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
mesh = CubeMesh(149)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node); hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
ctx = hip.SolverContext()
ctx.upload(m, what=hip.FX_UP_PROFILE)
ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
t0 = time.perf_counter(); ctx.download_matrix(m); t_down = time.perf_counter() - t0
gb = (m.D.nbytes + m.AL.nbytes + m.AU.nbytes) / 1e9
print("download %.2f GB in %.3f s = %.1f GB/s" % (gb, t_down, gb / t_down))
ctx.close()
ctx = hip.SolverContext()
for k in range(3):
    t0 = time.perf_counter(); ctx.upload(m, what=hip.FX_UP_ALL if k == 0 else hip.FX_UP_VALUES); ctx.synchronize(); dt = time.perf_counter() - t0
    print("upload #%d (%s) %.3f s = %.1f GB/s" % (k, "profile+values" if k == 0 else "values", dt, gb / dt))
m.Iarray[0] = 10000; m.Iarray[1] = 1; m.Iarray[2] = 1
t0 = time.perf_counter(); code = hip.hecmw_solve(None, m, ctx=ctx); t_all = time.perf_counter() - t0
print("hecmw_solve from host arrays: %.2f s total, %d iterations, setup %.2f s, solve %.2f s" % (t_all, ctx.info.iterations, ctx.info.time_setup, ctx.info.time_sol))
