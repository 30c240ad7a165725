import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
mesh = CubeMesh(149)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node); hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
ctx = hip.SolverContext()
t0 = time.perf_counter(); ctx.upload(m, what=hip.FX_UP_PROFILE); ctx.synchronize(); t1 = time.perf_counter()
ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet()); ctx.synchronize(); t2 = time.perf_counter()
ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet()); ctx.synchronize(); t3 = time.perf_counter()
print("FX_ASM_FIRST=%s: upload %.3f s, first assemble %.3f s, second %.3f s" % (os.environ.get("FX_ASM_FIRST", "1"), t1 - t0, t2 - t1, t3 - t2))
