"""One SSOR set-up at 150^3 nodes (for rocprofv3 --kernel-trace --stats of the set-up kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
mesh = CubeMesh(n - 1)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
os.environ["FX_TIMING"] = "1"
for rep in range(2):
    ctx = hip.SolverContext()
    ctx.upload(m, what=hip.FX_UP_PROFILE)
    ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
    m.Iarray[2] = 1
    t0 = time.time()
    ctx.precond_setup(m)
    print("precond_setup %.3f s" % (time.time() - t0), file=sys.stderr, flush=True)
    ctx.close()
