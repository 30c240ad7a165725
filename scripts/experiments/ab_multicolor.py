"""Device multicolouring of the SSOR set-up at 150^3 nodes: host look-in interval sweep, host walk last (FX_TIMING lines on stderr)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
mesh = CubeMesh(n - 1)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
os.environ["FX_TIMING"] = "1"
os.environ["FX_TUNE_PLACEMENT"] = "0"
for window, batch in ((0, 32), (0, 16), (0, 64), (0, 128), (-1, 0)):
    if window < 0:
        os.environ["FX_MC_DEVICE_MIN"] = str(2**31 - 1)
    else:
        os.environ["FX_MC_BATCH"] = str(batch)
    ctx = hip.SolverContext()
    ctx.upload(m, what=hip.FX_UP_PROFILE)
    ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
    m.Iarray[2] = 1
    print("=== window", window, "batch", batch, file=sys.stderr, flush=True)
    t0 = time.time()
    ctx.precond_setup(m)
    print("precond_setup %.3f s" % (time.time() - t0), file=sys.stderr, flush=True)
    ctx.close()
