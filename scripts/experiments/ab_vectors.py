"""Does the placement of the GATHERED / WRITTEN vectors matter to the SpMV?  One context, one matrix placement, the timed product
run over different pairs of work vectors (FX_SPMV_XY is read at every call)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
mesh = CubeMesh(149)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
ctx = hip.SolverContext()
ctx.upload(m, what=hip.FX_UP_PROFILE)
ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
m.Iarray[1] = 1; m.Iarray[2] = 1
ctx.precond_setup(m)
for rep in range(2):
    os.environ.pop("FX_SPMV_XY", None)
    print("rep", rep, "Bs->W7  %.4f ms" % ctx.spmv_resident_ms(1, 20), flush=True)
    for xy in ("2,1", "1,2", "0,1", "3,4", "5,6", "6,5", "2,7", "0,7"):
        os.environ["FX_SPMV_XY"] = xy
        a, b = xy.split(",")
        print("rep", rep, "W%s->W%s  %.4f ms" % (a, b, ctx.spmv_resident_ms(1, 20)), flush=True)
