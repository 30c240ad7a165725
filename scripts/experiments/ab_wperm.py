"""Does it matter WHICH physical work vectors play r / z,q / p in the CG loop?  One context, FX_W_PERM read at every krylov_begin."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
mesh = CubeMesh(149)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
ctxs = []
for dummy in ("0", "100", "3333", "7777"):
    os.environ["FX_DUMMY_MB"] = dummy
    ctx = hip.SolverContext()
    ctx.upload(m, what=hip.FX_UP_PROFILE)
    ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
    m.Iarray[1] = 1; m.Iarray[2] = 1
    ctx.precond_setup(m)
    ctxs.append(ctx)
os.environ.pop("FX_W_PERM", None)
def cg_ms(ctx, steps=40):
    m.Iarray[0] = 1000; m.Rarray[0] = 1e-30
    ctx.krylov_begin(m); ctx.krylov_steps(8); ctx.synchronize()
    t0 = time.perf_counter(); ctx.krylov_steps(steps); ctx.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps
for rep in range(2):
    for k, ctx in enumerate(ctxs):
        os.environ.pop("FX_W_PERM", None)
        print("rep", rep, "ctx", k, "precond %.4f spmv %.4f" % (ctx.precond_apply_ms(10), ctx.spmv_resident_ms(1, 10)), flush=True)
        for perm in ("0,1,2", "3,4,5", "5,6,7", "7,8,9", "0,7,2", "6,1,8"):
            os.environ["FX_W_PERM"] = perm
            print("rep", rep, "ctx", k, "roles r,z/q,p <- buffers", perm, " CG iteration %.4f ms" % cg_ms(ctx), flush=True)
