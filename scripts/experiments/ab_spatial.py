"""Same-process A/B: SpMV rows walked in slot (colour-major) order vs natural order with slot-mapped output
(FX_SPMV_SPATIAL), CG + SSOR at 10.1M DOF."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
mesh = CubeMesh(int(sys.argv[1]) if len(sys.argv) > 1 else 149)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node); hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
for rnd in range(2):
    for nat in (0, 1):
        os.environ["FX_SPMV_SPATIAL"] = str(nat)
        ctx = hip.SolverContext()
        ctx.upload(m, what=hip.FX_UP_PROFILE)
        ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
        m.Iarray[0] = 400; m.Iarray[1] = 1; m.Iarray[2] = 1; m.Rarray[0] = 1e-30
        ctx.precond_setup(m)
        ctx.krylov_begin(m); ctx.krylov_steps(10); ctx.synchronize()
        t0 = time.perf_counter(); it, st, rs = ctx.krylov_steps(100); ctx.synchronize(); dt = time.perf_counter() - t0
        a = [ctx.matvec_resident_ms(10) for _ in range(3)]
        s = ctx.stats()
        print("spatial slice order %d: %.1f it/s  spmv ms %s  resid %.6e" % (nat, 100 / dt, ["%.4f" % x for x in a], rs), flush=True)
        ctx.close()
