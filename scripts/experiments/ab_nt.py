"""A/B of non-temporal matrix-stream loads: two library builds, alternating processes."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, time
sys.path.insert(0, %r)
from frontistr_amd import hecmw as hip
hip.LIBPATH = os.path.join(%r, "frontistr_amd", sys.argv[1])
from frontistr_amd.mesh import CubeMesh
mesh = CubeMesh(149)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node); hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
ctx = hip.SolverContext()
ctx.upload(m, what=hip.FX_UP_PROFILE)
ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
m.Iarray[0] = 400; m.Iarray[1] = 1; m.Iarray[2] = 1; m.Rarray[0] = 1e-30
ctx.precond_setup(m)
ctx.krylov_begin(m); ctx.krylov_steps(10); ctx.synchronize()
t0 = time.perf_counter(); ctx.krylov_steps(100); ctx.synchronize(); dt = time.perf_counter() - t0
sp = [ctx.matvec_resident_ms(20) for _ in range(2)]; pr = [ctx.precond_apply_ms(10) for _ in range(2)]
print(sys.argv[1], "%%.1f it/s  spmv %%s  ssor %%s" %% (100 / dt, ["%%.4f" %% x for x in sp], ["%%.4f" %% x for x in pr]), flush=True)
''' % (ROOT, ROOT)
for rnd in range(2):
    for lib in ("libfistr_hip.so", "libfistr_hip_nt.so"):
        subprocess.run([sys.executable, "-c", code, lib])
