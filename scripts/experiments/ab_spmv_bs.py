"""SpMV workgroup size 64 vs 256 (FX_SPMV_BS) with the spatial slice order, pipelined row loop; 4 alternating rounds
in one process to average the allocation-placement effect out."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
mesh = CubeMesh(149)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node); hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
acc = {64: [], 256: []}
for rnd in range(4):
    for bs in ((64, 256) if rnd % 2 == 0 else (256, 64)):
        os.environ["FX_SPMV_BS"] = str(bs)
        ctx = hip.SolverContext()
        ctx.upload(m, what=hip.FX_UP_PROFILE)
        ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
        m.Iarray[0] = 400; m.Iarray[1] = 1; m.Iarray[2] = 1; m.Rarray[0] = 1e-30
        ctx.precond_setup(m)
        ctx.krylov_begin(m); ctx.krylov_steps(10); ctx.synchronize()
        t0 = time.perf_counter(); it, st, rs = ctx.krylov_steps(100); ctx.synchronize(); dt = time.perf_counter() - t0
        b = min(ctx.matvec_resident_ms(20) for _ in range(3))
        acc[bs].append((100 / dt, b))
        print("spmv block %3d: %.1f it/s  spmv %.4f ms" % (bs, 100 / dt, b), flush=True)
        ctx.close()
for bs, v in acc.items():
    print("block %3d: mean %.1f it/s, mean spmv %.4f ms, best %.4f ms" % (bs, sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v), min(x[1] for x in v)))
