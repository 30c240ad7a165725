"""Same-context A/B of the SpMV (the CG loop's variant) under fx_set_option knobs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frontistr_amd import hecmw as hip
if os.environ.get('FX_LIBPATH'): hip.LIBPATH = os.environ['FX_LIBPATH']
from frontistr_amd.mesh import CubeMesh
mesh = CubeMesh(int(os.environ.get("AB_N", "149")))
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node); hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
ctx = hip.SolverContext(); ctx.upload(m, what=hip.FX_UP_PROFILE)
ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
m.Iarray[1] = 1; m.Iarray[2] = int(os.environ.get("AB_PRECOND", "1")); ctx.precond_setup(m)
variants = ["default"] + sys.argv[1:]
keys = sorted({kv.split("=")[0] for v in variants[1:] for kv in v.split(",")})
for rep in range(3):
    for v in variants:
        for k in keys: ctx.set_option(k, 0)
        if v != "default":
            for kv in v.split(","):
                k, val = kv.split("="); ctx.set_option(k, float(val))
        print("rep %d %-30s spmv(dot) %.4f ms" % (rep, v, ctx.spmv_resident_ms(1, 20)), flush=True)
