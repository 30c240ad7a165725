"""In-process A/B of the multicolour SSOR apply at 10.1M DOF: contexts differ only in the FX_* knobs given as
KEY=VAL,KEY=VAL variants.  usage: python scripts/experiments/ab_ssor.py "FX_SSOR_BS=256" "FX_SPLIT_MAX_SLICES=1000" ..."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh

n = int(os.environ.get("AB_N", "149"))
mesh = CubeMesh(n)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
variants = ["default"] + sys.argv[1:]
ctxs = {}
for v in variants:
    saved = dict(os.environ)
    if v != "default":
        for kv in v.split(","):
            k, val = kv.split("=")
            os.environ[k] = val
    ctx = hip.SolverContext()
    os.environ.clear(); os.environ.update(saved)
    ctx.upload(m, what=hip.FX_UP_PROFILE)
    ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
    m.Iarray[1] = 1; m.Iarray[2] = 1
    ctx.precond_setup(m)
    ctxs[v] = ctx
    print("set up", v, flush=True)
import time
def cg_ms(ctx, steps=40):
    m.Iarray[0] = 1000; m.Rarray[0] = 1e-30
    ctx.krylov_begin(m)
    ctx.krylov_steps(8)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.krylov_steps(steps)
    ctx.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps
for rep in range(3):
    for v in variants:
        print("rep %d  %-40s precond_apply %.4f ms   spmv(dot) %.4f ms   CG iteration %.4f ms" % (rep, v, ctxs[v].precond_apply_ms(20), ctxs[v].spmv_resident_ms(1, 20), cg_ms(ctxs[v])), flush=True)
