"""In-process A/B of the dataflow sweep's knobs at 10.1M DOF (BiCGSTAB + ILU(0)): one matrix, one box, contexts differ only in
FX_DF_* (read at fx_create).  usage: python scripts/experiments/ab_dataflow3.py [variants...]  (variant = poll:wps:grid[:sleep])"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh

n = int(os.environ.get("AB_N", "149"))
mesh = CubeMesh(n)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
variants = sys.argv[1:] or ["0:8:128", "1:8:128", "0:4:128", "1:4:128"]
ctxs = {}
for v in variants:
    poll, wps, grid, slp = (v.split(":") + ["2"])[:4]
    os.environ.update(FX_DATAFLOW="1", FX_DF_POLL=poll, FX_DF_WPS=wps, FX_DF_GRID=grid, FX_DF_SLEEP=slp)
    if poll == "L":
        os.environ["FX_DATAFLOW"] = "0"
    ctx = hip.SolverContext()
    ctx.upload(m, what=hip.FX_UP_PROFILE)
    ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
    m.Iarray[1] = 2; m.Iarray[2] = 10
    ctx.precond_setup(m)
    ctxs[v] = ctx
    print("set up", v, flush=True)
for rep in range(3):
    for v in variants:
        print("rep %d  %-12s precond_apply %.3f ms" % (rep, v, ctxs[v].precond_apply_ms(10)), flush=True)
