"""In-process A/B of the ILU(0) sweeps at 10.1M DOF: level dataflow (FX_DATAFLOW=1) against chain sweeps (FX_DATAFLOW=3) with
several grids / hop prices.  usage: python scripts/experiments/ab_chain.py [variants...]  (variant = D (dataflow) or grid:hop)"""
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
variants = sys.argv[1:] or ["D", "0:12", "1024:12", "2048:12", "0:6", "0:24"]
ctxs = {}
os.environ["FX_TIMING"] = "1"
for v in variants:
    if v == "D":
        os.environ.update(FX_DATAFLOW="1")
    else:
        grid, hop = v.split(":")
        os.environ.update(FX_DATAFLOW="3", FX_CH_GRID=grid, FX_CH_HOP=hop)
    ctx = hip.SolverContext()
    ctx.upload(m, what=hip.FX_UP_PROFILE)
    ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
    m.Iarray[1] = 2; m.Iarray[2] = 10
    t0 = time.time()
    ctx.precond_setup(m)
    ctxs[v] = ctx
    print("set up %-10s %.2f s chain=%d" % (v, time.time() - t0, ctx.stats()["chain_sweeps"]), flush=True)
r = np.cos(0.11 * np.arange(3 * m.NP) + 0.3)
zref = None
for v in variants:
    z = ctxs[v].precond_apply(r)
    if zref is None:
        zref = z
    print("%-10s max rel diff to first %.2e" % (v, np.abs(z - zref).max() / np.abs(zref).max()), flush=True)
for rep in range(3):
    for v in variants:
        print("rep %d  %-10s precond_apply %.3f ms" % (rep, v, ctxs[v].precond_apply_ms(10)), flush=True)
