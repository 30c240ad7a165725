"""Plane march (csrc/fx_march.h) against k_tri_dataflow on one system: ms per preconditioner apply, bit-equality of z, the program
report.  usage: python scripts/r4/march_bench.py [--precond 10|1] [--n 149] "chunk:waves[:grid[:xcd]]" ...   (0 = automatic)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
args = sys.argv[1:]
precond, n = 10, 149
if "--precond" in args:
    i = args.index("--precond"); precond = int(args[i + 1]); del args[i:i + 2]
if "--n" in args:
    i = args.index("--n"); n = int(args[i + 1]); del args[i:i + 2]
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
mesh = CubeMesh(n)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
r = np.cos(0.37 * np.arange(3 * m.NP) + 0.1)
zref = None
for spec in (args or ["0:0"]):
    f = [int(x) for x in spec.split(":")] + [0, 0, 0, 1]
    chunk, waves, grid = f[0], f[1], f[2]
    xcd = f[3] if len(spec.split(":")) > 3 else 1
    ctx = hip.SolverContext()
    ctx.set_option("FX_MARCH", 2)
    ctx.set_option("FX_MARCH_CHUNK", chunk); ctx.set_option("FX_MARCH_WAVES", waves); ctx.set_option("FX_MARCH_GRID", grid); ctx.set_option("FX_MARCH_XCD", xcd)
    if precond == 1:
        ctx.set_option("FX_SSOR_NATURAL", 1)
    ctx.upload(m, what=hip.FX_UP_PROFILE)
    ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
    m.Iarray[1] = 2 if precond == 10 else 1; m.Iarray[2] = precond
    t0 = time.perf_counter()
    ctx.precond_setup(m)
    t_setup = time.perf_counter() - t0
    rep = ctx.march_report()
    print("== %s  precond %d  N %d: setup %.2f s, report %s" % (spec, precond, m.N, t_setup, rep), flush=True)
    if not rep["built"]:
        ctx.close(); continue
    z = ctx.precond_apply(r)[:3 * m.N].copy()
    ms_m = [ctx.precond_apply_ms(10) for _ in range(3)]
    ctx.set_option("FX_MARCH", 0)
    if zref is None:
        zref = ctx.precond_apply(r)[:3 * m.N].copy()
    ms_d = [ctx.precond_apply_ms(10) for _ in range(2)]
    st = ctx.stats()
    print("   march %s ms per apply | dataflow %s ms | bit-identical %s | fallbacks %d | grid %d"
          % (" ".join("%.3f" % x for x in ms_m), " ".join("%.3f" % x for x in ms_d), bool(np.array_equal(z, zref)), st["df_fallbacks"],
             ctx.march_report()["grid"]), flush=True)
    ctx.close()
