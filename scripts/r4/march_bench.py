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
if os.environ.get('FX_LIBPATH'): hip.LIBPATH = os.environ['FX_LIBPATH']   # timing experiments: a library built with -DFX_MARCH_EXP_*
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
    if os.environ.get("MARCH_TRACE"):
        tr = ctx.march_trace()
        np.save("gpurun_out/r4/march_trace_%s.npy" % spec.replace(":", "_"), tr)
        k = np.linspace(0, len(tr) - 1, 12).astype(int)
        print("   chunk   fwd start..end      bwd start..end (us)   waited/polls fwd  bwd")
        for i in k:
            print("   %5d  %8.1f %8.1f   %8.1f %8.1f   %6d %8d  %6d %8d" % ((i,) + tuple(tr[i, :4]) + tuple(int(v) for v in tr[i, 4:])), flush=True)
        print("   totals: fwd done %.1f us, all done %.1f us, waited rounds %d / %d, polls %d / %d"
              % (tr[:, 1].max(), tr[:, 3].max(), tr[:, 4].sum(), tr[:, 6].sum(), tr[:, 5].sum(), tr[:, 7].sum()), flush=True)
    for ch in [int(x) for x in os.environ.get("MARCH_ROUNDS", "").split(",") if x]:
        if ch >= rep["chunks"]:
            continue
        rr = ctx.march_rounds(ch)
        tt = np.abs(rr); dt = np.diff(tt); w = rr[1:] < 0
        print("   chunk %d: first round at %.1f us, %d rounds in %.1f us; %d waited; round time median %.2f us, free rounds mean %.2f, waited rounds mean %.2f, max %.1f"
              % (ch, tt[0], len(rr), tt[-1] - tt[0], int(w.sum()), np.median(dt), dt[~w].mean() if (~w).any() else 0, dt[w].mean() if w.any() else 0, dt.max()), flush=True)
        print("     first 60 round times (us, * = waited): " + " ".join(("%.1f%s" % (d, "*" if ww else "")) for d, ww in zip(dt[:60], w[:60])), flush=True)
        mid = len(dt) // 2
        print("     middle 60: " + " ".join(("%.1f%s" % (d, "*" if ww else "")) for d, ww in zip(dt[mid:mid + 60], w[mid:mid + 60])), flush=True)
    ctx.set_option("FX_MARCH", 0)
    if zref is None:
        zref = ctx.precond_apply(r)[:3 * m.N].copy()
    ms_d = [ctx.precond_apply_ms(10) for _ in range(2)]
    st = ctx.stats()
    print("   march %s ms per apply | dataflow %s ms | bit-identical %s | fallbacks %d | grid %d"
          % (" ".join("%.3f" % x for x in ms_m), " ".join("%.3f" % x for x in ms_d), bool(np.array_equal(z, zref)), st["df_fallbacks"],
             ctx.march_report()["grid"]), flush=True)
    ctx.close()
