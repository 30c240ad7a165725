"""Diagnostic: per-iteration deviation of the GPU residual history from the oracle's."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh
from oracle import pyoracle as po
from oracle.refrun import default_params
from conftest import golden_matrix, load_golden

def run(A, meth, pc, name):
    I, R = default_params(method=meth, precond=pc)
    o = po.solve_iterative(A, I, R, nthreads=4)
    m = hip.hecmwST_matrix.from_arrays(A.N, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, A.B)
    m.Iarray[0] = 10000; m.Iarray[1] = meth; m.Iarray[2] = pc
    ctx = hip.SolverContext()
    code = hip.hecmw_solve(None, m, ctx=ctx)
    h, ho = ctx.history, o["history"]
    n = min(len(h), len(ho))
    rel = np.abs(h[:n] - ho[:n]) / ho[:n]
    print(f"{name} m{meth} p{pc}: code {code} gpu it {ctx.info.iterations} orc it {o['iter']} "
          f"xerr {np.abs(m.X - o['X']).max() / np.abs(o['X']).max():.2e} final resid gpu {ctx.info.resid:.3e} rel {ctx.info.rel_resid:.3e}")
    idx = list(range(0, n, max(1, n // 12)))
    print("   k     h_orc       h_gpu       rel")
    for k in idx + [n - 1]:
        print(f"  {k+1:4d} {ho[k]:.6e} {h[k]:.6e} {rel[k]:.1e}")
    ctx.close()

mesh = CubeMesh(20)
A20 = po.assemble(1, mesh.coord, mesh.conn, 210000.0, 0.3, bc=mesh.dirichlet(), load=mesh.load())
for meth, pc in [(1, 3), (1, 1), (2, 3), (2, 1)]:
    run(A20, meth, pc, "cube20")
for deck in ("cube3s", "exA_A361"):
    A = golden_matrix(load_golden(deck))
    for meth, pc in [(1, 3), (2, 3), (2, 1)]:
        run(A, meth, pc, deck)
