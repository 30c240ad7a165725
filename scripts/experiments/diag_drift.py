"""Measured drift of the GPU residual history from the CPU oracle (same ordering, bit-exact with the reference), per deck and
configuration: the numbers behind the tolerances of DESIGN.md section 5.  One line per run:
deck method precond | it gpu/oracle | max rel deviation over history lines 1-10, 11-30, 31-end | x error."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
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
    seg = lambda a, b: ("%.1e" % rel[a:b].max()) if rel[a:b].size else "   -   "
    print("%-9s m%d p%-2d | it %4d / %4d | hist 1-10 %s  11-30 %s  31-end %s | x err %.1e | code %d" %
          (name, meth, pc, ctx.info.iterations, o["iter"], seg(0, 10), seg(10, 30), seg(30, n),
           np.abs(m.X - o["X"]).max() / np.abs(o["X"]).max(), code), flush=True)
    ctx.close()

mesh = CubeMesh(20)
A20 = po.assemble(1, mesh.coord, mesh.conn, 210000.0, 0.3, bc=mesh.dirichlet(), load=mesh.load())
decks = [("cube4", golden_matrix(load_golden("cube4"))), ("cube3s", golden_matrix(load_golden("cube3s"))), ("cube20", A20),
         ("exA_A361", golden_matrix(load_golden("exA_A361")))]
for name, A in decks:
    for meth, pc in [(1, 3), (1, 1), (1, 10), (2, 3), (2, 1), (2, 10)]:
        run(A, meth, pc, name)
