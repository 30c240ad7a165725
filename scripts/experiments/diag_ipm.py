import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from frontistr_amd import hecmw as hip
from oracle import pyoracle as po
from oracle.refrun import default_params
from conftest import golden_matrix, load_golden
g = load_golden("cube3s"); A = golden_matrix(g)
for meth, pc, ipm in [(1, 3, 3), (1, 1, 0), (1, 3, 2), (1, 1, 2)]:
    I, R = default_params(method=meth, precond=pc, iterpremax=ipm)
    o = po.solve_iterative(A, I, R, nthreads=4)
    m = hip.hecmwST_matrix.from_arrays(A.N, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, A.B.copy())
    m.Iarray[0] = 10000; m.Iarray[1] = meth; m.Iarray[2] = pc; m.Iarray[4] = ipm
    ctx = hip.SolverContext()
    code = hip.hecmw_solve(None, m, ctx=ctx)
    h, ho = ctx.history, o["history"]
    n = min(len(h), len(ho), 12)
    print(meth, pc, ipm, "code", code, o["code"], "it", ctx.info.iterations, o["iter"], "xerr", np.abs(m.X - o["X"]).max() / np.abs(o["X"]).max())
    print("  gpu", h[:n]); print("  orc", ho[:n])
    ctx.close()
