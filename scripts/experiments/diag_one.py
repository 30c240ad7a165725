import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from frontistr_amd import hecmw as hip
from conftest import golden_matrix, load_golden
deck, meth, pc = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
g = load_golden(deck); A = golden_matrix(g)
m = hip.hecmwST_matrix.from_arrays(A.N, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, A.B.copy())
m.Iarray[0] = 10000; m.Iarray[1] = meth; m.Iarray[2] = pc
ctx = hip.SolverContext()
code = hip.hecmw_solve(None, m, ctx=ctx)
h = ctx.history
print("code", code, "iter", ctx.info.iterations, "resid", ctx.info.resid, "rel", ctx.info.rel_resid, "flags", m.Iarray[80], m.Iarray[81])
print("hist tail", h[-8:], "min", h.min(), "argmin", h.argmin(), "nan", np.isnan(h).any())
k = int(np.argmax(np.isnan(h))) if np.isnan(h).any() else len(h)
print("first nan at", k + 1, "before:", h[max(0, k - 12):k + 1])
