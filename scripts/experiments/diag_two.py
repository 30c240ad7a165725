import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
from frontistr_amd import hecmw as hip
from conftest import golden_matrix, load_golden
g = load_golden("exA_A361"); A = golden_matrix(g)
names = "rho rho1 beta c1 alpha omega c2 cg0 cg1 dnrm2 bnrm2 resid tol iter status need_verify".split()
m = hip.hecmwST_matrix.from_arrays(A.N, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, A.B.copy())
m.Iarray[0] = 10000; m.Iarray[1] = 2; m.Iarray[2] = 1
ctx = hip.SolverContext()
ctx.upload(m); ctx.precond_setup(m); ctx.krylov_begin(m)
ctx.krylov_steps(190)
for k in range(191, 201):
    it, st, rs = ctx.krylov_steps(1)
    out = (C.c_double * 16)()
    hip.lib().fx_debug_state(ctx.h, out)
    print(k, " ".join("%s=%.6g" % (n, v) for n, v in zip(names, out) if n not in ("tol", "bnrm2", "c1")))
