"""Which (method, precond[, scaling]) runs on exA_A361 converge on the GPU and what the device-resident scalars look like
when one does not (for the parity tests' breakdown assertions)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_matrix, load_golden
from frontistr_amd import hecmw as hip
import ctypes as C
names = "rho rho1 beta c1 alpha omega c2 cg0 cg1 dnrm2 bnrm2 resid tol iter status need_verify".split()
A = golden_matrix(load_golden("exA_A361"))
for scaling in (0, 1):
    for meth in (2, 4):
        for pc in (3, 1, 10):
            if meth == 4 and pc == 10 and False:
                continue
            m = hip.hecmwST_matrix.from_arrays(A.N, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, A.B.copy())
            m.Iarray[0] = 10000; m.Iarray[1] = meth; m.Iarray[2] = pc; m.Iarray[6] = scaling
            ctx = hip.SolverContext()
            try:
                code = hip.hecmw_solve(None, m, ctx=ctx)
            except Exception as e:
                print("scaling", scaling, "meth", meth, "pc", pc, "EXC", e); ctx.close(); continue
            out = (C.c_double * 16)()
            hip.lib().fx_debug_state(ctx.h, out)
            st = dict(zip(names, list(out)))
            h = ctx.history
            print("scaling %d meth %d pc %2d: code %4d iters %4d conv %d rel_resid %.3e | last hist %s | rho %.3e c2 %.3e omega %.3e cg0 %.3e cg1 %.3e dnrm2 %.3e bnrm2 %.3e"
                  % (scaling, meth, pc, code, ctx.info.iterations, m.Iarray[80], ctx.info.rel_resid,
                     " ".join("%.2e" % v for v in h[-3:]), st["rho"], st["c2"], st["omega"], st["cg0"], st["cg1"], st["dnrm2"], st["bnrm2"]), flush=True)
            ctx.close()
