"""tests/golden/krylov2.npz: the REFERENCE's hecmw_solve (oracle/_ref/ref_solve[_omp]) with METHOD=3 (GMRES)
and METHOD=4 (GPBiCG) on the matrices of the committed decks.  Run in the build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from conftest import golden_matrix, load_golden             # noqa: E402
from oracle import refrun                                    # noqa: E402
import test_oracle_golden as T                               # noqa: E402

out = {}
for deck, meth, pc, thr, maxit, nrest in T.KRYLOV2_CASES:
    A = golden_matrix(load_golden(deck))
    I, R = refrun.default_params(method=meth, precond=pc, maxit=maxit)
    I[5] = nrest
    r = refrun.run_solve(A, I, R, threads=thr)
    tag = T.krylov2_tag(deck, meth, pc, thr, maxit, nrest)
    out[tag + "iter"] = np.int32(r["iter"])
    out[tag + "resid"] = np.float64(r["resid"])
    out[tag + "hist"] = np.array([h[1] for h in r["history"]])
    out[tag + "X"] = r["X"]
    out[tag + "Iarray"] = r["Iarray"]
    print(tag, r.get("banner"), "iter", r["iter"], "resid", r["resid"], "hist", len(r["history"]), "conv", r["Iarray"][80])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "krylov2.npz"), **out)
