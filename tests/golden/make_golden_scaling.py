"""tests/golden/scaling.npz: the REFERENCE's hecmw_solve with SCALING=YES (Iarray(7)=1; symmetric diagonal scaling,
las/hecmw_solver_scaling_33.f90) on the matrices of the committed decks.  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from conftest import golden_matrix, load_golden             # noqa: E402
from oracle import refrun                                    # noqa: E402
import test_oracle_golden as T                               # noqa: E402

out = {}
for deck, meth, pc, thr in T.SCALING_CASES:
    A = golden_matrix(load_golden(deck))
    I, R = refrun.default_params(method=meth, precond=pc)
    I[6] = 1
    r = refrun.run_solve(A, I, R, threads=thr)
    tag = T.scaling_tag(deck, meth, pc, thr)
    out[tag + "iter"] = np.int32(r["iter"])
    out[tag + "hist"] = np.array([h[1] for h in r["history"]])
    out[tag + "X"] = r["X"]
    out[tag + "Iarray"] = r["Iarray"]
    print(tag, r.get("banner"), "iter", r["iter"], "conv", r["Iarray"][80])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "scaling.npz"), **out)
