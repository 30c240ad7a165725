"""tests/golden/nn.npz: the REFERENCE's hecmw_solve (oracle/_ref/ref_solve_omp, 4 OpenMP threads = RCM + multicolour SSOR) on
the synthetic NDOF = 1, 2, 4, 5, 6 systems of tests/nn_cases.py -- the 11 / 22 / 44 / nn / 66 code paths of
hecmw1/src/solver/{las,precond}.  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from nn_cases import NN_CASES, NN_NDOF, nn_system, nn_tag   # noqa: E402
from oracle import refrun                                    # noqa: E402

out = {}
for nd in NN_NDOF:
    A = nn_system(nd)
    for meth, pc in NN_CASES:
        I, R = refrun.default_params(method=meth, precond=pc)
        r = refrun.run_solve(A, I, R, threads=4)
        tag = nn_tag(nd, meth, pc)
        out[tag + "iter"] = np.int32(r["iter"])
        out[tag + "hist"] = np.array([h[1] for h in r["history"]])
        out[tag + "X"] = r["X"]
        print(tag, r["iter"], r["Iarray"][80])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "nn.npz"), **out)
