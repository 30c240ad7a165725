"""tests/golden/recycle.npz: the REFERENCE's hecmw_solve run six times in a row by oracle/ref_solve_driver.f90 mode 4 -- after every
solve the diagonal blocks grow by 10 %, X is reset and Iarray(97) = 1 is raised, as a Newton loop does -- so that the recycle policy
of the preconditioner (hecmw_mat_recycle_precond_setting, hecmw_matrix_misc.f90:678-697: solves 2-4 re-use the preconditioner of
solve 1, solve 5 rebuilds) shows in the iteration counts.  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from conftest import golden_matrix, load_golden              # noqa: E402
from nn_cases import nn_system                                # noqa: E402
from oracle import refrun                                     # noqa: E402
import test_oracle_golden as T                                # noqa: E402

out = {}
for name, meth, pc, thr in T.RECYCLE_CASES:
    A = nn_system(int(name[2:])) if name.startswith("nn") else golden_matrix(load_golden(name))
    I, R = refrun.default_params(method=meth, precond=pc)
    r = refrun.run_solve(A, I, R, threads=thr, mode=4, nrepeat=6)
    tag = T.recycle_tag(name, meth, pc)
    out[tag + "iters"] = np.array(r["iters"], dtype=np.int32)
    out[tag + "X"] = r["X"]
    out[tag + "Iarray"] = r["Iarray"]
    print(tag, r["iters"], r["Iarray"][95:98])
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "recycle.npz"), **out)
