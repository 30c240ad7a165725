"""tests/golden/sections_cube3s.npz: the REFERENCE's assembly (STF_C3D8IC / STF_C3D8Bbar / STF_C3 through oracle/_ref/ref_fem,
hecmw_mat_ass_elem, hecmw_mat_ass_bc) of the cube3s deck with THREE sections / materials (hecMESH%section_ID ->
fstrSOLID%materials, one tMaterial per section) and the reference's CG + SSOR solution of that system.  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from conftest import load_golden                             # noqa: E402
from oracle import refrun                                    # noqa: E402

g = load_golden("cube3s")
n_elem = g["conn"].shape[0]
Es = np.array([210000.0, 70000.0, 3500.0])
nus = np.array([0.3, 0.33, 0.42])
elem_mat = (1 + np.arange(n_elem) % 3).astype(np.int32)
out = {"E": Es, "nu": nus, "elem_mat": elem_mat}
for eo, tag in [(1, "ic_"), (2, "bbar_"), (3, "fi_")]:
    A, ke, _ = refrun.run_fem(g["coord"], g["conn"], 0.0, 0.0, g["bc_node"], g["bc_dof"], g["bc_val"], g["load"], elemopt=eo,
                              sections=(Es, nus, elem_mat))
    for k in ("D", "AL", "AU", "B"):
        out[tag + k] = getattr(A, k)
    if eo == 1:
        I, R = refrun.default_params(method=1, precond=1)
        r = refrun.run_solve(A, I, R, threads=1)
        out["ic_X"], out["ic_iter"] = r["X"], np.int32(r["iter"])
        print("CG+SSOR", r["iter"], "iterations")
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "sections_cube3s.npz"), **out)
