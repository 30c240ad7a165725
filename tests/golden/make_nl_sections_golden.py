"""tests/golden/nl_steps_sections_*.npz: the REFERENCE's load-step loop (oracle/ref_nl_driver.f90 mode 2: STF_C3D8Bbar, Update_C3D8Bbar,
hecmw_mat_ass_*, hecmw_solve) on the step deck of tests/test_oracle_nl.py with THREE sections / materials in one element group:
 * ul:    bilinear Mises + multilinear Mises + elastic, all UPDATELAG
 * mixed: bilinear Mises (UPDATELAG) + elastic TOTALLAG -- what an !ELASTIC part next to a !PLASTIC part gives in an NLSTATIC run.
Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import test_oracle_nl as T                                   # noqa: E402
from oracle import refrun                                    # noqa: E402

for name in T.SECTION_CASES:
    ms, emat, m, bc, cload, I, R, conv = T.sections_case(name)
    r = refrun.run_nl_steps(ms, m.coord, m.conn, *bc, cload, 3, 12, conv, I, R, elem_mat=emat)
    st = r["state"]
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "nl_steps_sections_%s.npz" % name),
                        log=r["log"], unode=r["unode"], qforce=r["qforce"], stress=st["stress"], strain=st["strain"],
                        plstrain=st["plstrain"], fstat=st["fstat"], istat=st["istat"], elem_mat=emat)
    print(name, r["log"].shape, st["plstrain"].max())
