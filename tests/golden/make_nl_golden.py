"""Generates tests/golden/nl_*.npz from the REFERENCE routines (oracle/_ref/ref_nl, built by
oracle/build_ref.py --only nl from /root/reference): element tangent / stress update / internal force
of the C3D8 B-bar elastoplastic path and whole load-step loops.  Run in the build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from oracle import refrun                                      # noqa: E402
import test_oracle_nl as T                                     # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
for name in T.materials():
    mat, m, unode, dunode, st = T.element_case(name)
    ke0, qf, ke1, out = refrun.run_nl_elements(mat, m.coord, m.conn, unode, dunode, st)
    np.savez_compressed(os.path.join(HERE, "nl_elements_%s.npz" % name), ke0=ke0, qf=qf, ke1=ke1,
                        stress=out["stress"], strain=out["strain"], fstat=out["fstat"], istat=out["istat"])
    print(name, "elements ok")
for name in T.STEP_CASES:
    mat, m, bc, cload, I, R = T.step_case(name)
    out = refrun.run_nl_steps(mat, m.coord, m.conn, *bc, cload, 3, 12, T.STEP_CONVERG[name], I, R, threads=2)
    s = out["state"]
    np.savez_compressed(os.path.join(HERE, "nl_steps_%s.npz" % name), log=out["log"], unode=out["unode"],
                        qforce=out["qforce"], stress=s["stress"], strain=s["strain"], plstrain=s["plstrain"],
                        fstat=s["fstat"], istat=s["istat"])
    print(name, "steps ok:", out["log"].shape[0], "newton iterations, max plstrain", s["plstrain"].max())
