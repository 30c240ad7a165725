"""tests/golden/nl_1elem.npz: the reference's one-element plasticity decks examples/static/1elem/{mises,swift,ramberg}
(.cnt/.msh: unit cube TYPE=361, symmetry planes x0/y0/z0, prescribed x-displacement of the x1 face in 10 substeps,
!PLASTIC with BILINEAR (perfectly plastic) / SWIFT / RAMBERG-OSGOOD hardening, CG + SSOR 1e-12) run through the
REFERENCE routines (oracle/_ref/ref_nl).  Deck data typed from the reference's files.  Build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import refrun                                    # noqa: E402

coord = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], dtype=np.float64)
conn = np.array([[1, 2, 3, 4, 5, 6, 7, 8]], dtype=np.int32)
x0, y0, z0, x1 = [1, 4, 5, 8], [1, 2, 5, 6], [1, 2, 3, 4], [2, 3, 6, 7]
# name: (E, nu, harden, plconst, x1 displacement, CONVERG)   -- mises.cnt / swift.cnt / ramberg.cnt
DECKS = {
    "mises": (1.0e5, 0.3, 0, (1000.0, 0.0, 0.0), 0.012, 1.0e-10),
    "swift": (21.0e4, 0.3, 2, (0.04, 649.0, 0.3), 0.1, 1.0e-6),
    "ramberg": (80000.0, 0.3, 3, (0.01, 800.0, 12.0), 0.1, 1.0e-6),
}
out = dict(coord=coord, conn=conn)
for name, (E, nu, hard, pl, disp, conv) in DECKS.items():
    bn = np.array(x0 + y0 + z0 + x1, dtype=np.int32)
    bd = np.array([1] * 4 + [2] * 4 + [3] * 4 + [1] * 4, dtype=np.int32)
    bv = np.array([0.0] * 12 + [disp] * 4)
    mat = refrun.Material(E, nu, plastic=True, harden=hard, plconst=pl, nlgeom=2)
    I, R = refrun.default_params(method=1, precond=1, maxit=10000, tol=1e-12, iterlog=0, timelog=0)
    r = refrun.run_nl_steps(mat, coord, conn, bn, bd, bv, np.zeros(24), 10, 50, conv, I, R, threads=2)
    s = r["state"]
    print(name, "newton iterations", r["log"].shape[0], "max plstrain %.5f" % s["plstrain"].max(), "sigma_xx %.3f" % s["stress"][0, 0, 0])
    for k, v in dict(E=E, nu=nu, harden=hard, plconst=np.array(pl), disp=disp, converg=conv, bc_node=bn, bc_dof=bd, bc_val=bv,
                     log=r["log"], unode=r["unode"], stress=s["stress"], strain=s["strain"], plstrain=s["plstrain"],
                     fstat=s["fstat"], istat=s["istat"]).items():
        out[name + "_" + k] = v
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "nl_1elem.npz"), **out)
