#!/usr/bin/env python3
"""Fixture for the stress update of linear static decks: the reference's own UpdateST_C3D8IC
(static_LIB_3dIC.f90:220-455), Update_C3D8Bbar (static_LIB_C3D8.f90:203-547) and UPDATE_C3 (static_LIB_3d.f90:516-837) called
element by element as fstr_UpdateNewton does (oracle/ref_update_driver.f90 -> oracle/_ref/ref_update) on a skewed 3^3-element
cube with two materials and a random displacement state (unode + dunode).  Stored: inputs and the reference's strain / stress at
the 8 quadrature points of every element and QFORCE, for ELEMOPT361 = IC, BBAR, FI -> tests/golden/update_linear.npz.
Run where /root/reference exists, after `python oracle/build_ref.py --only update`."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from frontistr_amd.mesh import CubeMesh
from oracle import refrun

m = CubeMesh(3, skew=0.2)
rng = np.random.default_rng(20261005)
unode = 1e-3 * rng.standard_normal(3 * m.n_node)
dunode = 3e-4 * rng.standard_normal(3 * m.n_node)
E = np.array([210000.0, 70000.0])
nu = np.array([0.3, 0.33])
elem_mat = (1 + (np.arange(m.conn.shape[0]) % 2)).astype(np.int32)
out = dict(coord=m.coord, conn=m.conn.astype(np.int32), unode=unode, dunode=dunode, E=E, nu=nu, elem_mat=elem_mat)
for eo, tag in ((1, "ic"), (2, "bbar"), (3, "fi")):
    s, t, q = refrun.run_update(eo, m.coord, m.conn, E, nu, unode, dunode, elem_mat=elem_mat)
    out[tag + "_strain"], out[tag + "_stress"], out[tag + "_qforce"] = s, t, q
    print(tag, "max |strain| %.3e  max |stress| %.3e  max |qforce| %.3e" % (np.abs(s).max(), np.abs(t).max(), np.abs(q).max()))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "update_linear.npz"), **out)
