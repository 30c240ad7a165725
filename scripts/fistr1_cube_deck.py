#!/usr/bin/env python3
"""A synthetic NLSTATIC deck for fistr1 at any size: unit-spacing cube of n^3 C3D8 (TYPE=361) elements, z = 0 clamped, the top
face pulled by 0.5 % in z with a little shear, multilinear Mises plasticity of tutorial/05_plastic_cylinder (necking.cnt), updated
Lagrange (NLSTATIC default), SUBSTEPS sub-steps, CG + SSOR (or what --solver says) with TIMELOG.  Writes <dir>/cube.msh, cube.cnt,
hecmw_ctrl.dat (with the restart work-around of oracle/fistr1_run.py).  Used to time fistr1_hip end to end: device assembly
(default) against HECMW_GPU_ASSEMBLY=0 (scripts/r3/fistr1_big.sh).  usage: fistr1_cube_deck.py DIR N [SUBSTEPS] [METHOD] [PRECOND] [STRAIN]"""
import os
import sys

import numpy as np

linear = "--linear" in sys.argv      # !SOLUTION, TYPE=STATIC: bench.py's workload (z = 0 clamped, unit load in x on every top node, E = 210000, nu = 0.3)
if linear:
    sys.argv.remove("--linear")
form361 = None                       # --form361 FI|BBAR|IC: `!SECTION, SECNUM=1, FORM361=...` (fstr_ctrl_common.f90:303-320); default: the program's (IC)
if "--form361" in sys.argv:
    k = sys.argv.index("--form361"); form361 = sys.argv[k + 1]; del sys.argv[k:k + 2]
d, n = sys.argv[1], int(sys.argv[2])
nsub = int(sys.argv[3]) if len(sys.argv) > 3 else 2
method = sys.argv[4] if len(sys.argv) > 4 else "CG"
precond = sys.argv[5] if len(sys.argv) > 5 else "1"
strain = float(sys.argv[6]) if len(sys.argv) > 6 else 0.005      # top-face stretch (yield strain of the table: 0.0022)
os.makedirs(d, exist_ok=True)
m = n + 1
k, j, i = np.meshgrid(np.arange(m), np.arange(m), np.arange(m), indexing="ij")
nid = (1 + i + m * (j + m * k)).ravel()
xyz = np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1).astype(float)
ek, ej, ei = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
n0 = (1 + ei + m * (ej + m * ek)).ravel()
conn = np.stack([n0, n0 + 1, n0 + 1 + m, n0 + m, n0 + m * m, n0 + 1 + m * m, n0 + 1 + m + m * m, n0 + m + m * m], axis=1)
with open(os.path.join(d, "cube.msh"), "w") as fh:
    fh.write("!HEADER\n synthetic cube, frontistr_amd scripts/fistr1_cube_deck.py\n!NODE\n")
    np.savetxt(fh, np.column_stack([nid, xyz]), fmt="%d,%.1f,%.1f,%.1f")
    fh.write("!ELEMENT,TYPE=361,EGRP=E1\n")
    np.savetxt(fh, np.column_stack([np.arange(1, conn.shape[0] + 1), conn]), fmt="%d", delimiter=",")
    fh.write("!MATERIAL,NAME=MAT1,ITEM=1\n!ITEM=1,SUBITEM=2\n 206900.0,0.29\n!SECTION,TYPE=SOLID,EGRP=E1,MATERIAL=MAT1\n")
    fh.write("!NGROUP, NGRP=FIX, GENERATE\n 1,%d,1\n" % (m * m))
    fh.write("!NGROUP, NGRP=TOP, GENERATE\n %d,%d,1\n!END\n" % (m * m * n + 1, m * m * m))
if linear:
    with open(os.path.join(d, "cube.cnt"), "w") as fh:
        fh.write("""!VERSION
 3
!SOLUTION, TYPE=STATIC
!WRITE,RESULT,FREQUENCY=100000
!BOUNDARY
 FIX, 1, 3, 0.0
!CLOAD
 TOP, 1, 1.0
!MATERIAL, NAME=MAT1
!ELASTIC
 210000.0, 0.3
%s!RESTART, FREQUENCY=100000
!SOLVER,METHOD=%s,PRECOND=%s,ITERLOG=NO,TIMELOG=YES
 10000, 1
 1.0e-8, 1.0, 0.0
!END
""" % ("!SECTION, SECNUM=1, FORM361=%s\n" % form361 if form361 else "", method, precond))
with open(os.path.join(d, "cube.cnt"), "a" if linear else "w") as fh:
    if not linear:
      fh.write("""!VERSION
 3
!SOLUTION, TYPE=NLSTATIC
!WRITE,RESULT,FREQUENCY=100000
!BOUNDARY, GRPID=1
 FIX, 1, 3, 0.0
 TOP, 3, 3, %g
 TOP, 1, 1, %g
!STEP, SUBSTEPS=%d, CONVERG=1.0e-3
 BOUNDARY, 1
!MATERIAL, NAME=MAT1
!ELASTIC
 206900.0, 0.29
!PLASTIC, YIELD=MISES, HARDEN=MULTILINEAR
 450.0, 0.0
 608.0, 0.05
 679.0, 0.1
 732.0, 0.2
 752.0, 0.3
 766.0, 0.4
 780.0, 0.5
!RESTART, FREQUENCY=100000
!SOLVER,METHOD=%s,PRECOND=%s,ITERLOG=NO,TIMELOG=YES
 5000, 1
 1.0e-8, 1.0, 0.0
!END
""" % (strain * n, 0.2 * strain * n, nsub, method, precond))
with open(os.path.join(d, "hecmw_ctrl.dat"), "w") as fh:
    fh.write("!MESH, NAME=fstrMSH,TYPE=HECMW-ENTIRE\n cube.msh\n!CONTROL,NAME=fstrCNT\n cube.cnt\n"
             "!RESULT,NAME=fstrRES,IO=OUT\n out.res\n!RESTART,NAME=restart_out,IO=OUT\n out.restart\n")
print("wrote", d, "nodes", m ** 3, "dof", 3 * m ** 3, "elements", n ** 3)
