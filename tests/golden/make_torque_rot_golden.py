#!/usr/bin/env python3
"""Fixtures for the two ROT_CENTER decks of the reference, examples/static/torque_rot/{rot,torque} (linear static, TYPE=361, CG + SSOR;
`!BOUNDARY, ROT_CENTER=` = a rotation prescribed about a centre node, fstr_AddBC.f90:69-160 -- the branch that READS hecMAT%B at
the centre node, which is why the hecmw_mat_ass_bc hook of the device path sits after the B assignment, INTEGRATION.md section 5;
`!CLOAD, ROT_CENTER=` = a torque spread over a node group).  The reference ships no *_correct.log for them: the deck (mesh +
control file) is copied to tests/golden/decks/torque_rot/<name>/ and 0.log of the UNMODIFIED program (oracle/_ref/fistr1_ref,
4 OpenMP threads) on that copy, with the two work-arounds of oracle/fistr1_run.py, is stored next to it as the expected output.
Run where /root/reference exists, after `python oracle/build_ref.py --only fistr1`."""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fistr1_run as f1

for name, mesh, cnt in (("rot", "rot_disp.msh", "rot_disp.cnt"), ("torque", "torque_load.msh", "torque_load.cnt")):
    sub = os.path.join("torque_rot", name)
    out = os.path.join(f1.DECKS, sub)
    os.makedirs(out, exist_ok=True)
    for f in (mesh, cnt):
        shutil.copy(os.path.join("/root/reference/examples/static/torque_rot", name, f), os.path.join(out, f))
    d = tempfile.mkdtemp(prefix=name + "_")
    r = f1.run_deck("fistr1_ref", sub, mesh, cnt, threads=4, keep=d)
    assert r["returncode"] == 0 and "FrontISTR Completed !!" in r["stdout"], r["stdout"][-2000:]
    shutil.copy(os.path.join(d, "0.log"), os.path.join(out, os.path.splitext(cnt)[0] + "_fistr1_ref_0.log"))
    shutil.rmtree(d)
    print(name, len(r["log"]), "summaries", r["log"][-1]["Node"])
