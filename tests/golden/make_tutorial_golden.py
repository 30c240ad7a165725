#!/usr/bin/env python3
"""Fixtures for the nonlinear tutorials the reference ships WITHOUT a *_correct.log: tutorial/03_hyperelastic_cylinder
(Mooney-Rivlin), 06_plastic_can (Drucker-Prager, TYPE=342), 07_viscoelastic_cylinder, 08_creep_cylinder (Norton).  The deck
(mesh + control file: data of the reference's tutorial) is copied to tests/golden/decks/<name>/, and 0.log + FSTR.sta of the
UNMODIFIED program (oracle/_ref/fistr1_ref, 4 OpenMP threads = the multicolour SSOR the GPU path reproduces) on that copy,
with the two work-arounds of oracle/fistr1_run.py, are stored next to it as the expected output.
Run where /root/reference exists, after `python oracle/build_ref.py --only fistr1`."""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fistr1_run as f1

DECKS = [("t03", "03_hyperelastic_cylinder", "cylinder.msh", "cylinder.cnt"),
         ("t06", "06_plastic_can", "can.msh", "can.cnt"),
         ("t07", "07_viscoelastic_cylinder", "cylinder.msh", "cylinder.cnt"),
         ("t08", "08_creep_cylinder", "cylinder.msh", "cylinder.cnt")]
for name, src, mesh, cnt in DECKS:
    out = os.path.join(f1.DECKS, name)
    os.makedirs(out, exist_ok=True)
    for f in (mesh, cnt):
        shutil.copy(os.path.join("/root/reference/tutorial", src, f), os.path.join(out, f))
    d = tempfile.mkdtemp(prefix=name + "_")
    r = f1.run_deck("fistr1_ref", name, mesh, cnt, threads=4, keep=d)
    assert r["returncode"] == 0 and "FrontISTR Completed !!" in r["stdout"], r["stdout"][-2000:]
    stem = os.path.splitext(cnt)[0]
    shutil.copy(os.path.join(d, "0.log"), os.path.join(out, stem + "_fistr1_ref_0.log"))
    shutil.copy(os.path.join(d, "FSTR.sta"), os.path.join(out, stem + "_fistr1_ref_FSTR.sta"))
    shutil.rmtree(d)
    print(name, len(r["log"]), "summaries, Newton iterations", [x[3] for x in r["sta"]])
