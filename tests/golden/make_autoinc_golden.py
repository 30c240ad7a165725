#!/usr/bin/env python3
"""Fixtures for automatic incrementation with cutback (fstr_Cutback.f90:108-198, fstr_solve_NLGEOM.f90:85-242):
  autoinc      examples/static/autoinc (C3D8beam: NLSTATIC, elastic, `!STEP ... MAXITER=3, INC_TYPE=AUTO, AUTOINCPARAM=AP1` with
               `!AUTOINC_PARAM`: Newton runs out of iterations, the increment is cut back) -- the reference's own deck;
  t05_autoinc  tutorial/05_plastic_cylinder (configs[4]'s deck) with the `!STEP` card switched to INC_TYPE=AUTO and an
               `!AUTOINC_PARAM` card: ten sub-steps of 0.025, then Newton runs into MAXITER at t = 0.25 seven times in a row -- seven
               cutbacks (fstr_cutback_load) -- before an increment of 2e-4 converges and the run goes on; it ends at its SUBSTEPS = 40
               bound ("Number of substeps reached max number", as the reference's own autoinc deck does).
The reference ships no *_correct.log for either: 0.log and FSTR.sta of the UNMODIFIED program (oracle/_ref/fistr1_ref, 4 OpenMP
threads) on the committed copy, with the work-arounds of oracle/fistr1_run.py, are stored next to the deck as the expected output.
Run where /root/reference exists, after `python oracle/build_ref.py --only fistr1`."""
import os
import re
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fistr1_run as f1

out = os.path.join(f1.DECKS, "autoinc")
os.makedirs(out, exist_ok=True)
for f in ("C3D8beam.msh", "C3D8beam.cnt"):
    shutil.copy(os.path.join("/root/reference/examples/static/autoinc", f), os.path.join(out, f))

out5 = os.path.join(f1.DECKS, "t05_autoinc")
os.makedirs(out5, exist_ok=True)
shutil.copy(os.path.join(f1.DECKS, "t05", "necking.msh"), os.path.join(out5, "necking.msh"))
cnt = open(os.path.join(f1.DECKS, "t05", "necking.cnt")).read()
cnt = cnt.replace("!STEP, SUBSTEPS=40, CONVERG=1.0e-3\n",
                  "!AUTOINC_PARAM, NAME=AP1\n 0.25, 10, 50, 10, 1\n 1.25, 10, 1, 2, 2\n 0.5, 8\n"
                  "!STEP, SUBSTEPS=40, CONVERG=1.0e-3, MAXITER=50, INC_TYPE=AUTO, AUTOINCPARAM=AP1\n 0.025, 0.3, 1.0e-6, 0.025\n")
assert "INC_TYPE=AUTO" in cnt
open(os.path.join(out5, "necking_autoinc.cnt"), "w").write(cnt)

for name, mesh, c in (("autoinc", "C3D8beam.msh", "C3D8beam.cnt"), ("t05_autoinc", "necking.msh", "necking_autoinc.cnt")):
    d = tempfile.mkdtemp(prefix=name + "_")
    r = f1.run_deck("fistr1_ref", name, mesh, c, threads=4, keep=d)
    # (the reference's autoinc deck ends by design at its SUBSTEPS bound: "Number of substeps reached max number", exit code 0)
    assert r["returncode"] == 0 and ("FrontISTR Completed !!" in r["stdout"] or "Number of substeps reached max number" in r["stdout"]), r["stdout"][-3000:]
    stem = os.path.splitext(c)[0]
    shutil.copy(os.path.join(d, "0.log"), os.path.join(f1.DECKS, name, stem + "_fistr1_ref_0.log"))
    shutil.copy(os.path.join(d, "FSTR.sta"), os.path.join(f1.DECKS, name, stem + "_fistr1_ref_FSTR.sta"))
    with open(os.path.join(f1.DECKS, name, stem + "_fistr1_ref_steps.txt"), "w") as fh:      # the sub-step / increment / Newton lines of stdout
        fh.write("\n".join(f1.step_lines(r["stdout"])) + "\n")
    ncut = len(re.findall(r"State has been restored", r["stdout"]))
    print(name, len(r["log"]), "summaries;", len(r["sta"]), "sub-step rows, statuses", [x[2] for x in r["sta"]], "; cutbacks", ncut)
    shutil.rmtree(d)


# examples/static/restart2/case02_resume: write a restart file every sub-step, stop after three, resume from the fourth (`!RESTART,
# FREQUENCY=-1`) -- the continuation takes the device path with the history read from the file (the restart guard of
# fsd_eligible was dead code).  Golden: FSTR.sta, step lines and 0.log of the RESUMED run of the unmodified program.
outr = os.path.join(f1.DECKS, "restart2")
os.makedirs(outr, exist_ok=True)
for f in ("C3D8beam.msh", "C3D8beam.cnt", "C3D8beam_res.cnt"):
    shutil.copy(os.path.join("/root/reference/examples/static/restart2/case02_resume", f), os.path.join(outr, f))
d = tempfile.mkdtemp(prefix="restart2_")
first, res = f1.run_restart_pair("fistr1_ref", outr, d, threads=4)
assert first["returncode"] == 0 and res["returncode"] == 0, (first["stdout"][-2000:], res["stdout"][-2000:])
assert "FrontISTR Completed !!" in res["stdout"] or "Number of substeps reached max number" in res["stdout"], res["stdout"][-3000:]   # ends at its SUBSTEPS = 100 bound
shutil.copy(os.path.join(d, "0.log"), os.path.join(outr, "resumed_fistr1_ref_0.log"))
shutil.copy(os.path.join(d, "FSTR.sta"), os.path.join(outr, "resumed_fistr1_ref_FSTR.sta"))
with open(os.path.join(outr, "resumed_fistr1_ref_steps.txt"), "w") as fh:
    fh.write("\n".join(f1.step_lines(res["stdout"])) + "\n")
print("restart2: first run", [x[2] for x in first["sta"]], "; resumed", len(res["sta"]), "rows", [x[2] for x in res["sta"]][:12], "...", len(res["log"]), "summaries")
shutil.rmtree(d)
