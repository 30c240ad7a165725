#!/usr/bin/env python3
"""Expected outputs of tutorial/05_plastic_cylinder, which ships no *_correct.log: 0.log and FSTR.sta of the UNMODIFIED
reference program (oracle/_ref/fistr1_ref, 4 OpenMP threads = the multicolour SSOR the GPU path reproduces) on the committed
copy of the deck with the two work-arounds of oracle/fistr1_run.py.  The run stops by itself at sub-step 9 (MAXITER, SURVEY
section 0) after 8 converged sub-steps: 36, 5, 5, 5, 5, 5, 5, 5 Newton iterations.
Run where /root/reference exists, after `python oracle/build_ref.py --only fistr1`."""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fistr1_run as f1

d = tempfile.mkdtemp(prefix="t05_")
r = f1.run_deck("fistr1_ref", "t05", "necking.msh", "necking.cnt", threads=4, keep=d)
assert [x[3] for x in r["sta"][:8]] == [36, 5, 5, 5, 5, 5, 5, 5], r["sta"]
out = os.path.join(f1.DECKS, "t05")
shutil.copy(os.path.join(d, "0.log"), os.path.join(out, "necking_fistr1_ref_0.log"))
shutil.copy(os.path.join(d, "FSTR.sta"), os.path.join(out, "necking_fistr1_ref_FSTR.sta"))
shutil.rmtree(d)
print("wrote", out, len(r["log"]), "summaries")
