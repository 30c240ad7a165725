#!/usr/bin/env python3
"""Fixture copies of the reference's regression decks examples/static/exA ... exG and FbarElement: mesh, control file and the
shipped *_correct.log of every model (data files of the reference's own test suite, examples/static/test_static.sh; the
pairing mesh <-> control file is the one of the per-directory test_ex?.sh scripts: X2nn -> X200.cnt, X3nn -> X300.cnt,
X7nn -> X700.cnt, or the model's own .cnt).  Writes tests/golden/decks/static/<dir>/ and the manifest
tests/golden/decks/static/manifest.json = [[dir, model, mesh, control file, NDOF], ...].  Run where /root/reference exists."""
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = "/root/reference/examples/static"
OUT = os.path.join(ROOT, "tests", "golden", "decks", "static")

manifest = []
for sub in ["exA", "exB", "exC", "exD", "exE", "exF", "exG", "FbarElement"]:
    d = os.path.join(SRC, sub)
    os.makedirs(os.path.join(OUT, sub), exist_ok=True)
    for log in sorted(glob.glob(d + "/*_correct.log")):
        model = os.path.basename(log)[:-len("_correct.log")]
        mesh, cnt = model + ".msh", model + ".cnt"
        if not os.path.exists(os.path.join(d, cnt)):
            m = re.match(r"[A-Z](\d)\d\d$", model)
            cands = glob.glob(d + "/?%s00.cnt" % m.group(1))
            assert len(cands) == 1, (sub, model, cands)
            cnt = os.path.basename(cands[0])
        m = re.match(r"[A-Z]([237])\d\d$", model)
        ndof = {"2": 2, "3": 3, "7": 6}[m.group(1)] if m else 3
        for f in (mesh, cnt, model + "_correct.log"):
            shutil.copy(os.path.join(d, f), os.path.join(OUT, sub, f))
        manifest.append([sub, model, mesh, cnt, ndof])
with open(os.path.join(OUT, "manifest.json"), "w") as fh:
    fh.write("[\n" + ",\n".join(json.dumps(x) for x in manifest) + "\n]\n")
print(len(manifest), "models ->", OUT)
