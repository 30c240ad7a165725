#!/usr/bin/env python3
"""Fixture copies of the reference's regression decks examples/static/exA ... exG and FbarElement: mesh, control file and the
shipped *_correct.log of every model (data files of the reference's own test suite, examples/static/test_static.sh; the
pairing mesh <-> control file is the one of the per-directory test_ex?.sh scripts: X2nn -> X200.cnt, X3nn -> X300.cnt,
X7nn -> X700.cnt, or the model's own .cnt).  Writes tests/golden/decks/static/<dir>/ and the manifest
tests/golden/decks/static/manifest.json = [[dir, model, mesh, control file, NDOF], ...].  Run where /root/reference exists.
The heat suite (examples/heat/exM ... exT, test_heat.sh) goes to tests/golden/decks/heat/ the same way."""
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

# examples/heat/exM ... exT (steady and transient heat conduction, NDOF = 1): pairing from the test_ex?.sh scripts
# (`${prg} ${test_log} <model> <control>`), judged on the Maximum / Minimum Temperature lines of 0.log (test_heat_sub.sh)
HSRC = "/root/reference/examples/heat"
HOUT = os.path.join(ROOT, "tests", "golden", "decks", "heat")
hman = []
for sh in sorted(glob.glob(HSRC + "/*/test_ex*.sh")):
    d = os.path.dirname(sh)
    sub = os.path.basename(d)
    for line in open(sh):
        m = re.search(r"prg} \$\{test_log} (\S+) (\S+)", line)
        if not m:
            continue
        model, cnt = m.group(1), m.group(2) + ".cnt"
        os.makedirs(os.path.join(HOUT, sub), exist_ok=True)
        for f in (model + ".msh", cnt, model + "_correct.log"):
            shutil.copy(os.path.join(d, f), os.path.join(HOUT, sub, f))
        hman.append([sub, model, model + ".msh", cnt, 1])
with open(os.path.join(HOUT, "manifest.json"), "w") as fh:
    fh.write("[\n" + ",\n".join(json.dumps(x) for x in hman) + "\n]\n")
print(len(hman), "models ->", HOUT)
