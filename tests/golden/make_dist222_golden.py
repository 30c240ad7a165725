#!/usr/bin/env python3
"""Iteration counts of the CPU oracle on bench.py's 2x2x2 decomposition (cube_subdomain(6, (2, 2, 2), r), 8 gloo ranks):
the -m gpu test of the same decomposition compares with these (it may not start 8 more processes beside the GPU contexts:
the box admits 6); tests/test_distributed.py re-runs the oracle and checks that this file still holds."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pathlib import Path                               # noqa: E402
from test_distributed import run_world                 # noqa: E402

out = {}
for meth, pc in ((1, 1), (2, 10)):
    with tempfile.TemporaryDirectory() as d:
        res = run_world("oracle", 8, 6, meth, pc, Path(d))
        assert len({int(r["it"]) for r in res}) == 1 and all(int(r["code"]) == 0 for r in res)
        out["m6_meth%d_pc%d" % (meth, pc)] = {"iter": int(res[0]["it"]), "history_head": [float(v) for v in res[0]["hist"][:10]]}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "dist_2x2x2_oracle.json"), "w"), indent=1)
print(out)
