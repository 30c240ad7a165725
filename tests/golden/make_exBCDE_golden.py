"""tests/golden/ex{B,C,D,E}_361.npz: the reference's example decks examples/static/exB..exE (40-element TYPE=361 beam
under face pressure / body force / gravity / centrifugal load; X300.cnt: !BOUNDARY FIX 1-3 = 0, CG + DIAG 1e-8).
Mesh, groups and material are read from the reference's files; the nodal load vector comes from the REFERENCE's own
DL_C3 (oracle/_ref/ref_load; load assembly is outside the hot path); the known answers are the displacement extrema of
X361_correct.log (the reference harness compares them at 1e-4 absolute).  Run in the build container only."""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import refrun                                    # noqa: E402

ROOT = "/root/reference/examples/static"
HERE = os.path.dirname(os.path.abspath(__file__))
# (deck, msh, log, DLOAD cards of the .cnt: (element group, ltype, params(0:6)))
DECKS = [
    ("exB_361", "exB/B361.msh", "exB/B361_correct.log", [("B360P2", 20, [1.0, 0, 0, 0, 0, 0, 0])]),                 # B360P2, P2, 1.0
    ("exC_361", "exC/C361.msh", "exC/C361_correct.log", [("ALL", 3, [7.85e-6, 0, 0, 0, 0, 0, 0])]),                  # ALL, BZ, 7.85E-6
    ("exD_361", "exD/D361.msh", "exD/D361_correct.log", [("ALL", 4, [9800.0, 0.0, 0.0, -1.0, 0, 0, 0])]),            # ALL, GRAV, 9800, 0,0,-1
    ("exE_361", "exE/E361.msh", "exE/E361_correct.log", [("ALL", 5, [6283.1852, 0.0, 0.5, 0.5, 0.0, 0.5, 1.0])]),    # ALL, CENT, ...
    # F300.cnt: !REFTEMP 20 ; !TEMPERATURE ALL, 120 ; expansion 1e-5 (ITEM=3 of the mesh material): thermal load of the IC element
    ("exF_361", "exF/F361.msh", "exF/F361_correct.log", [("ALL", -1, [120.0, 20.0, None, None, None, 20.0, 0])]),
]


def parse_msh(path):
    nodes, elems, eids, ngrp, egrp, mat = [], [], [], {}, {}, []
    mode, name, gen = None, None, False
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#") or line.startswith("!!"):
            continue
        if line.startswith("!"):
            head = [t.strip() for t in line.split(",")]
            key = head[0].upper()
            opts = {t.split("=")[0].strip().upper(): (t.split("=")[1].strip() if "=" in t else True) for t in head[1:]}
            gen = "GENERATE" in opts
            if key == "!NODE": mode = "node"
            elif key == "!ELEMENT":
                assert opts["TYPE"] == "361"; mode = "elem"
            elif key == "!NGROUP": mode, name = "ngrp", opts["NGRP"]; ngrp.setdefault(name, [])
            elif key == "!EGROUP": mode, name = "egrp", opts["EGRP"]; egrp.setdefault(name, [])
            elif key.startswith("!ITEM"): mode = "item"
            else: mode = None
            continue
        v = line.replace(",", " ").split()
        if mode == "node": nodes.append((int(v[0]), float(v[1]), float(v[2]), float(v[3])))
        elif mode == "elem": eids.append(int(v[0])); elems.append([int(t) for t in v[1:9]])
        elif mode in ("ngrp", "egrp"):
            tgt = ngrp if mode == "ngrp" else egrp
            if gen: tgt[name] += list(range(int(v[0]), int(v[1]) + 1, int(v[2])))
            else: tgt[name] += [int(t) for t in v]
        elif mode == "item": mat.append([float(t) for t in v])
    used = {t for e in elems for t in e}                      # the entire-mesh reader drops nodes no element uses
    nodes = [n for n in nodes if n[0] in used]                # (post_remove_unused_node, hecmw_io_mesh.c:3039) and numbers
    nid = {n[0]: k + 1 for k, n in enumerate(nodes)}          # the rest in file order
    eid = {e: k + 1 for k, e in enumerate(eids)}
    coord = np.array([n[1:] for n in nodes], dtype=np.float64)
    conn = np.array([[nid[t] for t in e] for e in elems], dtype=np.int32)
    ngrp = {k: np.array([nid[t] for t in v if t in nid], dtype=np.int32) for k, v in ngrp.items()}
    egrp = {k: np.array([eid[t] for t in v if t in eid], dtype=np.int32) for k, v in egrp.items()}
    egrp["ALL"] = np.arange(1, len(elems) + 1, dtype=np.int32)
    return coord, conn, ngrp, egrp, mat


def parse_log(path):
    out, glob = {}, False
    for line in open(path):
        if "Global Summary" in line: glob = True
        m = re.match(r"\s*//(U[123])\s+([-0-9.E+]+)\s+([-0-9.E+]+)\s*$", line)
        if m and glob: out[m.group(1)] = (float(m.group(2)), float(m.group(3)))
    return np.array([out["U1"], out["U2"], out["U3"]])


def ref_load(coord, conn, loads):
    exe = os.path.join(refrun.REFDIR, "ref_load")
    with tempfile.TemporaryDirectory() as td:
        fin, fout = os.path.join(td, "in.bin"), os.path.join(td, "out.bin")
        with open(fin, "wb") as f:
            np.array([1179208772, coord.shape[0], conn.shape[0], len(loads)], dtype=np.int32).tofile(f)
            coord.tofile(f); conn.tofile(f)
            for ltype, params, rho, els in loads:
                np.array([ltype, len(els)], dtype=np.int32).tofile(f)
                np.array(list(params) + [rho], dtype=np.float64).tofile(f)
                np.ascontiguousarray(els, dtype=np.int32).tofile(f)
        subprocess.run([exe, fin, fout], check=True)
        return np.fromfile(fout, dtype=np.float64)


for name, msh, log, dloads in DECKS:
    coord, conn, ngrp, egrp, mat = parse_msh(os.path.join(ROOT, msh))
    E, nu = mat[0][0], mat[0][1]
    rho = mat[1][0] if len(mat) > 1 else 0.0
    fix = ngrp["FIX"]
    bc_node = np.repeat(fix, 3).astype(np.int32)
    bc_dof = np.tile(np.array([1, 2, 3], dtype=np.int32), fix.size)
    alpha = mat[2][0] if len(mat) > 2 else 0.0
    dloads = [(g_, lt, [alpha if k == 2 and v is None else E if k == 3 and v is None else nu if k == 4 and v is None else v
                        for k, v in enumerate(p)]) for g_, lt, p in dloads]
    load = ref_load(coord, conn, [(lt, p, rho, egrp[g]) for g, lt, p in dloads])
    expect = parse_log(os.path.join(ROOT, log))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), coord=coord, conn=conn, bc_node=bc_node, bc_dof=bc_dof,
                        bc_val=np.zeros(bc_node.size), load=load, E=E, nu=nu, rho=rho, expect=expect)
    print(name, coord.shape, conn.shape, "E", E, "nu", nu, "rho", rho, "|load|", np.abs(load).sum(), "U3", expect[2])
