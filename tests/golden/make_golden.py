#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference (oracle/_ref, built
from /root/reference by oracle/build_ref.py) on small decks.  Run in the build
container only (the reference does not exist on the GPU box); the .npz files
are data -- inputs and the reference's outputs -- and are committed.

Decks
  cube4        4^3 unit C3D8 cube (125 nodes), E=210000 nu=0.3, z=0 clamped, x-load on top
  cube3s       3^3 cube with deterministically skewed interior nodes
  exA_A361     /root/reference/examples/static/exA/A361.msh + A300.cnt (mesh numbers are
               data read from the reference's own example; expected extrema are the
               reference's A361_correct.log)
For each deck: assembled+BC'd BSR for ELEMOPT361 = IC / Bbar / FI, first element
stiffness, and for the IC matrix the solver outputs of
{CG,BiCGSTAB} x {DIAG, SSOR(1 thread), SSOR(4 threads: RCM+multicolour), ILU(0)}.
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from frontistr_amd.mesh import CubeMesh  # noqa: E402
from oracle import refrun  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CONFIGS = [(1, 3, 1), (1, 1, 1), (1, 1, 4), (1, 10, 1), (2, 3, 1), (2, 1, 1), (2, 1, 4), (2, 10, 1)]


def parse_hecmw_msh(path):
    """Minimal HEC-MW mesh reader for the exA deck: !NODE, !ELEMENT TYPE=361, !NGROUP."""
    nodes, elems, groups = {}, [], {}
    mode, grp, gen = None, None, False
    for line in open(path):
        s = line.strip()
        if not s or s.startswith("!!") or s.startswith("#"):
            continue
        if s.startswith("!"):
            u = s.upper()
            mode, gen = None, False
            if u.startswith("!NODE"):
                mode = "node"
            elif u.startswith("!ELEMENT"):
                mode = "elem"
            elif u.startswith("!NGROUP"):
                mode = "ngrp"
                grp = re.search(r"NGRP\s*=\s*(\w+)", u).group(1)
                gen = "GENERATE" in u
                groups.setdefault(grp, [])
            continue
        f = [x.strip() for x in s.split(",") if x.strip()]
        if mode == "node":
            nodes[int(f[0])] = [float(x) for x in f[1:4]]
        elif mode == "elem":
            elems.append([int(x) for x in f[1:9]])
        elif mode == "ngrp":
            if gen:
                a, b, c = (int(x) for x in f[:3])
                groups[grp] += list(range(a, b + 1, c))
            else:
                groups[grp] += [int(x) for x in f]
    return nodes, elems, groups


def exA_deck():
    d = "/root/reference/examples/static/exA"
    nodes, elems, groups = parse_hecmw_msh(os.path.join(d, "A361.msh"))
    used = sorted({n for e in elems for n in e})          # HEC-MW drops unreferenced nodes
    lid = {g: i + 1 for i, g in enumerate(used)}
    coord = np.array([nodes[g] for g in used])
    conn = np.array([[lid[g] for g in e] for e in elems], dtype=np.int32)
    fix = [lid[g] for g in groups["FIX"] if g in lid]      # A300.cnt: !BOUNDARY FIX,1,3,0.0
    bn = np.repeat(np.array(fix, dtype=np.int32), 3)
    bd = np.tile(np.array([1, 2, 3], dtype=np.int32), len(fix))
    bv = np.zeros(bn.size)
    load = np.zeros(3 * len(used))
    for g in groups["CL1"]:                                # !CLOAD CL1,3,-1.0
        load[3 * (lid[g] - 1) + 2] = -1.0
    # material from the .msh: 4000., 0.3
    expect = {}
    for line in open(os.path.join(d, "A361_correct.log")):
        m = re.match(r"\s*//(U[123])\s+([-0-9.E+]+)\s+\d+\s+([-0-9.E+]+)\s+\d+", line)
        if m and m.group(1) not in expect:
            expect[m.group(1)] = (float(m.group(2)), float(m.group(3)))
    return coord, conn, (bn, bd, bv), load, 4000.0, 0.3, expect, np.array(used)


def pack_matrix(prefix, A, out):
    for k in ("indexL", "itemL", "indexU", "itemU", "D", "AL", "AU", "B"):
        out[prefix + k] = getattr(A, k)
    out[prefix + "N"] = np.int32(A.N)
    out[prefix + "NP"] = np.int32(A.NP)


def make_deck(name, coord, conn, bc, load, E, nu, extra=None):
    out = dict(coord=coord, conn=conn, bc_node=bc[0], bc_dof=bc[1], bc_val=bc[2], load=load,
               E=np.float64(E), nu=np.float64(nu))
    A_ic = None
    for eo, tag in ((1, "ic_"), (2, "bbar_"), (3, "fi_")):
        A, ke, _ = refrun.run_fem(coord, conn, E, nu, *bc, load, elemopt=eo)
        pack_matrix(tag, A, out)
        out[tag + "ke"] = ke
        if eo == 1:
            A_ic = A
    for meth, pc, thr in CONFIGS:
        I, R = refrun.default_params(method=meth, precond=pc)
        r = refrun.run_solve(A_ic, I, R, threads=thr)
        tag = "sol_m%d_p%d_t%d_" % (meth, pc, thr)
        out[tag + "iter"] = np.int32(r["iter"])
        out[tag + "hist"] = np.array([h[1] for h in r["history"]])
        out[tag + "X"] = r["X"]
        out[tag + "Iarray"] = r["Iarray"]
        out[tag + "rel_resid"] = np.float64(r["rel_resid"])
        print(name, r["banner"], "threads", thr, "iter", r["iter"])
    if extra:
        out.update(extra)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)


def main():
    m = CubeMesh(4)
    make_deck("cube4", m.coord, m.conn, m.dirichlet(), m.load(), 210000.0, 0.3)
    m = CubeMesh(3, skew=0.15)
    make_deck("cube3s", m.coord, m.conn, m.dirichlet(), m.load(), 210000.0, 0.3)
    coord, conn, bc, load, E, nu, expect, gid = exA_deck()
    extra = {"expect_" + k: np.array(v) for k, v in expect.items()}
    extra["global_id"] = gid
    make_deck("exA_A361", coord, conn, bc, load, E, nu, extra)


if __name__ == "__main__":
    main()
