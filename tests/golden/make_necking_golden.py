"""tests/golden/nl_necking.npz: the reference's own elastoplastic deck tutorial/05_plastic_cylinder
(necking.msh: 629 nodes, 432 TYPE=361 elements; necking.cnt: NLSTATIC, !PLASTIC YIELD=MISES HARDEN=MULTILINEAR,
40 substeps to -7.0 on LOADS, CONVERG=1e-3, CG + SSOR 1e-8) -- mesh data read from the reference's file, the first
NSUB substeps run through the REFERENCE routines (oracle/_ref/ref_nl).  Run in the build container only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle import refrun                                    # noqa: E402

DECK = "/root/reference/tutorial/05_plastic_cylinder/necking.msh"
NSUB, NSUB_TOTAL, MAX_ITER, CONVERG = 3, 40, 50, 1.0e-3


def parse_msh(path):
    nodes, elems, groups = [], [], {}
    mode, name, gen = None, None, False
    for line in open(path):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        if line.startswith("!"):
            head = [t.strip() for t in line.split(",")]
            key = head[0].upper()
            opts = {t.split("=")[0].strip().upper(): (t.split("=")[1].strip() if "=" in t else True) for t in head[1:]}
            if key == "!NODE":
                mode = "node"
            elif key == "!ELEMENT":
                assert opts["TYPE"] == "361"
                mode = "elem"
            elif key == "!NGROUP":
                mode, name, gen = "ngrp", opts["NGRP"], "GENERATE" in opts
                groups.setdefault(name, [])
            else:
                mode = None
            continue
        v = [t for t in line.replace(",", " ").split()]
        if mode == "node":
            nodes.append((int(v[0]), float(v[1]), float(v[2]), float(v[3])))
        elif mode == "elem":
            elems.append([int(t) for t in v[1:9]])
        elif mode == "ngrp":
            if gen:
                a, b, st = int(v[0]), int(v[1]), int(v[2])
                groups[name] += list(range(a, b + 1, st))
            else:
                groups[name] += [int(t) for t in v]
    ids = np.array([n[0] for n in nodes])
    assert np.array_equal(ids, np.arange(1, len(nodes) + 1))   # file order == local numbering
    coord = np.array([n[1:] for n in nodes], dtype=np.float64)
    conn = np.array(elems, dtype=np.int32)
    groups = {k: np.unique(np.array(v, dtype=np.int32)) for k, v in groups.items()}
    return coord, conn, groups


def main():
    coord, conn, g = parse_msh(DECK)
    # !BOUNDARY of necking.cnt: LOADS 3 3 -7.0 ; FIX 3 3 0 ; XSYMM 1 1 0 ; YSYMM 2 2 0
    bn = np.concatenate([g["LOADS"], g["FIX"], g["XSYMM"], g["YSYMM"]]).astype(np.int32)
    bd = np.concatenate([np.full(g["LOADS"].size, 3), np.full(g["FIX"].size, 3), np.full(g["XSYMM"].size, 1),
                         np.full(g["YSYMM"].size, 2)]).astype(np.int32)
    bv = np.concatenate([np.full(g["LOADS"].size, -7.0), np.zeros(g["FIX"].size + g["XSYMM"].size + g["YSYMM"].size)])
    table = [[450.0, 0.0], [608.0, 0.05], [679.0, 0.1], [732.0, 0.2], [752.0, 0.3], [766.0, 0.4], [780.0, 0.5]]
    mat = refrun.Material(206900.0, 0.29, plastic=True, harden=1, table=table, nlgeom=2)
    I, R = refrun.default_params(method=1, precond=1, maxit=2000, tol=1e-8, iterlog=0, timelog=0)
    # the first NSUB of the deck's 40 equal substeps == NSUB substeps to NSUB/40 of the load
    out = refrun.run_nl_steps(mat, coord, conn, bn, bd, bv * NSUB / NSUB_TOTAL, np.zeros(3 * coord.shape[0]), NSUB,
                              MAX_ITER, CONVERG, I, R, threads=2)
    s = out["state"]
    print(out["stdout"][-1500:])
    print("newton iterations", out["log"].shape[0], "max plstrain", s["plstrain"].max(), "plastic points", int(s["istat"].sum()))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "nl_necking.npz"), coord=coord, conn=conn,
                        bc_node=bn, bc_dof=bd, bc_val=bv, table=np.array(table), nsub=NSUB, nsub_total=NSUB_TOTAL,
                        max_iter=MAX_ITER, converg=CONVERG, log=out["log"], unode=out["unode"], qforce=out["qforce"],
                        stress=s["stress"], strain=s["strain"], plstrain=s["plstrain"], fstat=s["fstat"], istat=s["istat"])


if __name__ == "__main__":
    main()
