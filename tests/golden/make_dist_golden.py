#!/usr/bin/env python3
"""Generate tests/golden/dist_cube4/ and tests/golden/dist_cube6x8/: the synthetic 4^3 (6^3) cube written as a HEC-MW mesh
and split into 4 (8) node-based subdomains by the REFERENCE partitioner (oracle/_ref/hecmw_part1, built from
/root/reference/hecmw1/tools/partitioner by oracle/build_ref.py; METHOD=RCB since METIS is absent).
The HECMW-DIST files are data produced by the reference; committed as fixtures."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from frontistr_amd.mesh import CubeMesh  # noqa: E402


def write_msh(path, m):
    with open(path, "w") as f:
        f.write("!HEADER\n synthetic cube\n!NODE\n")
        for i, c in enumerate(m.coord):
            f.write("%d, %.6f, %.6f, %.6f\n" % (i + 1, c[0], c[1], c[2]))
        f.write("!ELEMENT, TYPE=361, EGRP=E1\n")
        for e, c in enumerate(m.conn):
            f.write("%d, %s\n" % (e + 1, ", ".join(str(x) for x in c)))
        f.write("!MATERIAL, NAME=M1, ITEM=1\n!ITEM=1, SUBITEM=2\n 210000.0, 0.3\n")
        f.write("!SECTION, TYPE=SOLID, EGRP=E1, MATERIAL=M1\n 1.0\n")
        f.write("!NGROUP, NGRP=FIX\n" + "\n".join(str(x) for x in m.bottom_nodes) + "\n")
        f.write("!NGROUP, NGRP=TOP\n" + "\n".join(str(x) for x in m.top_nodes) + "\n!END\n")


def main(n=4, ndom=4, name="dist_cube4", axes="x,y"):
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), name)
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(ROOT, "oracle", "_ref", "hecmw_part1")
    with tempfile.TemporaryDirectory() as td:
        write_msh(os.path.join(td, "cube.msh"), CubeMesh(n))
        open(os.path.join(td, "hecmw_ctrl.dat"), "w").write(
            "!MESH, NAME=part_in, TYPE=HECMW-ENTIRE\n cube.msh\n!MESH, NAME=part_out, TYPE=HECMW-DIST\n cube_p\n")
        open(os.path.join(td, "hecmw_part_ctrl.dat"), "w").write(
            "!PARTITION,TYPE=NODE-BASED,METHOD=RCB,DOMAIN=%d\n %s\n" % (ndom, axes))
        subprocess.run([exe], cwd=td, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        for r in range(ndom):
            shutil.copy(os.path.join(td, "cube_p.%d" % r), os.path.join(out, "cube_p.%d" % r))
        shutil.copy(os.path.join(td, "hecmw_part_ctrl.dat"), out)
    print("wrote", out, os.listdir(out))


if __name__ == "__main__":
    main()
    # configs[3] stand-in (tutorial/02's hinge.msh is absent from the mount): a 6^3-element cube in EIGHT subdomains
    main(n=6, ndom=8, name="dist_cube6x8", axes="x,y,z")
