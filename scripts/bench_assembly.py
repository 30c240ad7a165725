"""Device assembly (fstr_StiffMatrix + fstr_AddBC) on the n^3-element cube: ms per call for ELEMOPT361 = IC / B-bar / FI.
FX_ASM_ATOMIC=1 selects the single-launch atomic scatter instead of the coloured one.  Usage: python scripts/bench_assembly.py [n]"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from frontistr_amd import hecmw as hip          # noqa: E402
if os.environ.get("FX_LIBPATH"):
    hip.LIBPATH = os.environ["FX_LIBPATH"]      # timing experiments: a library built with -DFXA_EXP_*
from frontistr_amd.mesh import CubeMesh          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 149
mesh = CubeMesh(n)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
ctx = hip.SolverContext()
ctx.upload(m, hm, what=hip.FX_UP_PROFILE)
out = {"n_elem": int(mesh.conn.shape[0]), "dof": 3 * mesh.n_node, "scatter": "atomic" if os.environ.get("FX_ASM_ATOMIC", "0") not in ("", "0") else "coloured"}
load, bc = mesh.load(), mesh.dirichlet()
for eo, name in ((1, "ic"), (2, "bbar"), (3, "fi")):
    ms = [ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=eo, load=load, bc=bc) for _ in range(3)]
    out[name + "_ms"] = [round(v, 2) for v in ms]
print(json.dumps(out))
