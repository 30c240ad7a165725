"""Fresh-process check of the arena hypothesis (scripts/r4/placement_probe.py found: the SpMV's speed class follows the PHYSICAL
contiguity of its value array; inside one 32 GiB hipMalloc every offset is the fast class).  One process = one line: the product on
the value array as the library placed it, then the same array moved into a 32 GiB arena at two offsets."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("FX_TUNE_PLACEMENT", "0")
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh

L = hip.lib()
mesh = CubeMesh(int(os.environ.get("AB_N", "149")))
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
ctx = hip.SolverContext()
ctx.upload(m, what=hip.FX_UP_PROFILE)
ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
m.Iarray[1] = 1
m.Iarray[2] = 1
ctx.precond_setup(m)


def ms(kind=0):
    v = C.c_float(0)
    hip._chk(L.fx_debug_spmv_ms(ctx.h, kind, 2, 1, 1, 10, C.byref(v)))
    return v.value


def replace(what, how, arg=0):
    a = C.c_uint64(0)
    L.fx_debug_replace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_uint64)]
    hip._chk(L.fx_debug_replace(ctx.h, what, how, arg, C.byref(a)))


out = ["as placed %.4f (apply %.4f)" % (ms(), ctx.precond_apply_ms(5))]
for off in (0, 7000, 20000):
    replace(0, 3, off)
    out.append("arena+%d %.4f" % (off, ms()))
print("  ".join(out), flush=True)
