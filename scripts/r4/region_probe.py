"""Region or contiguity?  A 64 GiB arena is taken FIRST (before the library allocates anything), the system is set up as usual, and
the SpMV is timed with its value array as the library placed it, at offsets through the arena, and in a walk of fresh power-of-two
allocations (each kept, so the walk moves through VRAM)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("FX_TUNE_PLACEMENT", "0")
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh

L = hip.lib()
ARENA = int(os.environ.get("ARENA_GIB", "64"))
ctx = hip.SolverContext()
if ARENA > 0:
    hip._chk(L.fx_debug_arena(ctx.h, ARENA))
mesh = CubeMesh(int(os.environ.get("AB_N", "149")))
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
ctx.upload(m, what=hip.FX_UP_PROFILE)
ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
m.Iarray[1] = 1
m.Iarray[2] = 1
ctx.precond_setup(m)
L.fx_debug_spmv_ms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]
L.fx_debug_replace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_uint64)]


def ms(mp=0, kind=0):
    v = C.c_float(0)
    hip._chk(L.fx_debug_spmv_ms(ctx.h, kind, 2, 1, 1, 5, C.byref(v), mp))
    return v.value


def replace(what, how, arg=0):
    a = C.c_uint64(0)
    hip._chk(L.fx_debug_replace(ctx.h, what, how, arg, C.byref(a)))
    return a.value


print("as placed (pow2 after the CSR arrays)   %.4f   apply %.4f" % (ms(), ctx.precond_apply_ms(5)), flush=True)
if ARENA > 0:
    for off in range(0, (ARENA - 7) * 1024, 4096):
        a = replace(0, 3, off)
        print("arena(%d GiB, taken first) +%5d MiB   %.4f   @0x%x" % (ARENA, off, ms(), a), flush=True)
for k in range(14):
    a = replace(0, 1, 0)
    print("pow2 walk #%2d   %.4f   @0x%x" % (k, ms(), a), flush=True)
