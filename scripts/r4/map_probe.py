"""Does the SpMV's speed class come from its access pattern -- eight XCDs each streaming ONE contiguous eighth of the value array, the
eight streams a fixed 0.81 GB apart -- meeting the physical placement?  One context, val2 re-placed by strategies that gave the slow
classes (hipMalloc of the exact size, VMM chunks) and the fast one (one 8 GiB block), the product timed under several blockIdx ->
workgroup maps: the kernel's own, identity (all XCDs walk the same region), chunks of C workgroups dealt round-robin to the XCDs."""
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
L.fx_debug_spmv_ms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]
L.fx_debug_replace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_uint64)]


def ms(mp=0, kind=0):
    v = C.c_float(0)
    hip._chk(L.fx_debug_spmv_ms(ctx.h, kind, 2, 1, 1, 8, C.byref(v), mp))
    return v.value


def replace(what, how, arg=0):
    a = C.c_uint64(0)
    hip._chk(L.fx_debug_replace(ctx.h, what, how, arg, C.byref(a)))


maps = [0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
print("%-28s %s" % ("val2 placement \\ map", "  ".join("%7s" % ("own" if x == 0 else "ident" if x == 1 else "C=%d" % x) for x in maps)), flush=True)


def row(tag):
    print("%-28s %s" % (tag, "  ".join("%7.4f" % ms(x) for x in maps)), flush=True)


row("as placed (pow2)")
for rep in range(3):
    for how, arg, tag in ((0, 0, "hipMalloc exact"), (2, 1024, "VMM 1 GiB chunks"), (2, 64, "VMM 64 MiB chunks"), (1, 0, "hipMalloc pow2")):
        replace(0, how, arg)
        row("%s #%d" % (tag, rep))
