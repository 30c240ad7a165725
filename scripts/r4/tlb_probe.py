"""Is the slow class of the SpMV address translation?  ONE context; the value array re-placed several times (strategies that gave both
classes); per placement the product is launched NREP+1 times in the spatial walk and NREP+1 times in ascending (= storage) order, and
k_stream_read once.  Run plain for the timings and under `rocprofv3 --pmc TCP_UTCL1_...` for the per-dispatch counters: the k-th
group of k_spmv dispatches in the counter CSV belongs to the k-th line printed here (scripts/r4/tlb_summary.py joins them)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("FX_TUNE_PLACEMENT", "0")
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh

NREP = 3
L = hip.lib()
mesh = CubeMesh(int(os.environ.get("AB_N", "149")))
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
ctx = hip.SolverContext()
ctx.upload(m, what=hip.FX_UP_PROFILE)
ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
m.Iarray[1] = 1
m.Iarray[2] = 3          # block-Jacobi would keep the natural numbering: we want the colour-major one
m.Iarray[2] = 1
ctx.precond_setup(m)
L.fx_debug_spmv_ms.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]
L.fx_debug_replace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_uint64)]


def ms(mp=0, kind=0):
    v = C.c_float(0)
    hip._chk(L.fx_debug_spmv_ms(ctx.h, kind, 2, 1, 1, NREP, C.byref(v), mp))
    return v.value


def replace(what, how, arg=0):
    a = C.c_uint64(0)
    hip._chk(L.fx_debug_replace(ctx.h, what, how, arg, C.byref(a)))


print("NREP %d" % NREP)
k = 0


def row(tag):
    global k
    print("group %2d  %-26s spatial %.4f  ascending %.4f  stream_read %.4f" % (k, tag, ms(0), ms(-1), ms(0, 1)), flush=True)
    k += 1


row("as placed (pow2)")
for rep in range(3):
    for how, arg, tag in ((0, 0, "hipMalloc exact"), (1, 0, "hipMalloc pow2"), (2, 2, "VMM 2 MiB chunks"), (2, 1024, "VMM 1 GiB chunks")):
        replace(0, how, arg)
        row("%s #%d" % (tag, rep))
replace(0, 3, 0)
row("arena 32 GiB +0")
