"""Which array and which allocation strategy decide the SpMV's speed class?  ONE context at 10.1M DOF, same
data and kernel throughout; the arrays the product touches are moved between allocations with fx_debug_replace and the product is
timed with fx_debug_spmv_ms on chosen vectors.  usage: python scripts/r4/placement_probe.py [phases]   (default: all)
Phases: vec (x / y choice), val (val2 re-placed: hipMalloc exact / pow2 / VMM chunks / arena offsets), col (col2), small (pair_ptr)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("FX_TUNE_PLACEMENT", "0")
from frontistr_amd import hecmw as hip
from frontistr_amd.mesh import CubeMesh

phases = sys.argv[1].split(",") if len(sys.argv) > 1 else ["vec", "val", "col", "small", "arena"]
n = int(os.environ.get("AB_N", "149"))
L = hip.lib()
mesh = CubeMesh(n)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
ctx = hip.SolverContext()
ctx.upload(m, what=hip.FX_UP_PROFILE)
ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
m.Iarray[1] = 1
m.Iarray[2] = 1
ctx.precond_setup(m)


def addrs():
    out = (C.c_uint64 * 20)()
    hip._chk(L.fx_debug_addresses(ctx.h, out))
    return list(out)


def ms(x=2, y=1, dot=1, kind=0, rep=5):
    v = C.c_float(0)
    hip._chk(L.fx_debug_spmv_ms(ctx.h, kind, x, y, dot, rep, C.byref(v)))
    return v.value


def replace(what, how, arg=0):
    a = C.c_uint64(0)
    L.fx_debug_replace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_uint64)]
    hip._chk(L.fx_debug_replace(ctx.h, what, how, arg, C.byref(a)))
    return a.value


a = addrs()
names = ["val2", "col2", "pair_ptr", "slice_order", "Bs", "Xs"] + ["W%d" % k for k in range(10)] + ["partials"]
print("addresses (GiB offset from the lowest):")
lo = min(x for x in a[:17] if x)
for k, nm in enumerate(names):
    print("  %-12s 0x%012x  +%.3f GiB%s" % (nm, a[k], (a[k] - lo) / 2.0**30, "  2MiB-aligned" if a[k] % (2 << 20) == 0 else ""))
print("val2 bytes %.3f GB, vector bytes %.1f MB" % (a[17] / 1e9, a[18] / 1e6), flush=True)


def line(tag):
    t1, t2, t3 = ms(2, 1, 1), ms(-1, 7, 0), ms(kind=1)
    print("%-46s loop(p=W2,q=W1,dot) %.4f  probe(Bs,W7) %.4f  stream_read %.4f ms (%.0f GB/s)"
          % (tag, t1, t2, t3, a[17] / (t3 * 1e-3) / 1e9), flush=True)


line("as set up")
if "vec" in phases:
    print("== x / y choice (dot=1), ms")
    sel = [-1, -2] + list(range(10))
    lab = {-1: "Bs", -2: "Xs"}
    print("      y=W1    y=W7    y=W9")
    for x in sel:
        row = []
        for y in (1, 7, 9):
            row.append(ms(x, y, 1) if x != y else float("nan"))
        print("x=%-3s %s" % (lab.get(x, "W%d" % x), "  ".join("%.4f" % v for v in row)), flush=True)
if "val" in phases:
    print("== val2 re-placed (everything else untouched)")
    for rep in range(2):
        for how, arg, tag in ((0, 0, "hipMalloc exact"), (1, 0, "hipMalloc pow2 (8 GiB)"), (2, 2, "VMM 2 MiB chunks"), (2, 64, "VMM 64 MiB chunks"),
                              (2, 1024, "VMM 1 GiB chunks"), (2, 8192, "VMM one 8 GiB chunk")):
            try:
                ad = replace(0, how, arg)
                line("val2 %-24s @+%.2f GiB" % (tag, (ad - lo) / 2.0**30))
            except Exception as e:
                print("val2 %s: FAILED %r" % (tag, e), flush=True)
if "arena" in phases:
    print("== val2 at offsets inside ONE arena (hipMalloc 32 GiB)")
    for off in (0, 1024, 2048, 4096, 8192, 12288, 16384, 20480, 24576, 0, 1, 65):
        try:
            ad = replace(0, 3, off)
            line("val2 arena +%d MiB" % off)
        except Exception as e:
            print("arena %d: FAILED %r" % (off, e), flush=True)
if "col" in phases:
    print("== col2 re-placed")
    for how, arg, tag in ((0, 0, "hipMalloc exact"), (1, 0, "hipMalloc pow2"), (2, 64, "VMM 64 MiB chunks"), (0, 0, "hipMalloc exact")):
        replace(1, how, arg)
        line("col2 %s" % tag)
if "small" in phases:
    print("== pair_ptr + slice_order re-placed")
    for how in (0, 0):
        replace(2, how, 0)
        line("pair_ptr/slice_order hipMalloc")
    print("== x (W2) and y (W1) re-placed")
    for how, arg, tag in ((0, 0, "hipMalloc"), (1, 0, "pow2"), (2, 2, "VMM 2 MiB"), (0, 0, "hipMalloc")):
        replace(12, how, arg)
        replace(11, how, arg)
        line("W2, W1 %s" % tag)
print("done", flush=True)
