"""Same-context A/B at 10.1M DOF: ONE context (same data, same placement), the knobs flipped on it with fx_set_option between
timings.  usage: python scripts/r3/ab_opts.py [--eis] [--method M --precond P] "K=V,K=V" "K=V" ...   ("default" is always first)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
args = sys.argv[1:]
eis = "--eis" in args
if eis:
    args.remove("--eis"); os.environ["FX_EISENSTAT"] = "1"
method, precond = 1, 1
if "--method" in args:
    i = args.index("--method"); method = int(args[i + 1]); del args[i:i + 2]
if "--precond" in args:
    i = args.index("--precond"); precond = int(args[i + 1]); del args[i:i + 2]
from frontistr_amd import hecmw as hip
if os.environ.get('FX_LIBPATH'): hip.LIBPATH = os.environ['FX_LIBPATH']
from frontistr_amd.mesh import CubeMesh
n = int(os.environ.get("AB_N", "149"))
mesh = CubeMesh(n)
hm = hip.hecmwST_local_mesh(n_node=mesh.n_node)
hm.elem_node_item = mesh.conn.ravel()
m = hip.hecmw_mat_con(hm, hip.hecmwST_matrix())
ctx = hip.SolverContext()
ctx.upload(m, what=hip.FX_UP_PROFILE)
ctx.assemble_c3d8(mesh.coord, mesh.conn, 210000.0, 0.3, elemopt=1, load=mesh.load(), bc=mesh.dirichlet())
m.Iarray[1] = method; m.Iarray[2] = precond
ctx.precond_setup(m)
variants = ["default"] + args
base = {}
def apply(v):
    for k, val in base.items():
        ctx.set_option(k, val)
    if v != "default":
        for kv in v.split(","):
            k, val = kv.split("=")
            ctx.set_option(k, float(val))
# defaults to restore: every key that any variant touches, read from the env-free defaults given as "K=V" in AB_BASE
for kv in os.environ.get("AB_BASE", "").split(","):
    if kv:
        k, val = kv.split("="); base[k] = float(val)
def it_ms(steps=40):
    m.Iarray[0] = 1000; m.Rarray[0] = 1e-30
    ctx.krylov_begin(m)
    ctx.krylov_steps(8)
    ctx.synchronize()
    t0 = time.perf_counter()
    it, status, resid = ctx.krylov_steps(steps)
    ctx.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps, resid
for rep in range(3):
    for v in variants:
        apply(v)
        ms, resid = it_ms()
        print("rep %d  %-50s precond_apply %.4f ms  iteration %.4f ms  (resid %.3e, eisenstat=%d)"
              % (rep, v, ctx.precond_apply_ms(10), ms, resid, ctx.stats()["eisenstat"]), flush=True)
