"""Block sizes other than 3x3 on the GPU: hecmw_solve / hecmw_matvec with NDOF = 1, 2, 4, 5, 6 against
 * the golden vectors of the REAL reference (tests/golden/nn.npz, made by tests/golden/make_nn_golden.py) and
 * the CPU oracle (bit-identical to the reference on these systems, tests/test_oracle_nn.py).
Tolerances: iteration count +-1, residual history 1e-8 relative to the first residual, X 1e-8 relative."""
import numpy as np
import pytest

from conftest import load_golden
from nn_cases import NN_CASES, NN_NDOF, dense, nn_system, nn_tag

pytestmark = pytest.mark.gpu


def to_hip(hip, A):
    return hip.hecmwST_matrix.from_arrays(A.N, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, A.B.copy(),
                                          NDOF=A.NDOF)


@pytest.mark.parametrize("nd", NN_NDOF)
def test_matvec_nn(hip, oracle, nd):
    A = nn_system(nd)
    m = to_hip(hip, A)
    rng = np.random.default_rng(nd)
    x = rng.standard_normal(nd * A.NP)
    y = np.zeros(nd * A.NP)
    ctx = hip.SolverContext()
    hip.hecmw_matvec(None, m, x, y, ctx=ctx)
    ref = oracle.matvec(A, x)
    assert np.abs(y - ref).max() <= 1e-13 * np.abs(ref).max()
    assert np.abs(dense(A) @ x - ref).max() <= 1e-12 * np.abs(ref).max()
    ctx.close()


@pytest.mark.parametrize("nd", NN_NDOF)
@pytest.mark.parametrize("meth,pc", NN_CASES)
def test_solve_nn_vs_reference_golden(hip, nd, meth, pc):
    g = load_golden("nn")
    A = nn_system(nd)
    m = to_hip(hip, A)
    m.Iarray[0], m.Iarray[1], m.Iarray[2] = 10000, meth, pc
    ctx = hip.SolverContext()
    code = hip.hecmw_solve(None, m, ctx=ctx)
    tag = nn_tag(nd, meth, pc)
    assert code == 0 and m.Iarray[80] == 1
    assert abs(ctx.info.iterations - int(g[tag + "iter"])) <= 1
    xr = g[tag + "X"]
    assert np.abs(m.X - xr).max() <= 1e-8 * np.abs(xr).max()
    h, hr = ctx.history, g[tag + "hist"]
    k = min(len(h), len(hr))
    assert k >= 3 and np.all(np.abs(h[:k] - hr[:k]) <= 2e-6 * hr[0] + 1e-6 * hr[:k])   # stdout prints 7 digits
    ctx.close()


def test_nn_flags_errors_and_reuse(hip, oracle):
    """Zero RHS (W-2002, X = 0), zero diagonal (E-2001), MAXIT (W-3001), unsupported options, and one context serving
    several block sizes and the 3x3 path in turn."""
    A = nn_system(2)
    ctx = hip.SolverContext()
    m = to_hip(hip, A)
    m.B[:] = 0.0
    m.X[:] = 1.0
    assert hip.hecmw_solve(None, m, ctx=ctx) == 2002 and not m.X.any()
    m = to_hip(hip, A)
    m.Iarray[0] = 3
    assert hip.hecmw_solve(None, m, ctx=ctx) == 3001 and m.Iarray[80] == 0 and ctx.info.iterations == 4
    m = to_hip(hip, A)
    m.D = m.D.copy()
    m.D[4 * 5 + 3] = 0.0
    with pytest.raises(hip.HecmwSolverError):
        hip.hecmw_solve(None, m, ctx=ctx)
    for bad in ((1, 10), (5, 3)):        # block ILU of the other block sizes / an unknown method are not on the GPU path
        m = to_hip(hip, A)
        m.Iarray[1], m.Iarray[2] = bad
        with pytest.raises(hip.HecmwSolverError):
            hip.hecmw_solve(None, m, ctx=ctx)
    for nd in (6, 1, 4):
        A2 = nn_system(nd)
        m = to_hip(hip, A2)
        m.Iarray[0] = 10000
        assert hip.hecmw_solve(None, m, ctx=ctx) == 0
        o = oracle.solve_iterative(A2, m.Iarray * 0 + hip.hecmwST_matrix().Iarray, m.Rarray, nthreads=4)
        assert np.abs(m.X - o["X"]).max() <= 1e-8 * np.abs(o["X"]).max()
    ctx.close()


def test_nn_recycled_preconditioner_and_new_values(hip, oracle):
    """Second solve with changed values and Iarray(97) = 2 (force) rebuilds layout + preconditioner; with the flags down the
    resident ones are reused (same answer for the same matrix).  Iarray(97) = 1 goes through the recycle policy:
    tests/test_gpu_parity.py::test_preconditioner_recycle_policy."""
    A = nn_system(4)
    ctx = hip.SolverContext()
    m = to_hip(hip, A)
    m.Iarray[0] = 10000
    assert hip.hecmw_solve(None, m, ctx=ctx) == 0
    x1 = m.X.copy()
    m.X[:] = 0.0
    assert m.Iarray[96] == 0 and m.Iarray[97] == 0
    assert hip.hecmw_solve(None, m, ctx=ctx) == 0 and np.array_equal(m.X, x1)
    A.D[:] = A.D * 1.5
    m = to_hip(hip, A)
    m.Iarray[0] = 10000
    m.Iarray[97] = 0
    m.Iarray[96] = 2
    assert hip.hecmw_solve(None, m, ctx=ctx) == 0
    I = hip.hecmwST_matrix().Iarray
    I[0] = 10000
    o = oracle.solve_iterative(A, I, m.Rarray, nthreads=4)
    assert np.abs(m.X - o["X"]).max() <= 1e-8 * np.abs(o["X"]).max() and abs(ctx.info.iterations - o["iter"]) <= 1
    ctx.close()


def test_matvec_follows_changed_values_and_block_sizes_interleave(hip, oracle):
    """hecmw_matvec has no 'matrix changed' flag: values passed are used on every call (3x3 and generic blocks); a NULL D means
    the resident values.  One context serves a 3x3 system, a generic-block system and the 3x3 system again."""
    from conftest import golden_matrix, load_golden
    g = load_golden("cube4")
    A3 = golden_matrix(g)
    ctx = hip.SolverContext()
    for A in (A3, nn_system(4)):
        nd = A.NDOF
        m = to_hip(hip, A)
        x = np.random.default_rng(3).standard_normal(nd * A.NP)
        y = np.zeros(nd * A.NP)
        hip.hecmw_matvec(None, m, x.copy(), y, ctx=ctx)
        assert np.abs(y - oracle.matvec(A, x)).max() <= 1e-13 * np.abs(y).max()
        m.D = m.D * 2.0                                        # the caller changes hecMAT between two products
        A2 = type(A)(A.N, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, m.D, A.AL, A.AU, A.B, NDOF=nd)
        hip.hecmw_matvec(None, m, x.copy(), y, ctx=ctx)
        ref2 = oracle.matvec(A2, x)
        assert np.abs(y - ref2).max() <= 1e-13 * np.abs(ref2).max()
        m.D = m.AL = m.AU = None                               # resident values
        y2 = np.zeros_like(y)
        hip.hecmw_matvec(None, m, x.copy(), y2, ctx=ctx)
        assert np.array_equal(y, y2)
    m = to_hip(hip, A3)                                        # back to 3x3 on the same context: solve
    m.Iarray[0], m.Iarray[2] = 10000, 1
    assert hip.hecmw_solve(None, m, ctx=ctx) == 0
    assert np.abs(m.X - g["sol_m1_p1_t4_X"]).max() < 1e-8 * np.abs(g["sol_m1_p1_t4_X"]).max()
    ctx.close()


@pytest.mark.parametrize("nd", [1, 4, 6])
@pytest.mark.parametrize("meth,pc", [(3, 3), (3, 1), (4, 1), (4, 3)])
def test_gmres_gpbicg_generic_blocks(hip, oracle, nd, meth, pc):
    """GMRES(m) / GPBiCG (the host-scalar solvers of the 3x3 path, shared through an operations policy) on generic blocks against
    the oracle, which is bit-identical to the real reference for these methods (tests/test_oracle_nn.py)."""
    from oracle.refrun import default_params
    A = nn_system(nd)
    I, R = default_params(method=meth, precond=pc)
    if nd == 6 and pc == 1:
        I[0] = 40          # the reference's SSOR_66 quirk: unsymmetric preconditioner, compare a fixed number of steps
    o = oracle.solve_iterative(A, I, R, nthreads=4)
    m = to_hip(hip, A)
    m.Iarray[:] = I
    m.Rarray[:] = R
    ctx = hip.SolverContext()
    code = hip.hecmw_solve(None, m, ctx=ctx)
    assert code == o["code"]
    assert abs(ctx.info.iterations - o["iter"]) <= max(1, int(0.1 * o["iter"])), (ctx.info.iterations, o["iter"])
    if code == 0:
        assert m.Iarray[80] == 1 and np.abs(m.X - o["X"]).max() <= 1e-7 * np.abs(o["X"]).max()
    k = min(5, len(ctx.history), len(o["history"]))
    assert np.all(np.abs(ctx.history[:k] - o["history"][:k]) <= 1e-7 * o["history"][0] + 1e-6 * o["history"][:k])
    ctx.close()


@pytest.mark.parametrize("nd", [1, 2, 6])
@pytest.mark.parametrize("meth,pc", [(1, 3), (1, 1), (2, 1), (3, 3)])
def test_scaling_option_generic_blocks(hip, oracle, nd, meth, pc):
    """SCALING=YES (Iarray(7) = 1, hecmw_solver_scaling_nn) on generic blocks against the oracle (bit-identical to the real reference on
    these runs); afterwards the resident matrix is the caller's again: an unscaled solve on the same context gives the usual answer."""
    from oracle.refrun import default_params
    A = nn_system(nd)
    I, R = default_params(method=meth, precond=pc)
    if nd == 6 and pc == 1 and meth == 1:
        I[0] = 25          # SSOR_66 quirk: CG stalls in the reference too; compare a fixed number of steps
    I[6] = 1
    o = oracle.solve_iterative(A, I, R, nthreads=4)
    m = to_hip(hip, A)
    m.Iarray[:] = I
    m.Rarray[:] = R
    ctx = hip.SolverContext()
    code = hip.hecmw_solve(None, m, ctx=ctx)
    assert code == o["code"] and abs(ctx.info.iterations - o["iter"]) <= max(1, int(0.1 * o["iter"])), (code, ctx.info.iterations, o["iter"])
    if code == 0:
        assert np.abs(m.X - o["X"]).max() <= 1e-7 * np.abs(o["X"]).max()
    k = min(5, len(ctx.history), len(o["history"]))
    assert np.all(np.abs(ctx.history[:k] - o["history"][:k]) <= 1e-7 * o["history"][0] + 1e-6 * o["history"][:k])
    x = np.random.default_rng(1).standard_normal(nd * A.NP)     # the resident values are unscaled again
    m.D = m.AL = m.AU = None
    y = np.zeros(nd * A.NP)
    hip.hecmw_matvec(None, m, x.copy(), y, ctx=ctx)
    ref = oracle.matvec(A, x)
    assert np.abs(y - ref).max() <= 1e-12 * np.abs(ref).max()
    ctx.close()
