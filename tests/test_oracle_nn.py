"""The CPU restatement of the reference's NDOF != 3 routines (oracle/hecmw_nn_oracle.c) against the golden vectors the REAL
reference produced (tests/golden/nn.npz <- tests/golden/make_nn_golden.py): bit-identical solutions for NDOF = 1, 2, 4, 5, 6,
including the SSOR_66 substitution quirk the reference carries (precond/66/hecmw_precond_SSOR_66.f90:420, :504)."""
import numpy as np
import pytest

from conftest import load_golden
from nn_cases import NN_CASES, NN_NDOF, dense, nn_system, nn_tag


@pytest.mark.parametrize("nd", NN_NDOF)
@pytest.mark.parametrize("meth,pc", NN_CASES)
def test_nn_oracle_bit_identical_to_reference(oracle, nd, meth, pc):
    from oracle.refrun import default_params
    g = load_golden("nn")
    A = nn_system(nd)
    I, R = default_params(method=meth, precond=pc)
    o = oracle.solve_iterative(A, I, R, nthreads=4)
    tag = nn_tag(nd, meth, pc)
    assert o["code"] == 0 and o["iter"] == int(g[tag + "iter"]) and o["Iarray"][80] == 1
    assert np.array_equal(o["X"], g[tag + "X"])
    hr = g[tag + "hist"]
    assert len(o["history"]) == len(hr) and np.all(np.abs(o["history"] - hr) <= 1e-6 * hr)      # stdout prints 7 digits
    xs = np.linalg.solve(dense(A), A.B)
    assert np.abs(o["X"] - xs).max() <= 1e-6 * np.abs(xs).max()


@pytest.mark.parametrize("nd", [1, 6])
def test_nn_oracle_vs_live_reference_other_methods(oracle, nd):
    """GMRES / GPBiCG and the serial (natural order) SSOR through the same generic restatement, where oracle/_ref exists."""
    from oracle import refrun
    if not refrun.have_ref("ref_solve"):
        pytest.skip("oracle/_ref not built")
    A = nn_system(nd)
    for meth, pc, thr in ((3, 3, 1), (4, 1, 4), (1, 1, 1)):
        I, R = refrun.default_params(method=meth, precond=pc)
        r = refrun.run_solve(A, I, R, threads=thr)
        o = oracle.solve_iterative(A, I, R, nthreads=thr)
        assert o["iter"] == r["iter"] and np.array_equal(o["X"], r["X"]), (meth, pc, thr)


def test_matvec_nn_matches_dense(oracle):
    for nd in NN_NDOF:
        A = nn_system(nd)
        x = np.random.default_rng(nd).standard_normal(nd * A.NP)
        y = oracle.matvec(A, x)
        assert np.abs(y - dense(A) @ x).max() <= 1e-12 * np.abs(y).max()


@pytest.mark.parametrize("nd", [1, 2, 4, 5, 6])
def test_nn_scaling_oracle_vs_live_reference(oracle, nd):
    """SCALING=YES (hecmw_solver_scaling_nn and its 44 / 66 copies) through the generic restatement: bit-identical to the reference."""
    from oracle import refrun
    if not refrun.have_ref("ref_solve"):
        pytest.skip("oracle/_ref not built")
    A = nn_system(nd)
    for meth, pc, thr in ((1, 3, 1), (1, 1, 4), (2, 1, 4), (3, 3, 1)):
        I, R = refrun.default_params(method=meth, precond=pc)
        I[6] = 1
        r = refrun.run_solve(A, I, R, threads=thr)
        o = oracle.solve_iterative(A, I, R, nthreads=thr)
        assert o["iter"] == r["iter"] and np.array_equal(o["X"], r["X"]), (meth, pc, thr)
