"""Profiles that do not come from a mesh: random symmetric sparsity with isolated nodes (rows holding only their diagonal
block), a hub row (more than 32 off-diagonal blocks: the wide-row kernels), very uneven row lengths (slice padding) and halo
columns without a communicator.  SPD block values: sum over edges of [[P, -P], [-P, P]] with P SPD, plus a diagonal shift.
GPU against the CPU oracle for the 3x3 path (all preconditioners) and the generic-block path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def random_system(nd, N, n_halo, seed, hub=True):
    from oracle.refrun import BSR
    rng = np.random.default_rng(seed)
    NP = N + n_halo
    edges = set()
    live = np.arange(N)
    iso = set(rng.choice(N, 5, replace=False).tolist())
    live = np.array([i for i in live if i not in iso])
    for _ in range(3 * N):
        i, j = rng.choice(live, 2, replace=False)
        edges.add((min(i, j), max(i, j)))
    if hub:
        h = int(live[len(live) // 2])
        for j in rng.choice(live, min(60, len(live) - 1), replace=False):
            if j != h:
                edges.add((min(h, int(j)), max(h, int(j))))
    for k in range(n_halo):                     # each halo node hangs on two internal rows
        for i in rng.choice(live, 2, replace=False):
            edges.add((int(i), N + k))
    low = [[] for _ in range(NP)]
    up = [[] for _ in range(NP)]
    for i, j in edges:
        up[i].append(j)
        if j < N:
            low[j].append(i)
    indexL, indexU = np.zeros(NP + 1, dtype=np.int32), np.zeros(NP + 1, dtype=np.int32)
    itemL, itemU = [], []
    for i in range(NP):
        low[i].sort(); up[i].sort()
        itemL += [c + 1 for c in low[i]]
        itemU += [c + 1 for c in up[i]]
        indexL[i + 1], indexU[i + 1] = len(itemL), len(itemU)
    posL = {(i, c): indexL[i] + k for i in range(NP) for k, c in enumerate(low[i])}
    posU = {(i, c): indexU[i] + k for i in range(NP) for k, c in enumerate(up[i])}
    D = np.tile(0.05 * np.eye(nd), (NP, 1, 1))
    AL = np.zeros((max(len(itemL), 1), nd, nd))
    AU = np.zeros((max(len(itemU), 1), nd, nd))
    for i, j in sorted(edges):
        G = rng.standard_normal((nd, nd))
        P = G @ G.T / nd + 0.1 * np.eye(nd)
        D[i] += P
        AU[posU[(i, j)]] = -P
        if j < N:
            D[j] += P
            AL[posL[(j, i)]] = -P
    B = rng.standard_normal(nd * NP)
    B[nd * N:] = 0.0
    A = BSR(N, NP, indexL, np.array(itemL, dtype=np.int32), indexU, np.array(itemU, dtype=np.int32), D.ravel(),
            AL[:len(itemL)].ravel(), AU[:len(itemU)].ravel(), B, NDOF=nd)
    return A


def to_hip(hip, A):
    return hip.hecmwST_matrix.from_arrays(A.N, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, A.B.copy(),
                                          NDOF=A.NDOF)


@pytest.mark.parametrize("nd,N,n_halo,seed", [(3, 300, 0, 1), (3, 517, 40, 2), (3, 64, 0, 3), (2, 300, 25, 4), (6, 130, 0, 5),
                                              (1, 1000, 0, 6)])
def test_random_pattern_matvec_and_solves(hip, oracle, nd, N, n_halo, seed):
    from oracle.refrun import default_params
    A = random_system(nd, N, n_halo, seed)
    ctx = hip.SolverContext()
    m = to_hip(hip, A)
    x = np.random.default_rng(seed).standard_normal(nd * A.NP)
    y = np.zeros(nd * A.NP)
    hip.hecmw_matvec(None, m, x.copy(), y, ctx=ctx)
    ref = oracle.matvec(A, x)
    assert np.abs(y[:nd * N] - ref[:nd * N]).max() <= 1e-13 * np.abs(ref).max()
    cases = [(1, 3), (1, 1), (2, 1)] + ([(2, 10), (1, 10), (3, 3), (4, 1)] if nd == 3 else [])
    for meth, pc in cases:
        I, R = default_params(method=meth, precond=pc)
        stagnates = (nd == 6 and meth == 1 and pc == 1)
        if stagnates:       # the reference's SSOR_66 quirk makes this preconditioner unsymmetric: its own CG stalls at 7e-5 here
            I[0] = 30
        o = oracle.solve_iterative(A, I, R, nthreads=4)
        m = to_hip(hip, A)
        m.Iarray[:] = I
        m.Rarray[:] = R
        code = hip.hecmw_solve(None, m, ctx=ctx)
        if stagnates:
            assert code == o["code"] == 3001 and m.Iarray[80] == 0 and ctx.info.iterations == o["iter"] == 31
            assert np.all(np.abs(ctx.history[:12] - o["history"][:12]) <= 1e-6 * o["history"][:12])
            continue
        assert code == o["code"] == 0 and m.Iarray[80] == 1, (meth, pc, code, o["code"])
        tol_it = max(1, int(0.15 * o["iter"])) if meth != 1 else 1
        assert abs(ctx.info.iterations - o["iter"]) <= tol_it, (meth, pc, ctx.info.iterations, o["iter"])
        assert np.abs(m.X - o["X"])[:nd * N].max() <= 2e-7 * np.abs(o["X"]).max(), (meth, pc)
    ctx.close()
