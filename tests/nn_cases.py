"""Synthetic block systems with NDOF = 1, 2, 4, 6 (the reference's 11 / 22 / 44 / 66 / nn code paths) on the
profile of a skewed cube of hex8 elements: every element contributes a random SPD local matrix (seeded), node-major with
NDOF rows per node, so the global matrix is SPD with the reference's D / AL / AU layout (row-major NDOF x NDOF blocks)."""
import numpy as np

NN_NDOF = [1, 2, 4, 5, 6]
# (METHOD, PRECOND): CG / BiCGSTAB x SSOR (RCM + multicolour, the reference's OpenMP path) / DIAG
NN_CASES = [(1, 1), (1, 3), (2, 1), (2, 3)]


def nn_tag(nd, meth, pc):
    return "n%d_m%d_p%d_" % (nd, meth, pc)


def nn_system(ndof, m=4, seed=0, halo=0):
    from frontistr_amd.mesh import CubeMesh
    from oracle import pyoracle
    from oracle.refrun import BSR
    mesh = CubeMesh(m, skew=0.1)
    NP = mesh.n_node
    indexL, itemL, indexU, itemU = pyoracle.mat_con(NP, mesh.conn)
    nd2 = ndof * ndof
    D = np.zeros((NP, ndof, ndof))
    AL = np.zeros((itemL.size, ndof, ndof))
    AU = np.zeros((itemU.size, ndof, ndof))
    posL = {(i, int(itemL[j]) - 1): j for i in range(NP) for j in range(indexL[i], indexL[i + 1])}
    posU = {(i, int(itemU[j]) - 1): j for i in range(NP) for j in range(indexU[i], indexU[i + 1])}
    rng = np.random.default_rng(1000 * ndof + seed)
    for e in range(mesh.conn.shape[0]):
        nod = mesh.conn[e] - 1
        G = rng.standard_normal((8 * ndof, 8 * ndof))
        L = G @ G.T / (8 * ndof) + 0.02 * np.eye(8 * ndof)
        for a in range(8):
            for b in range(8):
                blk = L[a * ndof:(a + 1) * ndof, b * ndof:(b + 1) * ndof]
                i, j = int(nod[a]), int(nod[b])
                if i == j:
                    D[i] += blk
                elif j < i:
                    AL[posL[(i, j)]] += blk
                else:
                    AU[posU[(i, j)]] += blk
    B = rng.standard_normal(ndof * NP)
    A = BSR(NP, NP, indexL, itemL, indexU, itemU, D.ravel(), AL.ravel(), AU.ravel(), B, NDOF=ndof)
    return A


def dense(A):
    """Dense (NDOF*N)^2 image of the internal rows / columns (tests only)."""
    nd, N = A.NDOF, A.N
    M = np.zeros((nd * N, nd * N))
    D = A.D.reshape(-1, nd, nd)
    AL = A.AL.reshape(-1, nd, nd)
    AU = A.AU.reshape(-1, nd, nd)
    for i in range(N):
        M[nd * i:nd * i + nd, nd * i:nd * i + nd] = D[i]
        for j in range(A.indexL[i], A.indexL[i + 1]):
            k = A.itemL[j] - 1
            M[nd * i:nd * i + nd, nd * k:nd * k + nd] = AL[j]
        for j in range(A.indexU[i], A.indexU[i + 1]):
            k = A.itemU[j] - 1
            if k < N:
                M[nd * i:nd * i + nd, nd * k:nd * k + nd] = AU[j]
    return M
