"""C-ABI surface: the library loads and exports every symbol include/fistr_hip.h declares
(no compute without a GPU), and the host-only entry point fx_mat_con matches the oracle."""
import os
import re

import numpy as np

from conftest import golden_matrix, load_golden, ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "fistr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from frontistr_amd import hecmw
    L = hecmw.lib()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), n
    assert b"gfx950" in L.fx_version()


def test_no_device_is_a_loud_error():
    """Without a GPU the product path must fail loudly, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        return
    from frontistr_amd import hecmw
    try:
        hecmw.SolverContext()
    except hecmw.HecmwSolverError as e:
        assert e.code < 0
    else:
        raise AssertionError("SolverContext() succeeded without a GPU")


def test_mat_con_matches_oracle(oracle):
    from frontistr_amd import hecmw
    for deck in ("cube4", "cube3s", "exA_A361"):
        g = load_golden(deck)
        mesh = hecmw.hecmwST_local_mesh(n_node=g["coord"].shape[0])
        mesh.elem_node_item = g["conn"].ravel()
        mat = hecmw.hecmw_mat_con(mesh, hecmw.hecmwST_matrix())
        for k in ("indexL", "itemL", "indexU", "itemU"):
            assert np.array_equal(getattr(mat, k), g["ic_" + k]), (deck, k)


def test_mat_con_larger_mesh(oracle):
    from frontistr_amd import hecmw
    from frontistr_amd.mesh import CubeMesh, cube_blocks
    m = CubeMesh(12)
    mesh = hecmw.hecmwST_local_mesh(n_node=m.n_node)
    mesh.elem_node_item = m.conn.ravel()
    mat = hecmw.hecmw_mat_con(mesh, hecmw.hecmwST_matrix())
    iL, jL, iU, jU = oracle.mat_con(m.n_node, m.conn)
    assert np.array_equal(mat.indexL, iL) and np.array_equal(mat.itemL, jL)
    assert np.array_equal(mat.indexU, iU) and np.array_equal(mat.itemU, jU)
    N, NPL, NPU, nb = cube_blocks(12)
    assert (mat.N, mat.NPL, mat.NPU) == (N, NPL, NPU)


def test_color_elements_is_a_valid_colouring():
    """Host-only fx_color_elements: every element appears once, no two elements of a colour share a node, 8 colours on a
    structured hex mesh, graceful answer (ncolor = 0 -> atomics on the device) when a node sits in more than 64 elements."""
    import ctypes as C
    from frontistr_amd import hecmw
    from frontistr_amd.mesh import CubeMesh
    L = hecmw.lib()

    def colour(conn, NP):
        conn = np.ascontiguousarray(conn, dtype=np.int32)
        order = np.zeros(conn.shape[0], dtype=np.int32)
        off = np.zeros(65, dtype=np.int32)
        nc = C.c_int32(0)
        assert L.fx_color_elements(NP, conn.shape[0], conn.shape[1], hecmw._ptr(conn), hecmw._ptr(order), hecmw._ptr(off),
                                   C.byref(nc)) == 0
        return order, off, nc.value

    m = CubeMesh(6, skew=0.1)
    order, off, nc = colour(m.conn, m.n_node)
    assert nc == 8 and off[nc] == m.conn.shape[0] and np.array_equal(np.sort(order), np.arange(m.conn.shape[0]))
    rng = np.random.default_rng(0)
    conn = m.conn[rng.permutation(m.conn.shape[0])]                 # an element order that is not the natural one
    g = load_golden("exA_A361")
    for cn, NP in ((conn, m.n_node), (g["conn"], g["coord"].shape[0])):
        order, off, nc = colour(cn, NP)
        assert nc >= 8 and np.array_equal(np.sort(order), np.arange(cn.shape[0]))
        for k in range(nc):
            nodes = cn[order[off[k]:off[k + 1]]].ravel()
            assert np.unique(nodes).size == nodes.size, k
    star = np.stack([np.r_[1, 2 + 7 * e + np.arange(7)] for e in range(70)]).astype(np.int32)   # node 1 in 70 elements
    assert colour(star, 1 + 7 * 70)[2] == 0
    bad = m.conn.copy()
    bad[0, 0] = m.n_node + 1
    order = np.zeros(bad.shape[0], dtype=np.int32)
    off = np.zeros(65, dtype=np.int32)
    nc = C.c_int32(0)
    assert L.fx_color_elements(m.n_node, bad.shape[0], 8, hecmw._ptr(bad), hecmw._ptr(order), hecmw._ptr(off), C.byref(nc)) != 0


def _lib_ordering(m, ncolor_in):
    import ctypes as C
    from frontistr_amd import hecmw
    L = hecmw.lib()
    perm = np.zeros(m.N, dtype=np.int32)
    cidx = np.zeros(m.N + 2, dtype=np.int32)
    nc = C.c_int32(0)
    ptr = lambda a: np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(C.c_void_p)
    iL, jL, iU, jU = (np.ascontiguousarray(a, dtype=np.int32) for a in (m.indexL, m.itemL, m.indexU, m.itemU))
    assert L.fx_ssor_ordering(m.N, ptr(iL), ptr(jL), ptr(iU), ptr(jU), ncolor_in, perm.ctypes.data_as(C.c_void_p),
                              cidx.ctypes.data_as(C.c_void_p), cidx.size, C.byref(nc)) == 0
    return perm, cidx[:nc.value + 1].copy()


def test_ssor_ordering_is_the_references(oracle):
    """Host only: the library's level ordering + capped greedy multicolouring (threaded graph build, the colouring walk on a
    graph relabelled by visiting position) against the oracle's restatement of hecmw_matrix_ordering_CM / _MC, which produces
    the reference's colours (bit-exact SSOR runs, tests/test_oracle_golden.py): same perm, same COLORindex, several
    NCOLOR_IN, a structured cube large enough for the threaded paths and a skewed one."""
    from frontistr_amd.mesh import CubeMesh
    cases = [(golden_matrix(load_golden(d)), nc) for d in ("cube4", "cube3s", "exA_A361") for nc in (10, 3)]
    for n, skew, nc in ((12, 0.05, 10), (33, 0.0, 10), (20, 0.0, 4)):
        mesh = CubeMesh(n, skew=skew)
        cases.append((oracle.assemble(1, mesh.coord, mesh.conn, 210000.0, 0.3, bc=mesh.dirichlet(), load=mesh.load()), nc))
    for A, nc in cases:
        P = oracle.Precond(A, 1, ncolor_in=nc, nthreads=4)
        perm, cidx = _lib_ordering(A, nc)
        assert np.array_equal(cidx, P.colorindex), (A.N, nc)
        assert np.array_equal(perm, P.perm), (A.N, nc)


def test_march_programs_replay_on_the_host():
    """The schedule builder of the plane march (csrc/fx_march.h, host-only entry fx_march_plan): for structured and unstructured profiles and
    chunkings aligned with the mesh planes or not, both programs pass the replay -- every in-chunk dependency is found in the LDS-ring slot
    its column code names, every other one is produced by an earlier chunk or round, the chained row counts reproduce the round table --
    and the numbers add up: chunks, one round at least per dependency level of a chunk, every block either near or far.  A mesh whose
    rows have more than 14 lower blocks (the mesher-made hex mesh of tutorial 05) is not admitted."""
    from frontistr_amd import hecmw
    from frontistr_amd.mesh import CubeMesh
    for n, chunks in ((5, (36, 18, 25, 216, 7)), (11, (144, 48, 100, 1728)), (20, (441, 147, 1000))):
        m = CubeMesh(n)
        mesh = hecmw.hecmwST_local_mesh(n_node=m.n_node)
        mesh.elem_node_item = m.conn.ravel()
        mat = hecmw.hecmw_mat_con(mesh, hecmw.hecmwST_matrix())
        nlower = int(mat.indexL[mat.N])
        for S in chunks:
            for waves in (1, 2, 3):
                p = hecmw.march_plan(mat, S, waves)
                assert p["admitted"] == 1, (n, S, waves)
                assert p["chunks"] == -(-mat.N // S) and p["max_round_rows"] <= 8 * waves
                assert p["near_blocks"] + p["far_blocks"] == nlower
                assert p["rounds_fwd"] >= p["chunks"] and p["rounds_bwd"] >= p["chunks"]
                if S >= mat.N:                       # one chunk: its levels are the matrix's, every round a level or a part of one
                    assert p["rounds_fwd"] >= p["levels"] == 7 * n + 1
        # whole planes per chunk: every lower neighbour of the same plane is at most 3 levels back -> in the ring; the 9 of the plane below are not
        p = hecmw.march_plan(mat, (n + 1) ** 2, 3)
        assert p["far_blocks"] > p["near_blocks"] > 0
    g = np.load(os.path.join(ROOT, "tests", "golden", "nl_necking.npz"))
    mesh = hecmw.hecmwST_local_mesh(n_node=g["coord"].shape[0])
    mesh.elem_node_item = g["conn"].ravel()
    mat = hecmw.hecmw_mat_con(mesh, hecmw.hecmwST_matrix())
    p = hecmw.march_plan(mat, 500, 2)
    assert p["levels"] > 0
    if p["admitted"]:
        assert p["near_blocks"] + p["far_blocks"] == int(mat.indexL[mat.N])
