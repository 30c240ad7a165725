"""C-ABI surface: the library loads and exports every symbol include/fistr_hip.h declares
(no compute without a GPU), and the host-only entry point fx_mat_con matches the oracle."""
import os
import re

import numpy as np

from conftest import load_golden, ROOT


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
