"""The reference's matrix dump files (DUMPTYPE=BSR; hecmw_matrix_dump.f90:235-369) as a fixture channel:
tests/golden/dump_cube4/ was written by the reference's own hecmw_solve_iterative (make_dump_golden.py)."""
import os

import numpy as np
import pytest

from conftest import golden_matrix, load_golden

DUMP = os.path.join(os.path.dirname(__file__), "golden", "dump_cube4", "dump_matrix_1_0")


def test_read_reference_dump():
    from frontistr_amd import hecmw_dump
    g = load_golden("cube4")
    A = golden_matrix(g)
    m = hecmw_dump.read_bsr(DUMP + ".bsr")
    assert (m.N, m.NP, m.NPL, m.NPU) == (A.N, A.NP, A.NPL, A.NPU)
    for k in ("indexL", "itemL", "indexU", "itemU"):
        assert np.array_equal(getattr(m, k), getattr(A, k)), k
    scale = np.abs(A.D).max()
    for k in ("D", "AL", "AU"):                                   # the dump prints 12 significant digits
        assert np.abs(getattr(m, k) - getattr(A, k)).max() <= 1e-11 * scale, k
    b, x = hecmw_dump.read_vector(DUMP + ".rhs"), hecmw_dump.read_vector(DUMP + ".sol")
    assert np.abs(b - A.B).max() <= 1e-11 * max(np.abs(A.B).max(), 1.0)
    assert np.abs(x - g["sol_m1_p3_t1_X"]).max() <= 1e-11 * np.abs(x).max()


def test_write_is_byte_identical_to_the_reference(tmp_path):
    """write_bsr(read_bsr(reference dump)) reproduces the reference's file byte for byte."""
    from frontistr_amd import hecmw_dump
    m = hecmw_dump.read_bsr(DUMP + ".bsr")
    out = str(tmp_path / "again.bsr")
    hecmw_dump.write_bsr(out, m)
    assert open(out).read() == open(DUMP + ".bsr").read()


@pytest.mark.gpu
def test_replay_dumped_system_on_gpu():
    """A system dumped by the reference replayed through hecmw_solve: the solution equals the dumped .sol."""
    from frontistr_amd import hecmw as hip, hecmw_dump
    m = hecmw_dump.read_bsr(DUMP + ".bsr")
    m.B[:] = hecmw_dump.read_vector(DUMP + ".rhs")
    m.Iarray[0] = 10000; m.Iarray[1] = 1; m.Iarray[2] = 3
    ctx = hip.SolverContext()
    assert hip.hecmw_solve(None, m, ctx=ctx) == 0
    x = hecmw_dump.read_vector(DUMP + ".sol")
    assert np.abs(m.X - x).max() < 1e-8 * np.abs(x).max()
    ctx.close()
