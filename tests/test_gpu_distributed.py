"""N > 1 path of the HIP library on ONE GPU: 2 (and 4) ranks share cuda:0 and exchange halos /
reduce scalars through the host-callback transport (gloo), so the library's pack/unpack,
import/export handling, nn_internal dot products and reduction placement are checked against
the serial oracle.  (RCCL itself refuses two ranks on one device; its calls are the only part of
the multi-GPU path this test cannot reach.)"""
import numpy as np
import pytest

from test_distributed import check_against_serial, dist_prefix, run_world, serial_cube4, serial_reference

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,m,meth,pc", [(2, 6, 1, 3), (4, 5, 1, 3), (2, 6, 2, 3)])
def test_hip_distributed_block_jacobi_equals_serial(oracle, tmp_path, world, m, meth, pc):
    dims = {2: (2, 1, 1), 4: (2, 2, 1)}[world]
    res = run_world("hip", world, m, meth, pc, tmp_path)
    ser = serial_reference(oracle, dims, m, meth, pc)
    check_against_serial(res, ser, meth)


def test_hip_distributed_ssor_matches_distributed_oracle(oracle, tmp_path):
    """Localized multicolour SSOR: same decomposition through the oracle (CPU, gloo) and through
    the HIP library -> same iteration count and field."""
    hipres = run_world("hip", 2, 6, 1, 1, tmp_path)
    (tmp_path / "o").mkdir()
    orcres = run_world("oracle", 2, 6, 1, 1, tmp_path / "o")
    for a, b in zip(hipres, orcres):
        assert abs(int(a["it"]) - int(b["it"])) <= 1
        assert np.abs(a["X"] - b["X"]).max() < 1e-8 * np.abs(b["X"]).max()
        k = min(10, len(a["hist"]), len(b["hist"]))
        assert np.all(np.abs(a["hist"][:k] - b["hist"][:k]) <= 1e-9 * b["hist"][:k])


def test_rccl_allreduce_path_single_rank(tmp_path):
    """RCCL binding (dlopen), ncclCommInitRank and the in-stream ncclAllReduce of the scalar stage,
    exercised with a 1-rank communicator (FX_FORCE_COMM=1) in a fresh process."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import numpy as np
from conftest import golden_matrix, load_golden
from frontistr_amd import hecmw as hip
g = load_golden('cube4'); A = golden_matrix(g)
m = hip.hecmwST_matrix.from_arrays(A.N, A.NP, A.indexL, A.itemL, A.indexU, A.itemU, A.D, A.AL, A.AU, A.B.copy())
m.Iarray[0] = 10000; m.Iarray[2] = 1
ctx = hip.SolverContext()
ctx.comm_init(hip.comm_unique_id(), 0, 1)
assert hip.hecmw_solve(None, m, ctx=ctx) == 0
assert abs(ctx.info.iterations - int(g['sol_m1_p1_t4_iter'])) <= 1
assert np.abs(m.X - g['sol_m1_p1_t4_X']).max() < 1e-8 * np.abs(g['sol_m1_p1_t4_X']).max()
print('rccl ok', ctx.info.iterations)
""" % (ROOT, ROOT)
    env = dict(os.environ, FX_FORCE_COMM="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert p.returncode == 0 and "rccl ok" in p.stdout, p.stdout[-3000:]


def test_hip_on_reference_partitioner_files(oracle, tmp_path):
    """configs[3] plumbing: 4 ranks read the HECMW-DIST files hecmw_part1 wrote and solve through the
    library; the field equals the serial solve of the undecomposed mesh."""
    res = run_world("hip", 4, "dist:" + dist_prefix(), 1, 3, tmp_path)
    check_against_serial(res, serial_cube4(oracle, 1, 3), 1)
