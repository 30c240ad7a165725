"""N > 1 path of the HIP library on ONE GPU: 2 (and 4) ranks share cuda:0 and exchange halos /
reduce scalars through the host-callback transport (gloo), so the library's pack/unpack,
import/export handling, nn_internal dot products and reduction placement are checked against
the serial oracle.  (RCCL itself refuses two ranks on one device; its calls are the only part of
the multi-GPU path this test cannot reach.)"""
import numpy as np
import pytest

from test_distributed import check_against_serial, dist_prefix, run_world, serial_cube4, serial_reference

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,m,meth,pc", [(2, 6, 1, 3), (4, 5, 1, 3), (2, 6, 2, 3), (2, 6, 3, 3), (2, 6, 4, 3)])
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


def test_hip_distributed_nonlinear_equals_serial_oracle(oracle, tmp_path):
    """The nonlinear static loop (elastoplastic, updated Lagrange) on 2 subdomains: halo handling of
    dunode / QFORCE / the residual and the nn_internal norms, against the serial oracle loop on the
    undecomposed mesh (block-Jacobi CG, so the linear solves are decomposition independent)."""
    from dist_worker import NL_MATERIAL, nl_bc
    from oracle import refrun
    m, dims = 5, (2, 1, 1)
    res = run_world("hipnl", 2, m, 1, 3, tmp_path)
    G = (dims[0] * m, dims[1] * m, dims[2] * m)
    kk, jj, ii = np.meshgrid(np.arange(G[2]), np.arange(G[1]), np.arange(G[0]), indexing="ij")
    coord = np.stack([ii.ravel(), jj.ravel(), kk.ravel()], axis=1).astype(float)
    ek, ej, ei = np.meshgrid(np.arange(G[2] - 1), np.arange(G[1] - 1), np.arange(G[0] - 1), indexing="ij")
    n0 = (1 + ei + G[0] * (ej + G[1] * ek)).ravel()
    s1, s2 = G[0], G[0] * G[1]
    conn = np.stack([n0, n0 + 1, n0 + 1 + s1, n0 + s1, n0 + s2, n0 + 1 + s2, n0 + 1 + s1 + s2, n0 + s1 + s2], axis=1).astype(np.int32)
    bottom = (np.nonzero(coord[:, 2] == 0)[0] + 1).astype(np.int32)
    clamp = (np.repeat(bottom, 3).astype(np.int32), np.tile(np.array([1, 2, 3], dtype=np.int32), bottom.size), np.zeros(3 * bottom.size))
    bc = nl_bc(coord, clamp, G[2] - 1)
    mat = refrun.Material(*NL_MATERIAL[0], plastic=True, harden=0, plconst=NL_MATERIAL[1]["plconst"], nlgeom=2)
    I, R = refrun.default_params(method=1, precond=3, tol=1e-10, iterlog=0, timelog=0)
    model = oracle.NonlinearModel(mat, coord, conn)
    olog = model.run_steps(*bc, np.zeros(3 * coord.shape[0]), 2, 4, 1e-12, I, R, nthreads=2)
    assert model.state["plstrain"].max() > 1e-3
    us = model.unode.reshape(-1, 3)
    for r in res:
        assert int(r["it"]) == olog.shape[0] == 8
        assert np.abs(r["X"].reshape(-1, 3) - us[r["gid"]]).max() < 1e-8 * np.abs(us).max()      # internal and halo nodes
        np.testing.assert_allclose(r["hist"], olog[:, 3], rtol=1e-6)                               # |residual| per Newton iteration
