"""N > 1 path of the HIP library on ONE GPU: 2 (and 4) ranks share cuda:0 and exchange halos /
reduce scalars through the host-callback transport (gloo), so the library's pack/unpack,
import/export handling, nn_internal dot products and reduction placement are checked against
the serial oracle.  (RCCL itself refuses two ranks on one device; its calls are the only part of
the multi-GPU path this test cannot reach.)"""
import numpy as np
import pytest

from test_distributed import check_against_serial, run_world, serial_reference

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
