"""bench.py's own launcher (`python bench.py --gpus N` without torch.distributed.run): host logic only.
On a box without a GPU the rank processes must fail loudly and the parent must report that -- never a silent
single-rank result.  The GPU side (2 / 4 ranks sharing one device through the gloo rehearsal transport, n_gpus from the
transport's own rank count) is tests/test_gpu_distributed.py::test_bench_self_launch."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_and_world_size_must_agree():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode != 0
    assert "must agree" in r.stderr
    assert r.stdout.strip() == ""


def test_self_launch_fails_loudly_without_gpus():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu test")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--elems", "6"],
                       env=_env(FX_BENCH_LAUNCH_TIMEOUT="240"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    assert "failed" in r.stderr and "needs a GPU" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]      # no JSON line, no wrong answer


def test_decomposition_and_parser():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.decomposition(1) == (1, 1, 1) and bench.decomposition(2) == (2, 1, 1)
    assert bench.decomposition(4) == (2, 2, 1) and bench.decomposition(8) == (2, 2, 2)
    a = bench.parse_args([])
    assert a.gpus == 1 and a.n == 149 and a.method == 1 and a.precond == 1
    # 7.009 GB per SpMV at 10.125M DOF
    assert bench.spmv_algorithmic_bytes(3375000, 89915392) == 7009069800


def test_ledger_cross_check_names_the_rank_that_differs():
    """bench.py's cross-rank check of the communication ledgers (fx_comm_ledger): consistent ledgers pass; a rank with one more
    all-reduce, or a message nobody received, is named."""
    import bench

    def led(ops, h, ar, peers):
        return {"ops": ops, "seq_hash": h, "allreduces": ar, "allreduce_bytes": 16 * ar, "halos": ops - ar, "own_halo_comm": True, "peers": peers}

    good = [led(10, 77, 6, {1: (4, 960, 4, 480)}), led(10, 77, 6, {0: (4, 480, 4, 960)})]
    chk = bench.check_ledgers(good)
    assert chk["errors"] == [] and chk["allreduces"] == 6 and chk["halo_exchanges"] == 4 and chk["halo_bytes_sent_per_rank"] == [960, 480]
    bad = [led(10, 77, 6, {1: (4, 960, 4, 480)}), led(11, 78, 7, {0: (4, 480, 4, 960)})]
    e = bench.check_ledgers(bad)["errors"]
    assert any("rank 1 ops = 11" in x for x in e) and any("seq_hash" in x for x in e)
    lost = [led(10, 77, 6, {1: (5, 1200, 4, 480)}), led(10, 77, 6, {0: (4, 480, 4, 960)})]
    e = bench.check_ledgers(lost)["errors"]
    assert e and "rank 0 -> 1" in e[0]
    alone = [led(10, 77, 6, {1: (4, 960, 4, 480)}), led(10, 77, 6, {})]
    assert any("does not list it" in x for x in bench.check_ledgers(alone)["errors"])
