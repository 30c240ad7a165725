import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONFIGS = [(1, 3, 1), (1, 1, 1), (1, 1, 4), (1, 10, 1), (2, 3, 1), (2, 1, 1), (2, 1, 4), (2, 10, 1)]


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def golden_matrix(g, tag="ic_"):
    from oracle.refrun import BSR
    return BSR(int(g[tag + "N"]), int(g[tag + "NP"]), g[tag + "indexL"], g[tag + "itemL"],
               g[tag + "indexU"], g[tag + "itemU"], g[tag + "D"], g[tag + "AL"], g[tag + "AU"], g[tag + "B"])


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def hip():
    from frontistr_amd import hecmw
    return hecmw
