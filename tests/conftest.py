import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_bank():
    return np.load(os.path.join(GOLDEN, "bank.npz"))


@pytest.fixture(scope="session")
def golden_signals():
    z = np.load(os.path.join(GOLDEN, "signals.npz"))
    return {"signals": z["signals"], "snr": z["snr"], "sparsity": z["sparsity"],
            "atoms": json.loads(str(z["atoms_json"]))}


@pytest.fixture(scope="session")
def oracle_dict():
    from oracle import csc_oracle
    return csc_oracle.canonical_dictionary()
