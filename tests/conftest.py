import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def data(*parts):
    return os.path.join(ROOT, "data", *parts)


def load_system(mol, basis):
    import qchem_rs_amd as q
    b = q.BasisSet.load(data("basis", basis + ".json"))
    return q.MolecularSystem.load(data("mol", mol + ".json"), b)


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "integrals_golden.json")) as f:
        return json.load(f)
