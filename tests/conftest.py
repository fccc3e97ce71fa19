import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def synthetic_sd():
    from vits_amd import synthetic
    return synthetic.synthetic_state_dict()


@pytest.fixture(scope="session")
def oracle_model(synthetic_sd):
    from oracle.mvts_oracle import OracleSynthesizer
    return OracleSynthesizer(synthetic_sd)
