import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope="session")
def short_manifest():
    return json.load(open(os.path.join(GOLDEN, "short_state_manifest.json")))


@pytest.fixture(scope="session")
def long_manifest():
    return json.load(open(os.path.join(GOLDEN, "long_state_manifest.json")))


@pytest.fixture(scope="session")
def short_sd(short_manifest):
    from emip_amd.filler import state_dict_from_manifest
    return state_dict_from_manifest(short_manifest, 0)


@pytest.fixture(scope="session")
def long_sd(long_manifest):
    from emip_amd.filler import state_dict_from_manifest
    return state_dict_from_manifest(long_manifest, 0)


@pytest.fixture(scope="session")
def model_args():
    return json.load(open(os.path.join(GOLDEN, "model_args.json")))
