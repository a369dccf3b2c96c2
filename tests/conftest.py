import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def flat_model():
    from oracle import urdf_model
    import wbc_quadruped_dob_amd as W
    return urdf_model.load_urdf(W.SYNTHETIC_URDF)


@pytest.fixture(scope="session")
def oracle(flat_model):
    from oracle import oracle_py
    return oracle_py.Oracle(flat_model)


@pytest.fixture(scope="session")
def golden():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz")))


@pytest.fixture(scope="session")
def hip_lib():
    """The product library, built in-tree; CPU tests only load it / use its host-side entry points."""
    import wbc_quadruped_dob_amd as W
    if not os.path.exists(W.LIB_PATH):
        W.build_library()
    return W.lib()


@pytest.fixture(scope="session")
def gpu_model(hip_lib):
    import wbc_quadruped_dob_amd as W
    return W.Model.from_urdf(W.SYNTHETIC_URDF)
