import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def robot_model():
    from locomanipulationrl_amd.model.robot_model import load_model
    return load_model("quadruped_robot_v2")


@pytest.fixture(scope="session")
def vertical_model():
    from locomanipulationrl_amd.model.robot_model import load_model
    return load_model("quadfinger")


GOLDEN = os.path.join(ROOT, "tests", "golden")
