import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def model():
    from kbot_joystick_amd.spec import compiler
    return compiler.load_model("kbot-headless")


@pytest.fixture(scope="session")
def model_full():
    from kbot_joystick_amd.spec import compiler
    return compiler.load_model("kbot")


@pytest.fixture()
def quiet_cfg():
    """Config with every stochastic feature off (deterministic physics KATs)."""
    from kbot_joystick_amd.spec import layout
    return layout.default_config(num_envs=4, enable_randomizers=0, enable_noise=0, enable_pushes=0, command_mode=1,
                                 drop_action_prob=0.0)
