import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import pkload  # noqa: E402

pk = pkload.load()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size oracle runs (tens of seconds)")


@pytest.fixture(scope="session")
def orc():
    import oracle
    if (os.cpu_count() or 1) >= 4:
        oracle.set_threads(4)
    return oracle


@pytest.fixture(scope="session")
def tiny_cfg():
    return pk.make_tiny_config()


@pytest.fixture(scope="session")
def tiny_weights(tiny_cfg):
    from parakeet_cpp_amd import synth
    return synth.synth_weights(tiny_cfg, seed=42)


@pytest.fixture(scope="session")
def tiny_oracle(orc, tiny_cfg, tiny_weights):
    return orc.Model(tiny_cfg, tiny_weights)
