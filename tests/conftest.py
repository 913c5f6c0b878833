import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _allow_multistream(monkeypatch, request):
    """The multi-stream mode of the library is experimental and opt-in (include/soccdpt_hip.h, soccdpt_set_streams); the tests
    that exercise it opt in through the environment, everything else runs with the default (refused for n > 1)."""
    name = request.node.name
    if any(t in name for t in ("multi_stream", "hip_graph_replay", "back_to_back_modes")):
        monkeypatch.setenv("SOCCDPT_ALLOW_MULTISTREAM", "1")
