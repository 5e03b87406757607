import importlib
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def load_pkg():
    """Imports the hyphen-named package `openvla-oft_amd` and aliases it as `ovla_amd`."""
    if "ovla_amd" not in sys.modules:
        pkg = importlib.import_module("openvla-oft_amd")
        sys.modules["ovla_amd"] = pkg
    return sys.modules["ovla_amd"]


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def ops():
    load_pkg()
    return importlib.import_module("openvla-oft_amd.ops")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ops_mod = importlib.import_module("openvla-oft_amd.ops")
    ops_mod.check_device(0)  # fails loudly on anything that is not gfx950
    return torch.device("cuda:0")
