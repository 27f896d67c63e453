import os
import sys

import pytest

try:  # torch must initialise ITS bundled HIP runtime before libclip.so pulls in /opt/rocm's (same soname): the other
    import torch  # noqa: F401  order leaves torch without a visible GPU in this process
    if torch.cuda.is_available():
        torch.cuda.init()
except Exception:  # torch is optional for everything except the device-pointer test
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CACHE = os.environ.get("CLIP_AMD_FIXTURE_CACHE", "/tmp/clip_amd_fixtures")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fixture_cache():
    os.makedirs(CACHE, exist_ok=True)
    return CACHE


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import ref
    ref.build()
    return ref


@pytest.fixture(scope="session")
def clip_lib():
    import clip_cpp_amd
    clip_cpp_amd.build_lib()
    return clip_cpp_amd


@pytest.fixture(scope="session")
def host_only_env():
    """Allow clip_model_load to return a host-only context on machines without a GPU (CPU test tier)."""
    os.environ["CLIP_AMD_ALLOW_NO_DEVICE"] = "1"
    yield
