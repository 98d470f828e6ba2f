import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import vk_renderer_amd  # noqa: E402,F401  (shim: registers the hyphenated package)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    from vk_renderer_amd import abi

    if not os.path.exists(abi.ORACLE_LIB):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-j8"])
    return abi.oracle()
