import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import vk_renderer_amd  # noqa: E402,F401  (shim: registers the hyphenated package)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _oracle():
    """The checker: oracle/binding.py builds the library if needed and registers the "oracle" backend with the
    flat chain driver.  The product package itself knows nothing about it."""
    from oracle import binding

    return binding.install(build_if_missing=True)


@pytest.fixture(scope="session")
def oracle_lib():
    return _oracle()


@pytest.fixture(scope="session", autouse=True)
def _oracle_backend_registered():
    _oracle()


@pytest.fixture
def parity_table(request):
    """Collects every [parity] row of the test and writes it to gpurun_out/parity_<test>.json (gpurun merges that
    directory back; the tables quoted in DESIGN.md are copied from there into profiles/)."""
    import json
    import re

    import parity

    parity.ROWS.clear()
    yield parity.ROWS
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        name = re.sub(r"[^A-Za-z0-9_.-]+", "_", request.node.name)
        with open(os.path.join(out, f"parity_{name}.json"), "w") as f:
            json.dump({"test": request.node.nodeid, "rows": list(parity.ROWS)}, f, indent=1)
    except OSError:
        pass
