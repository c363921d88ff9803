import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The suite's small problems exist to exercise K1 / K3 (the per-launch path); the library's default would hand every dense
# one-GPU problem of n <= 2048 to the LDS-resident kernel instead.  tests/test_gpu_resident.py covers that kernel and asks
# for it explicitly (or removes this variable again where the default choice itself is under test).
os.environ.setdefault("CGX_RESIDENT", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip: a silent skip would read as "parity green".
    pass


@pytest.fixture(scope="session")
def oracle():
    import __graft_entry__ as g
    O = g.load_oracle()
    O.build()
    return O


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    return g.load_package()


@pytest.fixture(scope="session")
def gpu_pkg(pkg):
    """The package with torch imported first (so libcgx binds to the HIP runtime torch loaded) and a
    hard requirement that the HIP library and an MI355X are there: no fallback, no skip."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X; libcgx has no CPU fallback"
    pkg.cgx.lib()
    return pkg


@pytest.fixture(scope="session")
def reference_probe():
    return json.load(open(os.path.join(GOLDEN, "reference_probe.json")))


@pytest.fixture(scope="session")
def oracle_large():
    return json.load(open(os.path.join(GOLDEN, "oracle_large.json")))


@pytest.fixture(scope="session")
def mtx_path():
    return os.path.join(GOLDEN, "lap2D_5pt_n100.mtx")
