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


@pytest.fixture(scope="session", autouse=True)
def _library_options_from_the_environment():
    """`GSR_OPTS=name=value,...` puts a non-default route of the library under the whole run (3dgs_hierarchical_training_amd/_lib.py
    applies it at load): the library is loaded here, once, so that the options stand before the first test's first render."""
    if os.environ.get("GSR_OPTS"):
        import importlib
        importlib.import_module("3dgs_hierarchical_training_amd._lib").load()
    yield
