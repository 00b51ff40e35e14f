import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "spark-data-repair-plugin_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A wedged collective / kernel must fail its test instead of hanging the whole run (pytest-timeout, if installed)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900 if item.get_closest_marker("gpu") else 600))


@pytest.fixture
def oracle_backend():
    """Routes repair.gbm to the CPU oracle for the duration of a (CPU-only) test."""
    from repair import gbm
    from tests.helpers import OracleBackend
    prev = gbm.set_backend(OracleBackend)
    yield OracleBackend
    gbm.set_backend(prev)


@pytest.fixture(autouse=True)
def _testing_env(monkeypatch):
    monkeypatch.setenv("REPAIR_TESTING", "1")   # bad option values raise (reference: SPARK_TESTING)
