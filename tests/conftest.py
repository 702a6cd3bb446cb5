import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture
def switches(monkeypatch):
    """Flip an environment switch of the library inside this process: the library reads its switch table once
    (csrc/sf_switches.h), so the variable is set AND the table re-read; both are undone after the test."""
    import streamformer_amd._native as nat

    def set_switch(name, value="1"):
        monkeypatch.setenv(name, str(value))
        nat.lib.sf_reload_switches()
    yield set_switch
    monkeypatch.undo()
    nat.lib.sf_reload_switches()
