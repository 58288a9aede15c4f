import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load_oracle()


@pytest.fixture(scope="session")
def reflib():
    import oracle_lib
    lib = oracle_lib.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref/libltrans_ref.so not built (needs /root/reference + flang)")
    return lib
