import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU-side native helpers (oracle, vector factory) are cheap to build; make sure they exist
    need = [os.path.join(ROOT, "oracle", "libj2k_oracle.so"), os.path.join(ROOT, "tools", "vecgen", "libhtj2k_vecgen.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", ROOT, "oracle", "vecgen"])


@pytest.fixture(scope="session")
def orc():
    import oracle
    d = oracle.OracleDecoder()
    yield d
    d.close()
