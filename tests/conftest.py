import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # build what is missing (the in-tree .so files normally travel with the snapshot)
    if not os.path.exists(os.path.join(ROOT, "lesseq_amd", "_build", "liblesseq_hip.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "lesseq_amd", "csrc"), "-j8"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "_build", "liblsq_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-j4"])


def has_gpu():
    try:
        import lesseq_amd as L
        c = L.Context(0)
        c.close()
        return True
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_ctx():
    import lesseq_amd as L
    c = L.Context(0)      # fails loudly when there is no gfx950 device
    yield c
    c.close()
