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


# A native abort (glibc's heap checks, std::terminate, an assertion inside the HIP runtime) prints its reason on file
# descriptor 2 and kills the process: under pytest's fd capture that text lands in a temporary file that dies with it
# (round 2 lost the one piece of evidence of such an abort that way).  The call phase of every GPU test therefore runs
# with fd 2 appended to a log that survives: gpurun_out/native_stderr.log when that directory can be made (gpurun carries
# it back), else the system's temporary directory; every test writes a header line first.  faulthandler (pytest's own
# plugin) adds the Python stack to the same descriptor.  (A hook wrapper, innermost: pytest's capture re-points fd 2 at
# the start of every phase, so a fixture's redirection would be undone before the test body runs.)
@pytest.hookimpl(hookwrapper=True, trylast=True)
def pytest_runtest_call(item):
    if item.get_closest_marker("gpu") is None or os.environ.get("LSQ_KEEP_STDERR") == "0":
        yield
        return
    import tempfile
    fd = None
    for d in (os.path.join(ROOT, "gpurun_out"), tempfile.gettempdir()):
        try:
            os.makedirs(d, exist_ok=True)
            fd = os.open(os.path.join(d, "native_stderr.log"), os.O_WRONLY | os.O_CREAT | os.O_APPEND, 0o644)
            break
        except OSError:
            fd = None
    if fd is None:
        yield
        return
    sys.stderr.flush()
    saved = os.dup(2)
    os.write(fd, ("=== %s\n" % item.nodeid).encode())
    os.dup2(fd, 2)
    os.close(fd)
    try:
        yield
    finally:
        sys.stderr.flush()
        os.dup2(saved, 2)
        os.close(saved)
