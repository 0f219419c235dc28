import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on the CPU oracle (test infrastructure); built on demand with gcc."""
    import oracle_ctypes
    return oracle_ctypes.load()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(autouse=True)
def inputs_ready_before_device_calls(request, monkeypatch):
    """A context decodes on a stream of its own (unless it was given the caller's), which is not
    ordered against the stream torch prepares the tests' inputs on (fills, clones, generators):
    for the GPU tests every device-pointer call of a context on its own stream first waits for the
    device, the way a caller of the C ABI must before it hands over buffers other streams wrote."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    from orc_format_amd import context

    def waiting(method):
        def call(self, *args, **kwargs):
            if not self.caller_stream:
                torch.cuda.synchronize()
            return method(self, *args, **kwargs)
        return call
    for name in dir(context.Context):
        if name.endswith("_device"):
            monkeypatch.setattr(context.Context, name, waiting(getattr(context.Context, name)))
    yield
