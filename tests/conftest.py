import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pytorch-unsup-pc_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def native_library():
    """The tests bind libdpc_render.so; a checkout that has not been built yet -- or whose library is older than a source
    it is built from -- is (re)built here (hipcc cross-compiles for gfx950 without a GPU, a few minutes) instead of failing
    on the first test that loads it, or passing on stale kernels."""
    lib = os.path.join(PKG, "csrc", "libdpc_render.so")
    csrc = os.path.join(PKG, "csrc")
    sources = [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")) or f == "Makefile"]
    sources.append(os.path.join(ROOT, "include", "dpc_render.h"))
    stale = os.path.exists(lib) and any(os.path.getmtime(f) > os.path.getmtime(lib) for f in sources)
    if (stale or not os.path.exists(lib)) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess

        # a library older than its sources would be tested silently (same ABI number, different kernels): make decides
        subprocess.run(["make", "-C", csrc, "-j3"], check=True, stdout=subprocess.DEVNULL)
    return lib


@pytest.fixture(scope="session", autouse=True)
def two_rank_runs(request, native_library, tmp_path_factory):
    """The two-rank GPU runs of tests/test_multi_gpu.py, started HERE -- at the start of the session, before any test has
    initialised the GPU in this process (a process that has must not start other programs on the GPU boxes) -- and only when
    that test is among the selected ones and the node has a GPU.  {backend: [result files] | error text}."""
    wanted = any("test_two_ranks_render_their_shards" in item.nodeid for item in request.session.items)
    if not wanted:
        return {}
    import torch

    if torch.cuda.device_count() < 1:   # counting devices does not initialise the GPU
        return {}
    import test_multi_gpu

    return test_multi_gpu.launch_ranks(str(tmp_path_factory.mktemp("two_ranks")))


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name)))
        return cache[name]

    return load
