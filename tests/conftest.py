import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HOSTEMU = os.path.join(ROOT, "tests", "hostemu", "libocnhip_hostemu.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # One native library per process.  `-m gpu` runs use the product (libocnhip.so, HIP); every other
    # run uses the host emulation of the same kernel sources (tests/hostemu, built by build()).
    expr = (config.getoption("-m") or "").strip()
    use_gpu = expr == "gpu" or os.environ.get("OCNHIP_TEST_BACKEND") == "gpu"
    config._ocn_backend = "gpu" if use_gpu else "hostemu"
    if use_gpu:
        os.environ.pop("OCNHIP_LIB", None)
        alt = os.environ.get("OCNHIP_TEST_LIB")      # kernel experiments: another GPU build of the same sources
        if alt and "hostemu" not in alt:
            os.environ["OCNHIP_LIB"] = os.path.abspath(alt)
    else:
        os.environ["OCNHIP_LIB"] = HOSTEMU
        # The CPU suite is ~540 small, independent runs (oracle, host emulation, 2-3 process gloo / shm jobs with their own free
        # ports and pid-named mailboxes): spread them over four pytest-xdist workers unless the caller chose -n / -p no:xdist.
        # `-m gpu` runs stay in ONE process (one context on the card, and the driver records which .so that process loaded).
        opt = config.option
        if (config.pluginmanager.hasplugin("xdist") and getattr(opt, "numprocesses", None) is None
                and not hasattr(config, "workerinput") and not getattr(opt, "collectonly", False)
                and not getattr(opt, "usepdb", False) and os.environ.get("OCNHIP_TEST_SERIAL") != "1"):
            opt.numprocesses = min(4, os.cpu_count() or 1)
            opt.dist = "load"
            opt.tx = ["popen"] * opt.numprocesses


@pytest.fixture(scope="session")
def backend(request):
    return request.config._ocn_backend


@pytest.fixture(scope="session")
def ocn(request):
    """the host-side package bound to the backend of this run"""
    import __graft_entry__ as ge
    if request.config._ocn_backend == "hostemu" and not os.path.exists(HOSTEMU):
        import subprocess
        subprocess.check_call(["make", "-C", ge.CSRC, "emu"])
    return ge.load_package()


def pytest_collection_modifyitems(config, items):
    skip_gpu = pytest.mark.skip(reason="needs the HIP backend: run with -m gpu on an MI355X")
    if config._ocn_backend != "gpu":
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip_gpu)
