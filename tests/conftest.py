import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """-m gpu runs: import torch BEFORE the first test.  On a fresh box the first `import torch` pages the image in and can take
    minutes (one run of this round sat 6 minutes in dlopen); inside a test that is the per-test timeout's business (pytest.ini:
    400 s) and would fail a test that has nothing to do with it."""
    expr = session.config.getoption("-m") or ""
    if "gpu" in expr and "not gpu" not in expr:
        try:
            import torch  # noqa: F401
        except Exception:
            pass


def _build_if_missing():
    need_oracle = not os.path.exists(os.path.join(ROOT, "oracle", "_build", "liboracle.so"))
    need_native = not (os.path.exists(os.path.join(ROOT, "quack_amd", "libquack_hip.so"))
                       and os.path.exists(os.path.join(ROOT, "quack_amd", "libquack_host.so"))
                       and os.path.exists(os.path.join(ROOT, "quack_amd", "libquack_dropin.so"))
                       and os.path.exists(os.path.join(ROOT, "quack_amd", "host", "quack")))
    if need_oracle:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    if need_native:
        subprocess.check_call(["make", "-C", ROOT, "all"])


_build_if_missing()


@pytest.fixture(scope="session")
def root():
    return ROOT


@pytest.fixture(scope="session")
def inputs():
    return os.path.join(ROOT, "tests", "golden", "inputs")


def has_gpu():
    try:
        import quack_amd
        return quack_amd.device_count() > 0
    except Exception:
        return False
