import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The built libraries are git-ignored: on a checkout that has none yet, compile them first (hipcc cross-compiles
    # gfx950 without a GPU; same recipe as __graft_entry__.build()).  This builds the HIP library -- it is not a
    # fallback for it: without hipcc the tests fail loudly on the missing library.
    so = os.path.join(ROOT, "voicecontrolledrobot-var_amd", "libvar_hip.so")
    orc = os.path.join(ROOT, "oracle", "libvar_oracle.so")
    if not (os.path.exists(so) and os.path.exists(orc)):
        import subprocess
        try:
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "voicecontrolledrobot-var_amd", "csrc"), "-j8", "-s"])
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
        except (OSError, subprocess.CalledProcessError) as e:       # report through the tests that need the library
            print(f"conftest: building the libraries failed: {e}", file=sys.stderr)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
