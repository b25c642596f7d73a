import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


@pytest.fixture(scope="session", autouse=True)
def _built_artifacts():
    """The suite needs the HIP library (hipcc cross-compiles without a GPU) and the C oracle; build whichever is
    missing, exactly as __graft_entry__.build() does.  Building is not a fallback: a box without hipcc fails here."""
    import subprocess

    root = os.path.dirname(HERE)
    if not os.path.exists(os.path.join(root, "groth_sahai_rs_amd", "lib", "libgs_amd.so")):
        subprocess.check_call(["make", "-C", os.path.join(root, "groth_sahai_rs_amd", "csrc"), "ARCH=gfx950"])
    if not os.path.exists(os.path.join(root, "oracle", "libgs_ref_bls12_381.so")):
        subprocess.check_call(["make", "-C", os.path.join(root, "oracle")])
    yield
