import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cref():
    """The C oracle (oracle/momref.c), built on demand with gcc."""
    from oracle import cref as c
    c.lib()
    return c


@pytest.fixture(scope="session")
def rtamd():
    import subprocess
    import rtamd as pkg
    if not pkg._lib.LIB_PATH.exists():  # normally built in-tree by __graft_entry__.build(); hipcc is in the image
        subprocess.check_call(["make", "-C", str(ROOT / "radiativetransfer.jl_amd" / "csrc"), "-j8"])
    return pkg
