import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.build()
    return oracle_lib


@pytest.fixture(scope="session")
def hiplib():
    """The HIP library must already be built in-tree (the GPU box receives the prebuilt .so)."""
    from smcsmc_amd import build, pf
    if not os.path.exists(pf.LIB_PATH):
        build.build_lib()
    return pf.load_library()


@pytest.fixture(scope="session", autouse=True)
def built_binary():
    """bin/smcsmc is a build product (git-ignored): every session builds it from the sources it is about to test
    (make compares time stamps, so this is a no-op when it is current) and fails -- not skips -- when that fails."""
    from smcsmc_amd import build
    build.build_all()
    path = os.path.join(ROOT, "bin", "smcsmc")
    assert os.path.exists(path), "bin/smcsmc was not produced by smcsmc_amd.build.build_all()"
    return path
