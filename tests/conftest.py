"""pytest configuration: marker registration, repo-root imports, shared fixtures."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def read_xyz(path):
    """Minimal XYZ reader (stands in for Chemfiles.Trajectory, test/runtests.jl:20-21)."""
    with open(path) as fh:
        n = int(fh.readline())
        fh.readline()
        return np.array([[float(t) for t in fh.readline().split()[1:4]] for _ in range(n)])


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def emdee():
    """The product package (directory 'emdee.jl_amd', imported under the alias emdee_jl_amd)."""
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def lj_sample():
    """The reference's own hot-path fixture: 800 atoms, Float32-rounded as CUDA.cu does (test/runtests.jl:22)."""
    return read_xyz(os.path.join(GOLDEN, "lj_sample.xyz")).astype(np.float32)


@pytest.fixture(scope="session")
def golden():
    return {name: np.load(os.path.join(GOLDEN, name + ".npz"))
            for name in ("lj_sample_expected", "fcc864_expected", "mix500_expected")}


@pytest.fixture(scope="session")
def emdee_synthetic():
    """The package's numpy-only synthetic-box generator, loaded without touching the HIP library."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("emdee_jl_amd_synthetic",
                                                  os.path.join(ROOT, "emdee.jl_amd", "synthetic.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
