"""A plain C program against include/emdee_hip.h (tests/c/abi_client.c): what a Julia `ccall` binding does, with no
Python or torch in the process that talks to the library.  Without a GPU the client must stop at
emdee_ctx_create with the library's own error; on the GPU it restates the reference's test and checks the O(N)
operator and 50 velocity-Verlet steps against the C oracle."""
import os
import subprocess

import pytest

from .conftest import ROOT


def _build(tmp_path):
    exe = str(tmp_path / "abi_client")
    lib_dir, orc_dir = os.path.join(ROOT, "emdee.jl_amd"), os.path.join(ROOT, "oracle")
    if not os.path.exists(os.path.join(orc_dir, "libemdee_oracle.so")):
        subprocess.run(["make", "-C", orc_dir], check=True, capture_output=True)
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", orc_dir,
           os.path.join(ROOT, "tests", "c", "abi_client.c"), "-o", exe, "-L", lib_dir, "-lemdee_hip", "-L", orc_dir,
           "-lemdee_oracle", "-lm", "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + orc_dir]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_client_builds_and_fails_loudly_without_a_device(tmp_path):
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu-marked test")
    exe = _build(tmp_path)
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "lj_sample.xyz")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2
    assert "emdee_ctx_create" in r.stderr and "no CPU path" in r.stderr


@pytest.mark.gpu
def test_c_client_against_oracle(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "lj_sample.xyz")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout and "FAILED" not in r.stdout
    assert "gfx950" in r.stdout
