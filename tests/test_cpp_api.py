"""include/pixo.hpp — the C++ mirror of the reference's Rust API — compiled with g++ against the
C-ABI library and run: option/builder semantics and every validation error on CPU; a whole
encode compared with the oracle on the GPU box."""
import os
import subprocess

import pytest

import oracle_lib as O
import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_pixo_hpp")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "test_pixo_hpp.cpp")
    lib = os.path.join(ROOT, "pixo_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", EXE, src, "-L" + lib, "-lpixo_hip",
                           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])


def test_cpp_header_semantics_and_errors():
    _build()
    r = subprocess.run([EXE], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


@pytest.mark.gpu
def test_cpp_header_encode_matches_oracle(tmp_path):
    _build()
    out = tmp_path / "cpp.jpg"
    r = subprocess.run([EXE, "gpu", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    want = O.encode(synth.noise(200, 120, 9), O.make_options(200, 120, O.RGB, 80, O.S420))
    assert out.read_bytes() == want
