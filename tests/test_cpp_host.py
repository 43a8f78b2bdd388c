"""C++ host mirror (pvw_rs_amd/host/pvw.hpp): compiles against include/pvw_hip.h everywhere;
on the GPU box the round-trip program is run against libpvw_hip.so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "host_roundtrip.cpp")
EXE = os.path.join(ROOT, "build", "host_roundtrip")
LIBDIR = os.path.join(ROOT, "pvw_rs_amd")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", SRC, "-o", EXE, "-L" + LIBDIR, "-lpvw_hip",
                           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"])


def test_cpp_host_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_host_round_trip():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "CPP_HOST_OK" in out.stdout, out.stdout + out.stderr
