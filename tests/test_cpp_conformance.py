"""Builds and runs tests/cpp/conformance.cpp: the reference's conformance suite written against
the C++ host mirror (zgml_amd/host/backend.hpp). The build runs everywhere; the run needs the GPU."""
import subprocess
from pathlib import Path

import pytest

from zgml_amd import capi

ROOT = Path(__file__).resolve().parent.parent
EXE = ROOT / "tests" / "cpp" / "_build" / "conformance"


def build_exe():
    EXE.parent.mkdir(parents=True, exist_ok=True)
    srcs = [ROOT / "tests" / "cpp" / "conformance.cpp", ROOT / "zgml_amd" / "host" / "hip_backend.cpp"]
    if not EXE.exists() or any(s.stat().st_mtime > EXE.stat().st_mtime for s in srcs):
        subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", "-I", str(ROOT / "include"), "-o", str(EXE), *map(str, srcs),
                        "-ldl"], check=True)
    return EXE


def test_conformance_cpp_builds():
    assert build_exe().exists()


@pytest.mark.gpu
def test_conformance_cpp_runs(oracle):
    exe = build_exe()
    r = subprocess.run([str(exe), str(capi.HIP_LIB_PATH), str(oracle.LIB_PATH)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "CONFORMANCE_OK" in r.stdout and "FAIL" not in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("PASS") == 12
