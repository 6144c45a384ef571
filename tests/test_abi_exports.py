"""The C-ABI libraries load without a GPU and export every symbol include/*.h declares (no compute calls)."""
import ctypes as C
import os
import re

import pytest

import rust_lbfgs_amd as R
from rust_lbfgs_amd import _build, _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"\w+)\s*\(", txt)) - {"lbfgs_hip_allreduce_cb"})


@pytest.fixture(scope="module")
def libs():
    hip, solver = R.build()
    return C.CDLL(hip, mode=C.RTLD_LOCAL), C.CDLL(solver, mode=C.RTLD_LOCAL)


def test_hip_library_exports_every_declared_symbol(libs):
    names = declared("lbfgs_hip.h", "lbfgs_hip_")
    assert len(names) >= 45
    for nm in names:
        assert hasattr(libs[0], nm), nm
    assert sorted(_ffi.HIP_SYMBOLS) == names  # the Python binding covers the whole header


def test_solver_library_exports_every_declared_symbol(libs):
    names = [n for n in declared("lbfgs_solver.h", "lbfgs_") if not n.startswith("lbfgs_hip_")]
    names = [n for n in names if not n.endswith("_cb")]
    for nm in names:
        assert hasattr(libs[1], nm), nm
    assert sorted(_ffi.SOLVER_SYMBOLS) == sorted(names)


def test_no_cpu_fallback_without_gpu(libs):
    """Without a GPU the product fails loudly instead of computing on the host."""
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present")
    prev, _ffi._LIB = _ffi._LIB, None
    try:
        with pytest.raises(R.LbfgsError) as e:
            R.Context(16)
        assert e.value.code == _ffi.HIP_ERR_NO_DEVICE
    finally:
        _ffi._LIB = prev


def test_product_libraries_do_not_link_the_oracle(libs):
    """Neither product .so may contain or depend on oracle symbols."""
    import subprocess

    for lib in (_build.HIP_LIB, _build.SOLVER_LIB):
        syms = subprocess.run(["nm", "-D", lib], capture_output=True, text=True).stdout
        assert "oracle_" not in syms
        needed = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True).stdout
        assert "oracle" not in needed


def test_abi_version(libs):
    assert libs[0].lbfgs_hip_abi_version() == 1
