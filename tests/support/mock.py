"""Build / load the CPU test double of the C-ABI (TEST INFRASTRUCTURE).

tests/support/liblbfgs_solver_mock.so = the product's host orchestration
(rust-lbfgs_amd/csrc/host/solver.cpp, compiled unchanged) linked against
tests/support/mock_lbfgs_hip.cpp (the C-ABI restated on the CPU from the oracle's
primitives) instead of the HIP library.  Used only by the `-m "not gpu"` suite.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
# LBFGS_SANITIZE=1 (tests/test_sanitizers_cpu.py): everything below compiled with -fsanitize=address,undefined
SANITIZE = os.environ.get("LBFGS_SANITIZE") == "1"
LIB = os.path.join(HERE, "liblbfgs_solver_mock_asan.so" if SANITIZE else "liblbfgs_solver_mock.so")
SRCS = [
    os.path.join(ROOT, "rust-lbfgs_amd", "csrc", "host", "solver.cpp"),
    os.path.join(HERE, "mock_lbfgs_hip.cpp"),
]
C_SRCS = [os.path.join(ROOT, "oracle", "lbfgs_oracle.c"), os.path.join(ROOT, "oracle", "objectives.c")]
HDRS = [os.path.join(ROOT, "include", "lbfgs_hip.h"), os.path.join(ROOT, "include", "lbfgs_solver.h"),
        os.path.join(ROOT, "oracle", "lbfgs_oracle.h"), os.path.join(ROOT, "rust-lbfgs_amd", "csrc", "ext_protocol.h")]


def _build_id(flags):
    """content hash of everything the test double is compiled from (same scheme as rust-lbfgs_amd/_build.py)"""
    import hashlib

    h = hashlib.sha256()
    for p in SRCS + C_SRCS + HDRS:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for fl in flags:
        h.update(fl.encode() + b"\0")
    return h.hexdigest()[:16]


def _embedded_id(lib):
    import re

    try:
        with open(lib, "rb") as f:
            m = re.search(rb"LBFGS_MOCK_BUILD_ID=([0-9a-f]{16})", f.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


def build(force=False):
    objs = []
    flags = ["-O2", "-ffp-contract=off", "-fPIC", "-Wall"]
    if SANITIZE:
        flags = ["-O1", "-g", "-ffp-contract=off", "-fPIC", "-Wall", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"]
    want = _build_id(flags)
    if not force and _embedded_id(LIB) == want:  # (by content, never by modification time)
        return LIB
    flags = flags + ['-DLBFGS_MOCK_BUILD_ID="%s"' % want]
    for s in C_SRCS:
        o = os.path.join(HERE, os.path.basename(s) + (".asan.o" if SANITIZE else ".o"))
        subprocess.run(["gcc", "-std=c11", *flags, "-c", s, "-o", o], check=True, capture_output=True)
        objs.append(o)
    r = subprocess.run(["g++", "-std=c++17", *flags, "-shared", *SRCS, *objs, "-o", LIB, "-lm", "-lrt", "-Wl,-Bsymbolic"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return LIB


def load():
    """-> ctypes library with the declarations of rust_lbfgs_amd._ffi attached."""
    import rust_lbfgs_amd  # noqa: F401  (registers the package alias)
    from rust_lbfgs_amd import _ffi

    return _ffi.declare(C.CDLL(build()))


def install():
    """Point the Python binding at the test double (tests only; the product has no hook for this: we patch the
    module attribute).  Returns the previous library for `restore`."""
    import rust_lbfgs_amd  # noqa: F401
    from rust_lbfgs_amd import _ffi

    prev = _ffi._LIB
    _ffi._LIB = load()
    return prev


def restore(prev):
    from rust_lbfgs_amd import _ffi

    _ffi._LIB = prev
