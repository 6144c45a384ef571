"""AddressSanitizer + UndefinedBehaviorSanitizer over the host side: the product's host
orchestration (rust-lbfgs_amd/csrc/host/solver.cpp, compiled unchanged), the CPU test double of the C-ABI and the
oracle are rebuilt with `-O1 -g -fsanitize=address,undefined` and the CPU suites that drive them are run once more on
those builds (LD_PRELOAD of the sanitizer runtimes, because the interpreter itself is not instrumented).  GPU code
cannot be sanitised on this pool; the device side is covered by the parity tests."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_host_logic_and_oracle_under_asan_ubsan():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("gcc sanitizer runtimes not installed")
    env = dict(os.environ, LBFGS_SANITIZE="1", LD_PRELOAD=f"{asan}:{ubsan}", OMP_NUM_THREADS="1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66",   # CPython itself "leaks" at exit
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=67")
    env.pop("LBFGS_TEST_BACKEND", None)
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_oracle_golden.py",
           "tests/test_host_logic_cpu.py"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    out = p.stdout + p.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out, out[-4000:]
    assert p.returncode == 0, out[-4000:]
    assert " passed" in p.stdout
    # the sanitised libraries are the ones that were loaded
    assert os.path.exists(os.path.join(ROOT, "tests", "support", "liblbfgs_solver_mock_asan.so"))
    assert os.path.exists(os.path.join(ROOT, "oracle", "liblbfgs_oracle_asan.so"))
