"""A compiled, non-Python caller of the C-ABI: tests/support/c_caller.c (plain C, gcc, links liblbfgs_solver.so +
liblbfgs_hip.so) runs the reference's own end-to-end tests -- tests/simple.rs:16-54 (Rosenbrock N = 100, then OWL-QN
from the converged x), tests/simple.rs:57-83 (Booth) and the src/math.rs:84-122 primitive known answers -- with C
callbacks as the closures.  It stands in for the Rust shim (integration/rust-shim), which cannot be compiled in this
image (no rustc)."""
import os
import subprocess

import pytest

from rust_lbfgs_amd import _build
from tests.support import mock

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "support", "c_caller.c")


def _build_caller(tmp_path, libdir, libs):
    exe = str(tmp_path / "c_caller")
    cmd = ["gcc", "-std=c11", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-L", libdir, *libs,
           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_c_caller_logic_on_the_test_double(tmp_path):
    """The same C program linked against the CPU test double of the C-ABI: every check of the program passes (this
    validates the program; it is not a product path)."""
    lib = mock.build()
    exe = _build_caller(tmp_path, os.path.dirname(lib), ["-l:" + os.path.basename(lib)])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_caller: all checks passed" in r.stdout


def test_c_caller_links_the_product_and_fails_loudly_without_a_gpu(tmp_path):
    if os.path.exists("/dev/kfd"):
        pytest.skip("GPU present: see the gpu test below")
    _build.build_all()
    exe = _build_caller(tmp_path, os.path.dirname(_build.HIP_LIB), ["-llbfgs_solver", "-llbfgs_hip"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77 and "no CPU fallback" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_caller_on_the_gpu(tmp_path):
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    exe = _build_caller(tmp_path, os.path.dirname(_build.HIP_LIB), ["-llbfgs_solver", "-llbfgs_hip"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_caller: all checks passed" in r.stdout
    # tests/simple.rs:37-40, :52-54 were asserted by the program; echo its lines into the test log
    print(r.stdout)


def test_the_hip_caller_compiles_for_gfx950(tmp_path):
    """tests/support/device_closure_caller.hip (a caller with kernels of its own, deferred trial points) cross-compiles
    against the headers: the GPU test below builds and runs it."""
    obj = str(tmp_path / "device_closure_caller.o")
    r = subprocess.run([_build.hipcc(), "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-std=c++17", "-Wall", "-I",
                        os.path.join(ROOT, "include"), "-c", os.path.join(ROOT, "tests", "support", "device_closure_caller.hip"),
                        "-o", obj], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and os.path.getsize(obj) > 0, r.stderr


@pytest.mark.gpu
def test_a_caller_with_its_own_hip_kernels_and_deferred_trials(tmp_path):
    """tests/support/device_closure_caller.hip: LBFGS_EVAL_DEVICE with the caller's own kernels as `evaluate`, and the
    deferred-trial callbacks of ABI version 4 (device_probe / device_accept: line.rs:283-288 reads f and g.d only) --
    the three runs (full evaluations / probes + accept / probes only) take the same discrete decisions, agree to 1e-10 per
    iteration and call the closure as often as lbfgs_solver.h says.  Built by hipcc here, on the GPU box."""
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU (the program launches kernels)")
    import json

    exe = str(tmp_path / "device_closure_caller")
    libdir = os.path.dirname(_build.HIP_LIB)
    cmd = [_build.hipcc(), "--offload-arch=gfx950", "-O2", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "support", "device_closure_caller.hip"), "-L", libdir, "-llbfgs_solver", "-llbfgs_hip",
           "-Wl,-rpath," + libdir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "4000000", "40"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "device_closure_caller: all checks passed" in r.stdout
    j = json.loads(r.stdout.splitlines()[0])
    print(json.dumps(j))
    assert j["worst_relative_deviation"] <= 1e-10 and j["iterations"] == 40
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        json.dump(j, open(os.path.join(out, "device_closure_caller.json"), "w"), indent=1)
    except OSError:
        pass


# ------------------------------------------------------------------------------------------------------------------
# include/lbfgs.hpp: the crate's public API for C++ callers; tests/support/cpp_api_tests.cpp = the reference's own tests on it
# ------------------------------------------------------------------------------------------------------------------
CPP_SRC = os.path.join(ROOT, "tests", "support", "cpp_api_tests.cpp")


def _build_cpp(tmp_path, libdir, libs):
    exe = str(tmp_path / "cpp_api_tests")
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), CPP_SRC, "-L", libdir, *libs,
           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_cpp_api_mirror_on_the_test_double(tmp_path):
    """include/lbfgs.hpp (lbfgs().with_*().minimize(x, evaluate, progress), build / propagate, Progress, Report, the
    setters' assert!s, Err propagation) through the reference's own tests -- tests/simple.rs:16-83, tests/owlqn.rs:6-63 with
    its CSV fixtures -- linked against the CPU test double of the C-ABI (a logic check of header and program)."""
    lib = mock.build()
    exe = _build_cpp(tmp_path, os.path.dirname(lib), ["-l:" + os.path.basename(lib)])
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    assert "cpp_api_tests: all checks passed" in r.stdout and "test_owlqn: fx = -42724.136705" in r.stdout


@pytest.mark.gpu
def test_cpp_api_mirror_on_the_gpu(tmp_path):
    if os.environ.get("LBFGS_TEST_BACKEND") == "mock":
        pytest.skip("needs the GPU")
    exe = _build_cpp(tmp_path, os.path.dirname(_build.HIP_LIB), ["-llbfgs_solver", "-llbfgs_hip"])
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr
    assert "cpp_api_tests: all checks passed" in r.stdout
    print("\n".join(ln for ln in r.stdout.splitlines() if ln.startswith("test_") or ln.startswith("cpp_api")))
