"""MI355X-native L-BFGS / OWL-QN inner loop behind the API of ybyygu/rust-lbfgs (crate `liblbfgs`).

    from rust_lbfgs_amd import lbfgs
    report = lbfgs().with_max_iterations(5).minimize(x, evaluate, progress)

The n-dimensional vectors live in HBM; the BLAS-1 primitives, the two-loop recursion, the
line-step updates and the OWL-QN operators are hand-written HIP kernels for gfx950 behind the
C-ABI of include/lbfgs_hip.h.  There is no CPU fallback: without the built extension and a GPU
the package raises.
"""
from . import _build, _ffi
from .api import (BuiltinObjective, Context, DeviceEvaluate, TorchEvaluate, Lbfgs, LbfgsError, LbfgsPanic, LbfgsState, Progress,
                  Report, default_evaluate, default_progress, lbfgs)


def build(force=False):
    """Compile liblbfgs_hip.so (hipcc, gfx950) and liblbfgs_solver.so (g++) in-tree."""
    return _build.build_all(force)


def __getattr__(name):  # lazy submodules (dist imports torch)
    if name in ("math", "objectives", "dist", "hotpath", "problem"):
        import importlib

        return importlib.import_module(f"{__name__}.{name}")
    raise AttributeError(name)


__all__ = ["lbfgs", "Lbfgs", "LbfgsState", "Progress", "Report", "LbfgsError", "LbfgsPanic", "Context",
           "DeviceEvaluate", "TorchEvaluate", "BuiltinObjective", "default_evaluate", "default_progress", "build"]
