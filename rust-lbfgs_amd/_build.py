"""Build the native libraries in-tree (they travel to the GPU box with the snapshot).

  liblbfgs_hip.so     HIP kernels + the C-ABI of include/lbfgs_hip.h   (hipcc, gfx950)
  liblbfgs_solver.so  host orchestration of include/lbfgs_solver.h    (g++, links the former)

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off on both sides: the
reference's arithmetic is mul-then-add with separate roundings (src/math.rs:35,41).
"""
import glob
import os
import re
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HIP_LIB = os.path.join(HERE, "liblbfgs_hip.so")
SOLVER_LIB = os.path.join(HERE, "liblbfgs_solver.so")
RESOURCES = os.path.join(HERE, "liblbfgs_hip.resources.txt")  # per-kernel register / scratch usage of the last build

HIP_SRCS = [os.path.join(CSRC, f) for f in ("lbfgs_hip.hip", "ops.h", "stream.h", "gram.h", "lj.h", "resident.h", "gram_combine.h")] + [
    os.path.join(ROOT, "include", "lbfgs_hip.h")
]
SOLVER_SRCS = [os.path.join(CSRC, "host", "solver.cpp"), os.path.join(ROOT, "include", "lbfgs_solver.h"),
               os.path.join(ROOT, "include", "lbfgs_hip.h")]


def _stale(target, srcs):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def _run(cmd, cwd=None):
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=cwd)
    if r.returncode != 0:
        raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
    return r


def hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return "hipcc"


def _audit_agprs(asm_text):
    """resident.h and gram_combine.h keep data in AGPRs by hand (inline asm).  Every AGPR operand the COMPILER emitted in those kernels --
    anything outside an inline-asm block -- would overwrite that data.  Returns {kernel symbol: count}."""
    out = {}
    for m in re.finditer(r"^(_ZN2lh(?:24two_loop_resident_kernel|28gram_combine_resident_kernel|25gram_rows_resident_kernel)\w+):", asm_text, re.M):
        body = asm_text[m.end(): asm_text.index("s_endpgm", m.end())]
        in_asm, n = False, 0
        for line in body.splitlines():
            if "#ASMSTART" in line:
                in_asm = True
            elif "#ASMEND" in line:
                in_asm = False
            elif not in_asm and not line.lstrip().startswith(";") and re.search(r"\ba\[?\d", line):
                n += 1
        out[m.group(1)] = n
    return out


def build_hip(force=False):
    if force or _stale(HIP_LIB, HIP_SRCS):
        # -Rpass-analysis=kernel-resource-usage: registers / scratch / occupancy of every kernel, kept next to the library
        # (tests/test_abi_exports.py requires ScratchSize == 0 everywhere: a kernel that touches scratch memory pays
        # ~12 us of extra dispatch cost per launch on MI355X, measured).  -save-temps keeps the device assembly long
        # enough to audit the resident kernels' AGPR usage (same test).
        with tempfile.TemporaryDirectory(prefix="lbfgs_hip_build_") as tmp:
            r = _run([hipcc(), "-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17",
                      # (resident.h parks data in the accumulation registers by hand: the compiler must not spill into them)
                      "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-save-temps",
                      "-Wall", "-Rpass-analysis=kernel-resource-usage", HIP_SRCS[0], "-o", HIP_LIB, "-ldl"], cwd=tmp)
            asm = glob.glob(os.path.join(tmp, "*amdgcn*gfx950.s"))
            if len(asm) != 1:
                raise RuntimeError("build: expected one device assembly file, found %r" % (asm,))
            with open(asm[0]) as f:
                audit = _audit_agprs(f.read())
        if not audit:
            raise RuntimeError("build: no resident kernel found in the device assembly")
        with open(RESOURCES, "w") as f:
            f.write(r.stderr)
            f.write("\n# AGPR operands emitted by the compiler (outside inline asm) in the resident kernels\n")
            for k in sorted(audit):
                f.write("agpr-audit: %s %d\n" % (k, audit[k]))
    return HIP_LIB


def build_solver(force=False):
    build_hip(force)
    if force or _stale(SOLVER_LIB, SOLVER_SRCS + [HIP_LIB]):
        _run(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wall", SOLVER_SRCS[0], "-o",
              SOLVER_LIB, "-L" + HERE, "-llbfgs_hip", "-Wl,-rpath,$ORIGIN"])
    return SOLVER_LIB


def build_all(force=False):
    return build_hip(force), build_solver(force)


if __name__ == "__main__":
    print(build_all(force=True))
