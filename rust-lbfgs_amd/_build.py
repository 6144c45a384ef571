"""Build the native libraries in-tree (they travel to the GPU box with the snapshot).

  liblbfgs_hip.so     HIP kernels + the C-ABI of include/lbfgs_hip.h   (hipcc, gfx950; three translation units:
                      csrc/context.hip, csrc/lbfgs_hip.hip, csrc/lj.hip, sharing csrc/ctx.h)
  liblbfgs_solver.so  host orchestration of include/lbfgs_solver.h    (g++, links the former)

hipcc cross-compiles for gfx950 without a GPU.  -ffp-contract=off on both sides: the
reference's arithmetic is mul-then-add with separate roundings (src/math.rs:35,41).

Builds are CONTENT-ADDRESSED: a library carries the hash of the sources and flags it was compiled from
(`lbfgs_hip_build_id()`, `lbfgs_solver_build_id()`; the string is also readable from the file), and it is rebuilt
whenever that differs from the hash of the checked-out sources.  Modification times play no part.
"""
import glob
import hashlib
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

# the library's translation units (compiled one by one, in parallel, then linked) ...
HIP_UNITS = [os.path.join(CSRC, f) for f in ("context.hip", "lbfgs_hip.hip", "lj.hip")]
# ... and everything the build id covers: the units, the headers they share, the public header
HIP_SRCS = HIP_UNITS + [os.path.join(CSRC, f) for f in ("ctx.h", "ops.h", "stream.h", "gram.h", "lj.h", "resident.h", "gram_combine.h", "ext_protocol.h")] + [
    os.path.join(ROOT, "include", "lbfgs_hip.h")
]
SOLVER_SRCS = [os.path.join(CSRC, "host", "solver.cpp"), os.path.join(ROOT, "include", "lbfgs_solver.h"),
               os.path.join(ROOT, "include", "lbfgs_hip.h")]


def source_hash(paths, flags=()):
    """16 hex digits over the CONTENTS of `paths` (in the order given, names included) and the compiler flags."""
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    for fl in flags:
        h.update(fl.encode() + b"\0")
    return h.hexdigest()[:16]


def embedded_id(lib, marker):
    """The build id a library carries (the string `<marker>=<16 hex digits>` inside the file), or None.  Reading the file
    instead of loading it: the answer must not depend on a HIP runtime being present."""
    try:
        with open(lib, "rb") as f:
            m = re.search(re.escape(marker.encode()) + rb"=([0-9a-f]{16})", f.read())
    except OSError:
        return None
    return m.group(1).decode() if m else None


HIP_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
             # (resident.h parks data in the accumulation registers by hand: the compiler must not spill into them)
             "-mllvm", "-amdgpu-spill-vgpr-to-agpr=0", "-Wall"]
SOLVER_FLAGS = ["-O2", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wall"]


def hip_build_id():
    """What liblbfgs_hip.so must carry to belong to the checked-out sources."""
    return source_hash(HIP_SRCS, HIP_FLAGS)


def solver_build_id():
    return source_hash(SOLVER_SRCS, SOLVER_FLAGS + [hip_build_id()])


def _stale(target, marker, want):
    """A library is stale when the id it carries is not the hash of the sources -- never by modification time: prebuilt,
    git-ignored libraries travel with snapshots and checkouts that reset mtimes."""
    return embedded_id(target, marker) != want


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


LAST_BUILD = {}  # name -> "compiled" | "reused" (what build_all did this time; __graft_entry__.build() prints it)


def _compile_hip(out_lib, flags, build_id):
    """Compile every unit of HIP_UNITS (in parallel: one hipcc -c each, in a directory of its own so that -save-temps keeps
    the device assembly apart) and link them into `out_lib`.  -> (the compiler's per-kernel resource report, the AGPR audit).
    -Rpass-analysis=kernel-resource-usage: registers / scratch / occupancy of every kernel (tests/test_abi_exports.py requires
    ScratchSize == 0 everywhere: a kernel that touches scratch memory pays ~12 us of extra dispatch cost per launch on
    MI355X, measured).  -save-temps keeps the device assembly long enough to audit the resident kernels' AGPR usage."""
    from concurrent.futures import ThreadPoolExecutor

    with tempfile.TemporaryDirectory(prefix="lbfgs_hip_build_") as tmp:
        def one(unit):
            d = os.path.join(tmp, os.path.basename(unit) + ".d")
            os.makedirs(d)
            obj = os.path.join(d, "unit.o")
            r = _run([hipcc(), *flags, "-c", "-save-temps", "-Rpass-analysis=kernel-resource-usage",
                      '-DLBFGS_HIP_BUILD_ID="%s"' % build_id, unit, "-o", obj], cwd=d)
            asm = glob.glob(os.path.join(d, "*amdgcn*gfx950.s"))
            if len(asm) != 1:
                raise RuntimeError("build: expected one device assembly file for %s, found %r" % (unit, asm))
            with open(asm[0]) as f:
                return obj, r.stderr, f.read()

        with ThreadPoolExecutor(max_workers=len(HIP_UNITS)) as ex:
            results = list(ex.map(one, HIP_UNITS))
        _run([hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", *[o for o, _, _ in results], "-o", out_lib, "-ldl", "-lrt"])
        report = "".join(err for _, err, _ in results)
        audit = {}
        for _, _, asm in results:
            audit.update(_audit_agprs(asm))
    return report, audit


def build_hip(force=False):
    want = hip_build_id()
    if force or _stale(HIP_LIB, "LBFGS_HIP_BUILD_ID", want) or not os.path.exists(RESOURCES):
        report, audit = _compile_hip(HIP_LIB, HIP_FLAGS, want)
        if not audit:
            raise RuntimeError("build: no resident kernel found in the device assembly")
        with open(RESOURCES, "w") as f:
            f.write(report)
            f.write("\n# AGPR operands emitted by the compiler (outside inline asm) in the resident kernels\n")
            for k in sorted(audit):
                f.write("agpr-audit: %s %d\n" % (k, audit[k]))
        if embedded_id(HIP_LIB, "LBFGS_HIP_BUILD_ID") != want:
            raise RuntimeError("build: %s does not carry the id it was built with" % HIP_LIB)
        LAST_BUILD["liblbfgs_hip.so"] = "compiled"
    else:
        LAST_BUILD.setdefault("liblbfgs_hip.so", "reused")
    return HIP_LIB


def build_solver(force=False):
    build_hip(force)
    want = solver_build_id()
    if force or _stale(SOLVER_LIB, "LBFGS_SOLVER_BUILD_ID", want):
        _run(["g++", *SOLVER_FLAGS, '-DLBFGS_SOLVER_BUILD_ID="%s"' % want, SOLVER_SRCS[0], "-o",
              SOLVER_LIB, "-L" + HERE, "-llbfgs_hip", "-Wl,-rpath,$ORIGIN"])
        LAST_BUILD["liblbfgs_solver.so"] = "compiled"
    else:
        LAST_BUILD["liblbfgs_solver.so"] = "reused"
    return SOLVER_LIB


def build_all(force=False):
    return build_hip(force), build_solver(force)


def build_variant(name, extra_flags):
    """A/B builds (tools/build_variants.sh): both libraries compiled with extra -D flags into tools/bin/variants/<name>/
    (load them with LBFGS_HIP_LIB_DIR), with the same resource report and AGPR audit as the in-tree build; raises if a
    kernel of the variant touches scratch memory or the compiler put something into an accumulation register."""
    out = os.path.join(ROOT, "tools", "bin", "variants", name)
    os.makedirs(out, exist_ok=True)
    hip_lib, solver_lib = os.path.join(out, "liblbfgs_hip.so"), os.path.join(out, "liblbfgs_solver.so")
    flags = HIP_FLAGS + list(extra_flags)
    bid = source_hash(HIP_SRCS, flags)
    report, audit = _compile_hip(hip_lib, flags, bid)
    with open(os.path.join(out, "liblbfgs_hip.resources.txt"), "w") as f:
        f.write(report)
        for k in sorted(audit):
            f.write("agpr-audit: %s %d\n" % (k, audit[k]))
    scratch = [ln for ln in report.splitlines() if "ScratchSize" in ln and not ln.rstrip().endswith(": 0 [-Rpass-analysis=kernel-resource-usage]")]
    bad = {k: v for k, v in audit.items() if v}
    if (scratch or bad) and not os.environ.get("LBFGS_VARIANT_ALLOW_BAD"):  # (experiments that will not RUN the bad kernels)
        raise RuntimeError("variant %s: scratch %r, compiler-placed AGPR operands %r" % (name, scratch[:3], bad))
    _run(["g++", *SOLVER_FLAGS, SOLVER_SRCS[0], "-o", solver_lib, "-L" + out, "-llbfgs_hip", "-Wl,-rpath,$ORIGIN"])
    return out


if __name__ == "__main__":
    import sys

    if len(sys.argv) >= 3 and sys.argv[1] == "--variant":  # python rust-lbfgs_amd/_build.py --variant NAME -DX=1 -DY=2
        print("built", build_variant(sys.argv[2], sys.argv[3:]))
    else:
        print(build_all(force=True))
