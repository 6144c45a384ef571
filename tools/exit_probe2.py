import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
mode = sys.argv[1]
if "torchfirst" in mode:
    import torch
    torch.ones(2, device="cuda:0").sum().item()
import numpy as np
import rust_lbfgs_amd as R
from rust_lbfgs_amd import _ffi
from rust_lbfgs_amd.dist import CommSpec
from rust_lbfgs_amd.math import DeviceVec
L = _ffi.load()
if "rccl" in mode:
    _ffi.torch_before_rccl()
    buf = (C.c_char * 128)()
    assert L.lbfgs_hip_rccl_unique_id(buf) == 0
    n = 1000
    ctx = R.Context(n, shard=_ffi.Shard(0, 1, n, 0, n), comm=CommSpec(_ffi.COMM_RCCL, unique_id=buf))
    x = DeviceVec(ctx); x.fill(2.0); print("dot", x.vecdot(x)); x.free()
    if "noclose" not in mode:
        ctx.close()
if "torchafter" in mode:
    import torch
    print("torch", torch.ones(2, device="cuda:0").sum().item())
print("done", mode, os.environ.get("LBFGS_HIP_RCCL"))
