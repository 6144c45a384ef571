"""`trait LbfgsMath` (src/math.rs:4-29) on device-resident vectors.

    y.vecadd(x, c); y.vecdot(x); y.vecscale(c); y.veccpy(x); y.vecncpy(x); z.vecdiff(x, y);
    y.vec2norm(); y.vec2norminv()

Method names and argument order are the trait's.  Reductions return host floats (one
stream synchronisation each); `*_slot` variants leave the result on the device board.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .api import Context, _dp

_SCRATCH_SLOT = 255  # board slot used by the host-returning reductions


class DeviceVec:
    """One shard of an f64 n-vector in HBM (lbfgs_hip_vec)."""

    def __init__(self, ctx: Context, data=None):
        self.ctx = ctx
        self._L = ctx._L
        self._h = C.c_void_p()
        ctx.check(self._L.lbfgs_hip_vec_alloc(ctx._h, C.byref(self._h)))
        if data is not None:
            self.upload(data)

    def __len__(self):
        return self.ctx.n_local

    def free(self):
        if self._h:
            self._L.lbfgs_hip_vec_free(self._h)
            self._h = C.c_void_p()

    def upload(self, data):
        a = np.ascontiguousarray(data, dtype=np.float64)
        self.ctx.check(self._L.lbfgs_hip_vec_upload(self._h, _dp(a), len(a)))
        return self

    def to_numpy(self):
        out = np.zeros(self.ctx.n_local)
        self.ctx.check(self._L.lbfgs_hip_vec_download(self._h, _dp(out), len(out)))
        return out

    def fill(self, value):
        self.ctx.check(self._L.lbfgs_hip_vec_fill(self._h, float(value)))
        return self

    @property
    def ptr(self):
        return self._L.lbfgs_hip_vec_ptr(self._h)

    # ---- LbfgsMath ------------------------------------------------------------------
    def vecadd(self, x: "DeviceVec", c: float):
        """y += c*x  (math.rs:33)"""
        self.ctx.check(self._L.lbfgs_hip_vecadd(self._h, x._h, float(c)))

    def vecadd_slot(self, x: "DeviceVec", c_slot: int):
        self.ctx.check(self._L.lbfgs_hip_vecadd_dev(self._h, x._h, c_slot))

    def vecdot(self, other: "DeviceVec") -> float:
        """s = x.dot(y)  (math.rs:40)"""
        self.vecdot_slot(other, _SCRATCH_SLOT)
        return float(self.ctx.scalars(_SCRATCH_SLOT)[0])

    def vecdot_slot(self, other: "DeviceVec", slot: int):
        self.ctx.check(self._L.lbfgs_hip_vecdot(self._h, other._h, slot))

    def veccpy(self, x: "DeviceVec"):
        """y = x  (math.rs:52)"""
        self.ctx.check(self._L.lbfgs_hip_veccpy(self._h, x._h))

    def vecncpy(self, x: "DeviceVec"):
        """y = -x  (math.rs:59)"""
        self.ctx.check(self._L.lbfgs_hip_vecncpy(self._h, x._h))

    def vecdiff(self, x: "DeviceVec", y: "DeviceVec"):
        """z = x - y  (math.rs:66)"""
        self.ctx.check(self._L.lbfgs_hip_vecdiff(self._h, x._h, y._h))

    def vecscale(self, c: float):
        """y *= c  (math.rs:45)"""
        self.ctx.check(self._L.lbfgs_hip_vecscale(self._h, float(c)))

    def vec2norm(self) -> float:
        """||x||  (math.rs:73-76): sqrt of the device dot"""
        out = C.c_double()
        self.ctx.check(self._L.lbfgs_hip_vec2norm(self._h, _SCRATCH_SLOT, C.byref(out)))
        return out.value

    def vec2norminv(self) -> float:
        """1/||x||  (math.rs:79-81)"""
        out = C.c_double()
        self.ctx.check(self._L.lbfgs_hip_vec2norminv(self._h, _SCRATCH_SLOT, C.byref(out)))
        return out.value
