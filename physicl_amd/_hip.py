"""ctypes binding of libphysicl_hip.so (include/physicl_hip.h) -- the only way Python reaches the GPU.

There is no CPU fallback: if the library is missing, or a call fails, this module raises.
"""
import ctypes
import os
from ctypes import POINTER, byref, c_char_p, c_double, c_int, c_int32, c_int64, c_uint8, c_uint32, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libphysicl_hip.so")

# enums of include/physicl_hip.h
R0, R1, R2, V0, V1, V2, DR0, DR1, DR2, DV0, DV1, DV2, E, NFIELDS = range(14)
FIELD_GROUPS = {"r": (R0, R1, R2), "v": (V0, V1, V2), "dr": (DR0, DR1, DR2), "dv": (DV0, DV1, DV2)}
SCATTER_WAVELENGTH, SCATTER_VARIABLE_N, FUSED_LAZY, SCATTER_PY_DV = 1, 2, 4, 8
RNG_INPUT, RNG_PHILOX = 0, 1
PHASE_ISOTROPIC, PHASE_DELETE = 0, 1
KIND_OBJECT, KIND_PHOTON = 0, 1
DTYPE_F64, DTYPE_F32 = 0, 1
_NP_DTYPE = {DTYPE_F64: np.float64, DTYPE_F32: np.float32}
CNT_N, CNT_XP, CNT_YP, CNT_ZP, CNT_PLANE0 = 0, 1, 2, 3, 4
MAX_PLANES = 12
PROF_NEWTON, PROF_SCATTER, PROF_DELETE_MASK, PROF_COMPACT, PROF_COUNTERS, PROF_FUSED, PROF_MULTI, PROF_ONEPASS, \
    PROF_DELETE_AHEAD = range(9)
PROF_NAMES = {PROF_NEWTON: "k_newton", PROF_SCATTER: "k_scatter", PROF_DELETE_MASK: "k_delete_mask",
              PROF_COMPACT: "k_compact", PROF_COUNTERS: "k_counters", PROF_FUSED: "k_fused", PROF_MULTI: "k_multi",
              PROF_ONEPASS: "k_delete_onepass", PROF_DELETE_AHEAD: "k_delete_ahead"}
ERR_NAMES = {-1: "PCL_ERR_HIP", -2: "PCL_ERR_ARG", -3: "PCL_ERR_STATE", -4: "PCL_ERR_RTC", -5: "PCL_ERR_EXPR",
             -6: "PCL_ERR_NOMEM"}


class HipError(RuntimeError):
    """A libphysicl_hip call returned a negative code; message is pcl_last_error()."""

    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "PCL_ERR"), code, msg))
        self.code = code


class ExpressionError(HipError, ValueError):
    """variable_n_fn was rejected (validator or hipRTC compile)."""


_dp = POINTER(c_double)
_vp = c_void_p

# name -> argtypes ; restype is always int except pcl_last_error
_PROTOTYPES = {
    "pcl_abi_version": [],
    "pcl_device_count": [POINTER(c_int)],
    "pcl_set_knob": [c_char_p, c_char_p],
    "pcl_comm_unique_id": [_vp],
    "pcl_comm_create": [_vp, c_char_p, c_int, c_int, POINTER(_vp)],
    "pcl_comm_allreduce_sum_i64": [_vp, _vp, c_int],
    "pcl_comm_info": [_vp, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int64)],
    "pcl_comm_destroy": [_vp],
    "pcl_store_last_multi_work": [_vp, POINTER(c_int64), POINTER(c_int64), POINTER(c_int), POINTER(c_int64)],
    "pcl_store_last_multi_clock": [_vp, POINTER(c_double)],
    "pcl_store_last_mixed_rows": [_vp, POINTER(c_int)],
    "pcl_store_ahead_clock": [_vp, POINTER(c_double)],
    "pcl_store_last_multi_hist": [_vp, _vp],
    "pcl_store_ahead_stats": [_vp, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)],
    "pcl_store_ahead_work": [_vp, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)],
    "pcl_store_alloc_info": [_vp, POINTER(c_int), POINTER(c_double), c_int, POINTER(c_double)],
    "pcl_ctx_set_rtc_background": [_vp, c_int],
    "pcl_ctx_rtc_wait": [_vp, POINTER(c_int)],
    "pcl_pool_trim": [POINTER(c_int64)],
    "pcl_pool_bytes": [POINTER(c_int64)],
    "pcl_pool_info": [POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_int)],
    "pcl_ctx_create": [c_int, _vp, POINTER(_vp)],
    "pcl_ctx_destroy": [_vp],
    "pcl_ctx_sync": [_vp],
    "pcl_ctx_stream": [_vp, POINTER(_vp)],
    "pcl_ctx_device_info": [_vp, c_char_p, c_int, POINTER(c_int64), POINTER(c_int), POINTER(c_int)],
    "pcl_ctx_device_pci": [_vp, c_char_p, c_int],
    "pcl_ctx_mem_info": [_vp, POINTER(c_int64), POINTER(c_int64)],
    "pcl_dev_alloc": [_vp, c_int64, POINTER(_vp)],
    "pcl_dev_free": [_vp, _vp],
    "pcl_h2d": [_vp, _vp, _vp, c_int64],
    "pcl_d2h": [_vp, _vp, _vp, c_int64],
    "pcl_dev_memset": [_vp, _vp, c_int, c_int64],
    "pcl_prof_enable": [_vp, c_int],
    "pcl_prof_read": [_vp, c_int, POINTER(c_int64), POINTER(c_double), POINTER(c_double), POINTER(c_double)],
    "pcl_timer_start": [_vp],
    "pcl_timer_stop": [_vp, POINTER(c_double)],
    "pcl_k_light_scatter_step_del": [_vp, _vp, _vp, _vp, _vp, c_double, c_double, _vp, c_int64],
    "pcl_k_scatter_delete_test": [_vp, _vp, _vp, _vp, _vp, c_double, c_double, _vp, c_int64],
    "pcl_k_light_scatter_step_sphere": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, c_double, c_double, _vp, _vp, _vp, _vp,
                                        _vp, _vp, _vp, c_int64, c_int, c_double, c_double, c_char_p],
    "pcl_k_compact_indices": [_vp, _vp, c_int64, _vp, POINTER(c_int64)],
    "pcl_expr_validate": [c_char_p],
    "pcl_user_kernel_build": [_vp, c_char_p, c_char_p, c_char_p, POINTER(_vp)],
    "pcl_user_kernel_launch": [_vp, _vp, c_int64, _vp, c_int64],
    "pcl_user_kernel_free": [_vp, _vp],
    "pcl_store_alloc": [_vp, c_int64],
    "pcl_store_alloc_dtype": [_vp, c_int64, c_int],
    "pcl_store_dtype": [_vp, POINTER(c_int)],
    "pcl_store_free": [_vp],
    "pcl_store_capacity": [_vp, POINTER(c_int64)],
    "pcl_store_count": [_vp, POINTER(c_int64)],
    "pcl_store_set_count": [_vp, c_int64, c_int64],
    "pcl_store_slots": [_vp, POINTER(c_int64), POINTER(c_int)],
    "pcl_store_reserve_compaction": [_vp],
    "pcl_store_upload": [_vp, c_int, _vp, c_int64, c_int64],
    "pcl_store_download": [_vp, c_int, _vp, c_int64, c_int64],
    "pcl_store_upload_ids": [_vp, _vp, c_int64, c_int64],
    "pcl_store_download_ids": [_vp, _vp, c_int64, c_int64],
    "pcl_store_upload_kind": [_vp, _vp, c_int64, c_int64],
    "pcl_store_download_kind": [_vp, _vp, c_int64, c_int64],
    "pcl_store_field_ptr": [_vp, c_int, POINTER(_vp)],
    "pcl_store_layout": [_vp, POINTER(c_int64), POINTER(c_int64)],
    "pcl_store_upload_rand": [_vp, c_int, _vp, c_int64],
    "pcl_store_upload_rand3": [_vp, _vp, c_int64, c_int64],
    "pcl_store_fill_photons": [_vp, c_int64, c_int64, c_double, c_double, c_double, c_uint64],
    "pcl_store_fill_photons_table": [_vp, c_int64, c_int64, c_double, _vp, _vp, c_int, c_uint64],
    "pcl_step_newton": [_vp, c_double],
    "pcl_step_scatter_isotropic": [_vp, c_double, c_double, c_int, c_double, c_double, c_char_p, c_int, c_uint64,
                                   c_uint32, POINTER(c_int64)],
    "pcl_step_fused_delete_multi": [_vp, c_double, c_int, c_double, c_double, c_uint64, c_uint32, _vp, c_int, _vp],
    "pcl_step_scatter_pcoll": [_vp, c_double, c_double, c_int, c_double, c_double, _vp],
    "pcl_step_delete_flags": [_vp, _vp, POINTER(c_int64), POINTER(c_int64)],
    "pcl_step_plane_energies": [_vp, _vp, _vp, c_int64, POINTER(c_int64)],
    "pcl_store_is_uniform": [_vp, POINTER(c_int)],
    "pcl_step_fused_multi": [_vp, c_double, c_int, c_double, c_double, c_int, c_double, c_double, c_char_p, c_uint64,
                             c_uint32, _vp, c_int, _vp],
    "pcl_step_fused": [_vp, c_double, c_int, c_double, c_double, c_int, c_double, c_double, c_char_p, c_int, c_uint64,
                       c_uint32, _vp, c_int, _vp],
    "pcl_step_mixed_multi": [_vp, c_double, c_int, c_int, _vp, c_double, c_double, c_int, c_double, c_double, c_char_p,
                             c_double, c_double, c_uint64, c_uint32, _vp, c_int, _vp],
    "pcl_store_trace_ahead": [_vp, _vp, c_int, c_double, c_int, c_int, _vp, c_int, c_double, c_double, c_int, c_double, c_double,
                              c_char_p, c_double, c_double, c_uint64, c_uint32, _vp],
    "pcl_store_trace_read": [_vp, _vp, c_int64],
    "pcl_step_fused_read": [_vp, c_int, _vp],
    "pcl_store_last_scatter_hits": [_vp, POINTER(c_int64)],
    "pcl_step_scatter_delete": [_vp, c_double, c_double, c_int, c_uint64, c_uint32, POINTER(c_int64),
                                POINTER(c_int64)],
    "pcl_step_fused_delete": [_vp, c_double, c_double, c_double, c_int, c_int, c_uint64, c_uint32, _vp, c_int, _vp],
    "pcl_store_last_delete_flags": [_vp, _vp, c_int64],
    "pcl_step_counters": [_vp, _vp, c_int, _vp],
    # device groups: several GPUs from one process (the C-level counterpart of physicl_amd.multidev.MultiDevice)
    "pcl_group_create": [c_int, POINTER(c_int), POINTER(_vp)],
    "pcl_group_destroy": [_vp],
    "pcl_group_size": [_vp, POINTER(c_int)],
    "pcl_group_ctx": [_vp, c_int, POINTER(_vp)],
    "pcl_group_shard": [_vp, c_int64, c_int, POINTER(c_int64), POINTER(c_int64)],
    "pcl_group_store_alloc": [_vp, c_int64, c_int],
    "pcl_group_store_dtype": [_vp, POINTER(c_int)],
    "pcl_group_fill_photons": [_vp, c_int64, c_int64, c_double, c_double, c_double, c_uint64],
    "pcl_group_count": [_vp, POINTER(c_int64)],
    "pcl_group_sync": [_vp],
    "pcl_group_reserve_compaction": [_vp],
    "pcl_group_step_fused": [_vp, c_double, c_int, c_double, c_double, c_int, c_double, c_double, c_char_p, c_int, c_uint64,
                             c_uint32, _vp, c_int, _vp],
    "pcl_group_step_fused_delete": [_vp, c_double, c_double, c_double, c_int, c_int, c_uint64, c_uint32, _vp, c_int, _vp],
    "pcl_group_step_fused_multi": [_vp, c_double, c_int, c_double, c_double, c_int, c_double, c_double, c_char_p, c_uint64,
                                   c_uint32, _vp, c_int, _vp],
    "pcl_group_step_fused_delete_multi": [_vp, c_double, c_int, c_double, c_double, c_uint64, c_uint32, _vp, c_int, _vp],
    "pcl_group_step_mixed_multi": [_vp, c_double, c_int, c_int, _vp, c_double, c_double, c_int, c_double, c_double, c_char_p,
                                   c_double, c_double, c_uint64, c_uint32, _vp, c_int, _vp],
    "pcl_group_download": [_vp, c_int, _vp, c_int64, c_int64],
    "pcl_group_download_ids": [_vp, _vp, c_int64, c_int64],
}
EXPORTS = sorted(list(_PROTOTYPES) + ["pcl_last_error"])

_lib = None


def load():
    """dlopen libphysicl_hip.so (built by ``python -m physicl_amd.build``).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "physicl_amd: %s is missing -- build it with `python -m physicl_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.pcl_last_error.argtypes = []
    lib.pcl_last_error.restype = c_char_p
    if lib.pcl_abi_version() != 1:
        raise ImportError("libphysicl_hip ABI version %d, expected 1" % lib.pcl_abi_version())
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = (_lib.pcl_last_error() or b"").decode("utf-8", "replace")
        raise (ExpressionError if rc in (-4, -5) else HipError)(rc, msg)


def set_knob(name, value=None):
    """An A/B switch of the library (pcl_set_knob): ``value`` = its text, None = back to the environment variable."""
    check(load().pcl_set_knob(name.encode(), None if value is None else str(value).encode()))


def device_count():
    n = c_int(0)
    rc = load().pcl_device_count(byref(n))
    return n.value if rc == 0 else 0


def pool_bytes():
    """Bytes of freed device blocks the library keeps for the next store of this process (PCL_POOL_GB bounds it)."""
    n = c_int64(0)
    check(load().pcl_pool_bytes(byref(n)))
    return n.value


def pool_info():
    """{'idle_blocks', 'idle_handles', 'parked_va', 'remaps_avoided', 'vmm_on'} (pcl_pool_info)."""
    a, b, c, d, e = c_int64(0), c_int64(0), c_int64(0), c_int64(0), c_int(0)
    check(load().pcl_pool_info(byref(a), byref(b), byref(c), byref(d), byref(e)))
    return {"idle_blocks": a.value, "idle_handles": b.value, "parked_va": c.value, "remaps_avoided": d.value, "vmm_on": bool(e.value)}


def pool_trim():
    """Hand every idle block back to the driver; returns the bytes released."""
    n = c_int64(0)
    check(load().pcl_pool_trim(byref(n)))
    return n.value


def validate_expr(expr):
    lib = load()
    rc = lib.pcl_expr_validate(expr.encode())
    if rc != 0:
        raise ExpressionError(rc, lib.pcl_last_error().decode())


def _host(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a, a.ctypes.data_as(c_void_p)


class DeviceArray:
    """Raw device allocation for Level-1 callers (what cl_array.to_device / cl_array.empty gave)."""

    def __init__(self, dev, n, dtype):
        self.dev, self.n, self.dtype = dev, int(n), np.dtype(dtype)
        p = c_void_p()
        check(dev.lib.pcl_dev_alloc(dev.ctx, self.n * self.dtype.itemsize, byref(p)))
        self.ptr = p

    @classmethod
    def from_host(cls, dev, arr, dtype=np.float64):
        arr, hp = _host(arr, dtype)
        self = cls(dev, arr.size, dtype)
        check(dev.lib.pcl_h2d(dev.ctx, self.ptr, hp, arr.nbytes))
        return self

    def fill_bytes(self, value):
        check(self.dev.lib.pcl_dev_memset(self.dev.ctx, self.ptr, value, self.n * self.dtype.itemsize))

    def get(self):
        out = np.empty(self.n, dtype=self.dtype)
        check(self.dev.lib.pcl_d2h(self.dev.ctx, out.ctypes.data_as(c_void_p), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr is not None and self.dev.ctx:
            check(self.dev.lib.pcl_dev_free(self.dev.ctx, self.ptr))
        self.ptr = None


_CTYPES = {"double": c_double, "int": c_int32, "float": ctypes.c_float, "long": c_int64}


class UserKernel:
    """A hipRTC-compiled user kernel (``pcl_user_kernel_*``): the device side of ``CLProgram``."""

    def __init__(self, dev, name, params, body):
        self.dev, self.name, self.params = dev, name, list(params)
        decl = ", ".join("%s %s%s" % (ct, "*" if ptr else "", nm) for ct, nm, ptr in self.params)
        h = c_void_p()
        rc = dev.lib.pcl_user_kernel_build(dev.ctx, name.encode(), decl.encode(), body.encode(), byref(h))
        if rc != 0:
            raise HipError(rc, dev.lib.pcl_last_error().decode("utf-8", "replace"))
        self.handle = h
        fields = [("a%d" % i, c_void_p if ptr else _CTYPES[ct]) for i, (ct, nm, ptr) in enumerate(self.params)]
        self._argtype = type("pcl_args_" + name, (ctypes.Structure,), {"_fields_": fields})

    def __call__(self, n, *args):
        assert len(args) == len(self.params), "kernel %s takes %d arguments" % (self.name, len(self.params))
        packed = self._argtype()
        for i, ((ct, nm, ptr), a) in enumerate(zip(self.params, args)):
            setattr(packed, "a%d" % i, a.ptr if ptr else _CTYPES[ct](a).value)
        check(self.dev.lib.pcl_user_kernel_launch(self.dev.ctx, self.handle, int(n), byref(packed), ctypes.sizeof(packed)))

    def free(self):
        if self.handle is not None and self.dev.ctx:
            self.dev.lib.pcl_user_kernel_free(self.dev.ctx, self.handle)
        self.handle = None


class Device:
    """One HIP context (device + stream) and, optionally, one resident particle store."""

    def __init__(self, device=0, stream=None):
        self.lib = load()
        ctx = c_void_p()
        check(self.lib.pcl_ctx_create(int(device), c_void_p(stream) if stream else None, byref(ctx)))
        self.ctx = ctx
        self.device = int(device)

    # ---------------------------------------------------------------- lifecycle
    def close(self):
        if getattr(self, "ctx", None):
            self.lib.pcl_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_rtc_background(self, on=True):
        """Built-in variable_n_fn shapes start on the ahead-of-time kernels while hipRTC compiles (pcl_ctx_set_rtc_background)."""
        check(self.lib.pcl_ctx_set_rtc_background(self.ctx, 1 if on else 0))

    def rtc_wait(self):
        """Block until every specialisation still compiling in the background is in place; returns how many were pending."""
        n = c_int(0)
        check(self.lib.pcl_ctx_rtc_wait(self.ctx, byref(n)))
        return n.value

    def sync(self):
        check(self.lib.pcl_ctx_sync(self.ctx))

    def info(self):
        name = ctypes.create_string_buffer(256)
        hbm, cu, wf = c_int64(), c_int(), c_int()
        check(self.lib.pcl_ctx_device_info(self.ctx, name, 256, byref(hbm), byref(cu), byref(wf)))
        pci = ctypes.create_string_buffer(64)
        check(self.lib.pcl_ctx_device_pci(self.ctx, pci, 64))
        return {"name": name.value.decode(), "hbm_bytes": hbm.value, "compute_units": cu.value,
                "wavefront": wf.value, "device": self.device, "pci_bus_id": pci.value.decode()}

    def mem_info(self):
        """(free, total) bytes of the device as the driver reports them (idle pool blocks count as used)."""
        f, t = c_int64(), c_int64()
        check(self.lib.pcl_ctx_mem_info(self.ctx, byref(f), byref(t)))
        return f.value, t.value

    def timer_start(self):
        check(self.lib.pcl_timer_start(self.ctx))

    def timer_stop(self):
        ms = c_double()
        check(self.lib.pcl_timer_stop(self.ctx, byref(ms)))
        return ms.value

    def prof_enable(self, on=True):
        check(self.lib.pcl_prof_enable(self.ctx, 1 if on else 0))

    def prof_read(self, kernel_id):
        n, tot, mn, mx = c_int64(), c_double(), c_double(), c_double()
        check(self.lib.pcl_prof_read(self.ctx, kernel_id, byref(n), byref(tot), byref(mn), byref(mx)))
        return {"launches": n.value, "total_ms": tot.value, "min_ms": mn.value, "max_ms": mx.value,
                "avg_ms": tot.value / n.value if n.value else 0.0}

    # ---------------------------------------------------------------- Level 1 (reference-ABI kernels)
    def array(self, host, dtype=np.float64):
        return DeviceArray.from_host(self, host, dtype)

    def empty(self, n, dtype=np.float64):
        return DeviceArray(self, n, dtype)

    def k_light_scatter_step_del(self, dx, dy, dz, rand, n, A, result, N):
        check(self.lib.pcl_k_light_scatter_step_del(self.ctx, dx.ptr, dy.ptr, dz.ptr, rand.ptr, n, A, result.ptr, N))

    def k_scatter_delete_test(self, d0, d1, d2, rand, A, n, res, N):
        check(self.lib.pcl_k_scatter_delete_test(self.ctx, d0.ptr, d1.ptr, d2.ptr, rand.ptr, A, n, res.ptr, N))

    def k_light_scatter_step_sphere(self, d0, d1, d2, rtheta, rphi, rand, A, n, E, r, res0, res1, res2, N, flags, c, h,
                                    n_expr=None):
        nul = c_void_p()
        r = r or (None, None, None)
        check(self.lib.pcl_k_light_scatter_step_sphere(
            self.ctx, d0.ptr, d1.ptr, d2.ptr, rtheta.ptr, rphi.ptr, rand.ptr, A, n, E.ptr if E is not None else nul,
            *[x.ptr if x is not None else nul for x in r], res0.ptr, res1.ptr, res2.ptr, N, flags, c, h,
            n_expr.encode() if n_expr is not None else None))

    def k_compact_indices(self, flags, N, idx_out):
        keep = c_int64()
        check(self.lib.pcl_k_compact_indices(self.ctx, flags.ptr, N, idx_out.ptr, byref(keep)))
        return keep.value

    def user_kernel(self, name, params, body):
        """Compile a kernel body (OpenCL-C dialect) with hipRTC.  params: [(ctype, name, is_pointer), ...]."""
        return UserKernel(self, name, params, body)

    # ---------------------------------------------------------------- Level 2 (resident store)
    def store_alloc(self, capacity, dtype="f64"):
        """dtype 'f64' (the reference's precision, default) or 'f32' (precision sweep)."""
        code = {"f64": DTYPE_F64, "f32": DTYPE_F32, np.float64: DTYPE_F64, np.float32: DTYPE_F32}[dtype]
        check(self.lib.pcl_store_alloc_dtype(self.ctx, int(capacity), code))

    @property
    def np_dtype(self):
        d = c_int()
        check(self.lib.pcl_store_dtype(self.ctx, byref(d)))
        return _NP_DTYPE[d.value]

    def store_free(self):
        check(self.lib.pcl_store_free(self.ctx))

    @property
    def capacity(self):
        v = c_int64()
        check(self.lib.pcl_store_capacity(self.ctx, byref(v)))
        return v.value

    @property
    def count(self):
        v = c_int64()
        check(self.lib.pcl_store_count(self.ctx, byref(v)))
        return v.value

    @property
    def slots(self):
        """Slots the store spans: the count when it is dense, more while the delete path keeps removed photons' slots
        behind its alive mask (pcl_store_slots)."""
        v = c_int64()
        check(self.lib.pcl_store_slots(self.ctx, byref(v), None))
        return v.value

    def last_multi_work(self):
        """(dense passes, wave-steps, photons per wave, wave-steps on exp's saturation shortcut or -1) of the last
        step_fused_multi launch (pcl_store_last_multi_work)."""
        a, b, c, e = c_int64(), c_int64(), c_int(), c_int64()
        check(self.lib.pcl_store_last_multi_work(self.ctx, byref(a), byref(b), byref(c), byref(e)))
        return a.value, b.value, c.value, e.value

    def last_multi_clock(self):
        """GHz the chip held under the last step_fused_multi launch (pcl_store_last_multi_clock); 0.0 before any."""
        g = c_double()
        check(self.lib.pcl_store_last_multi_clock(self.ctx, byref(g)))
        return g.value

    def last_mixed_rows(self):
        """Rows of 64 particles per wave and trip in the last step_mixed_multi launch (pcl_store_last_mixed_rows): 2 or 3."""
        g = c_int()
        check(self.lib.pcl_store_last_mixed_rows(self.ctx, byref(g)))
        return g.value

    def ahead_clock(self):
        """GHz the chip held under this context's k_delete_ahead_live launches (pcl_store_ahead_clock); 0.0 before any."""
        g = c_double()
        check(self.lib.pcl_store_ahead_clock(self.ctx, byref(g)))
        return g.value

    def last_multi_hist(self):
        """Hits queued per wave and step in the last step_fused_multi launch, 129 bins (debug builds: pcl_store_last_multi_hist)."""
        out = np.zeros(129, dtype=np.int64)
        check(self.lib.pcl_store_last_multi_hist(self.ctx, out.ctypes.data))
        return out

    def ahead_stats(self):
        """(launches, bodies answered, launches not used up) of the delete bodies worked out ahead (pcl_store_ahead_stats)."""
        a, b, c = c_int64(), c_int64(), c_int64()
        check(self.lib.pcl_store_ahead_stats(self.ctx, byref(a), byref(b), byref(c)))
        return a.value, b.value, c.value

    def ahead_work(self):
        """(groups of 128 slots loaded with a first pass of two bodies, ... of one body, rounds deciding two bodies, rounds
        deciding one) summed over the k_delete_ahead_live launches of this context, as the kernel tallied them
        (pcl_store_ahead_work)."""
        a, b, c, d = c_int64(), c_int64(), c_int64(), c_int64()
        check(self.lib.pcl_store_ahead_work(self.ctx, byref(a), byref(b), byref(c), byref(d)))
        return a.value, b.value, c.value, d.value

    def reserve_compaction(self):
        """Allocate the second slab, id arrays and mask scratch of the delete path now (pcl_store_reserve_compaction)."""
        check(self.lib.pcl_store_reserve_compaction(self.ctx))

    def set_count(self, count, id_base=0):
        check(self.lib.pcl_store_set_count(self.ctx, int(count), int(id_base)))

    def upload(self, field, host, offset=0):
        a, hp = _host(host, self.np_dtype)                 # rounded to the store's precision on the host
        check(self.lib.pcl_store_upload(self.ctx, field, hp, offset, a.size))

    def download(self, field, n=None, offset=0):
        """Field values in the store's own dtype (float64 or float32 array)."""
        n = self.count - offset if n is None else n
        out = np.empty(n, dtype=self.np_dtype)
        check(self.lib.pcl_store_download(self.ctx, field, out.ctypes.data_as(c_void_p), offset, n))
        return out

    def upload_ids(self, host, offset=0):
        a, hp = _host(host, np.int64)
        check(self.lib.pcl_store_upload_ids(self.ctx, hp, offset, a.size))

    def download_ids(self, n=None, offset=0):
        n = self.count - offset if n is None else n
        out = np.empty(n, dtype=np.int64)
        check(self.lib.pcl_store_download_ids(self.ctx, out.ctypes.data_as(c_void_p), offset, n))
        return out

    def upload_kind(self, host, offset=0):
        a, hp = _host(host, np.uint8)
        check(self.lib.pcl_store_upload_kind(self.ctx, hp, offset, a.size))

    def download_kind(self, n=None, offset=0):
        n = self.count - offset if n is None else n
        out = np.empty(n, dtype=np.uint8)
        check(self.lib.pcl_store_download_kind(self.ctx, out.ctypes.data_as(c_void_p), offset, n))
        return out

    def field_ptr(self, field):
        p = c_void_p()
        check(self.lib.pcl_store_field_ptr(self.ctx, field, byref(p)))
        return p.value

    def alloc_info(self):
        """How the store's slab was chosen: {'candidates_GBps': [...], 'chosen_GBps': x} (empty list: no selection)."""
        n, chosen = c_int(0), c_double(0.0)
        rates = (c_double * 8)()
        check(self.lib.pcl_store_alloc_info(self.ctx, byref(n), rates, 8, byref(chosen)))
        return {"candidates_GBps": [round(rates[k], 1) for k in range(n.value)], "chosen_GBps": round(chosen.value, 1)}

    def layout(self):
        """(tile_len, tile_stride) in elements: element i of a row is at row0[(i // T) * stride + i % T]."""
        t, ts = c_int64(), c_int64()
        check(self.lib.pcl_store_layout(self.ctx, byref(t), byref(ts)))
        return t.value, ts.value

    def upload_rand(self, which, host):
        a, hp = _host(host, self.np_dtype)
        check(self.lib.pcl_store_upload_rand(self.ctx, which, hp, a.size))

    RAND3_CHUNK = 1 << 20

    def upload_rand3(self, u3, offset=0):
        """Rows [offset, offset + len(u3)) of the reference's per-photon draws (rtheta-, rphi-, rand-uniform): ``u3`` is
        what ``np.random.random((n, 3))`` returns, at most RAND3_CHUNK rows per call, chunks in particle order from 0."""
        a = np.ascontiguousarray(u3, dtype=np.float64).reshape(-1, 3)
        check(self.lib.pcl_store_upload_rand3(self.ctx, a.ctypes.data, int(offset), len(a)))

    def fill_photons(self, n, id_base, c, e_min, e_max, seed):
        check(self.lib.pcl_store_fill_photons(self.ctx, int(n), int(id_base), c, e_min, e_max, int(seed)))

    def fill_photons_table(self, n, id_base, c, cdf, grid, seed):
        cdf, cp = _host(cdf, np.float64)
        grid, gp = _host(grid, np.float64)
        assert cdf.shape == grid.shape
        check(self.lib.pcl_store_fill_photons_table(self.ctx, int(n), int(id_base), c, cp, gp, len(cdf), int(seed)))

    def upload_state(self, state):
        """state: dict with 'r','v','dr','dv' -> (n,3) or 3 arrays, 'E' -> (n,).  Sets count = n."""
        n = len(np.asarray(state["E"]))
        self.set_count(n, int(state.get("id_base", 0)))
        for g, fids in FIELD_GROUPS.items():
            a = state.get(g)
            for k, fid in enumerate(fids):
                col = np.zeros(n) if a is None else (np.asarray(a)[:, k] if np.ndim(a) == 2 else a[k])
                self.upload(fid, col)
        self.upload(E, state["E"])
        if state.get("id") is not None:
            self.upload_ids(state["id"])
        if state.get("kind") is not None:
            self.upload_kind(state["kind"])

    def download_state(self):
        n = self.count
        out = {g: [self.download(fid, n) for fid in fids] for g, fids in FIELD_GROUPS.items()}
        out["E"] = self.download(E, n)
        out["id"] = self.download_ids(n)
        return out

    def step_newton(self, dt):
        check(self.lib.pcl_step_newton(self.ctx, float(dt)))

    def step_scatter_isotropic(self, A, n, flags, c, h, n_expr=None, rng_mode=RNG_PHILOX, seed=0, step=0,
                               want_hits=True):
        hits = c_int64()
        check(self.lib.pcl_step_scatter_isotropic(
            self.ctx, float(A), float(n), int(flags), float(c), float(h),
            n_expr.encode() if n_expr is not None else None, int(rng_mode), int(seed), int(step) & 0xFFFFFFFF,
            byref(hits) if want_hits else None))
        return hits.value if want_hits else None

    def scatter_pcoll(self, A, n, flags, c, h):
        """Collision probability of every particle (constant n; ``flags`` & SCATTER_WAVELENGTH adds the wavelength
        term), as the scatter kernels compute it: input of the host-side RNG replay of the reference's CPU paths."""
        out = np.empty(self.count, dtype=self.np_dtype)
        check(self.lib.pcl_step_scatter_pcoll(self.ctx, float(A), float(n), int(flags), float(c), float(h),
                                              out.ctypes.data_as(c_void_p)))
        return out

    def step_delete_flags(self, flags):
        """Remove the particles whose flag is 1 (stable compaction of the whole state).  Returns (alive, removed)."""
        f = np.ascontiguousarray(flags, dtype=np.int32)
        if f.size != self.count:
            raise ValueError("one flag per particle: got %d for %d" % (f.size, self.count))
        alive, removed = c_int64(), c_int64()
        check(self.lib.pcl_step_delete_flags(self.ctx, f.ctypes.data_as(c_void_p), byref(alive), byref(removed)))
        return alive.value, removed.value

    def step_fused(self, dt, scatter=None, planes=None, sync=True, lazy=False):
        """One pass: Newton, then ScatterIsotropic if ``scatter`` (dict: A, n, flags, c, h, n_expr, rng_mode,
        seed, step), then counters if ``planes`` is not None (sequence of [x,y,z] rows, may be empty).
        Returns {'N','sign','planes','hits'} when counters are on and sync, else None."""
        sc = scatter or {}
        if planes is None:
            npl, pp, out, op = -1, None, None, None
        else:
            pl = np.ascontiguousarray(np.asarray(planes, dtype=np.float64).reshape(-1, 3))
            npl = len(pl)
            pp = pl.ctypes.data_as(c_void_p) if npl else None
            out = np.zeros(5 + npl, dtype=np.int64) if sync else None
            op = out.ctypes.data_as(c_void_p) if sync else None
        expr = sc.get("n_expr")
        check(self.lib.pcl_step_fused(
            self.ctx, float(dt), 1 if scatter else 0, float(sc.get("A", 0.0)), float(sc.get("n", 0.0)),
            int(sc.get("flags", 0)) | (FUSED_LAZY if lazy else 0), float(sc.get("c", 0.0)), float(sc.get("h", 0.0)),
            expr.encode() if expr is not None else None, int(sc.get("rng_mode", RNG_PHILOX)), int(sc.get("seed", 0)),
            int(sc.get("step", 0)) & 0xFFFFFFFF, pp, npl, op))
        if out is None:
            return None
        return {"N": int(out[0]), "sign": out[1:4].copy(), "planes": out[4:4 + npl].copy(), "hits": int(out[4 + npl])}

    def is_uniform(self):
        """All photons with implicit ids: eligible for step_fused_multi."""
        u = c_int()
        check(self.lib.pcl_store_is_uniform(self.ctx, byref(u)))
        return bool(u.value)

    def step_fused_multi(self, dt, k_steps, scatter, planes=(), sync=True, raw=False):
        """``k_steps`` consecutive lazy fused steps (Newton + ScatterIsotropic + sign / plane counters, device RNG,
        launch indices scatter['step'] .. +k_steps-1) in one pass over the store.  Returns a list of k_steps dicts like
        step_fused's (or None if not sync)."""
        sc = scatter
        pl = np.ascontiguousarray(np.asarray(planes, dtype=np.float64).reshape(-1, 3))
        npl = len(pl)
        out = np.zeros((k_steps, 5 + npl), dtype=np.int64) if sync else None
        expr = sc.get("n_expr")
        check(self.lib.pcl_step_fused_multi(
            self.ctx, float(dt), int(k_steps), float(sc["A"]), float(sc["n"]), int(sc.get("flags", 0)),
            float(sc.get("c", 0.0)), float(sc.get("h", 0.0)), expr.encode() if expr is not None else None,
            int(sc.get("seed", 0)), int(sc.get("step", 0)) & 0xFFFFFFFF, pl.ctypes.data_as(c_void_p) if npl else None, npl,
            out.ctypes.data_as(c_void_p) if sync else None))
        if out is None:
            return None
        if raw:                   # (k_steps, 5 + n_planes) int64: [N, sign x 3, planes ..., hits] per step
            return out
        return [{"N": int(o[0]), "sign": o[1:4].copy(), "planes": o[4:4 + npl].copy(), "hits": int(o[4 + npl])} for o in out]

    def step_mixed_multi(self, dt, k_passes, phases, scatter=None, delete=None, planes=(), seed=0, step=0, raw=False):
        """``k_passes`` passes of a loop whose body holds the phases ``phases`` -- a sequence of "iso" / "delete", at
        most one of each -- every phase being Newton + the light step + the counters of the measure steps behind it;
        one pass over the store and (with a delete phase) one compaction.  ``scatter``: dict A, n, flags, c, h, n_expr
        (kernel constants) of the isotropic phase; ``delete``: (A, n) of the delete phase.  Device RNG; phase j of
        pass p uses launch index ``step + p * len(phases) + j``.  Returns one dict per phase, in order:
        {'phase', 'N' (alive after it), 'sign', 'planes', 'hits' | 'removed'}."""
        kinds = np.array([{"iso": PHASE_ISOTROPIC, "delete": PHASE_DELETE}[p] for p in phases], dtype=np.int32)
        sc = scatter or {}
        A_d, n_d = delete if delete is not None else (0.0, 0.0)
        pl = np.ascontiguousarray(np.asarray(planes, dtype=np.float64).reshape(-1, 3))
        npl = len(pl)
        out = np.zeros((k_passes * len(kinds), 5 + npl), dtype=np.int64)
        expr = sc.get("n_expr")
        check(self.lib.pcl_step_mixed_multi(
            self.ctx, float(dt), int(k_passes), len(kinds), kinds.ctypes.data_as(c_void_p), float(sc.get("A", 0.0)),
            float(sc.get("n", 0.0)), int(sc.get("flags", 0)), float(sc.get("c", 0.0)), float(sc.get("h", 0.0)),
            expr.encode() if expr is not None else None, float(A_d), float(n_d), int(seed), int(step) & 0xFFFFFFFF,
            pl.ctypes.data_as(c_void_p) if npl else None, npl, out.ctypes.data_as(c_void_p)))
        if raw:                   # (k_passes * len(phases), 5 + n_planes) int64: [N, sign x 3, planes ..., hits | removed]
            return out
        rows = []
        for k, o in enumerate(out):
            ph = phases[k % len(kinds)]
            rows.append({"phase": ph, "N": int(o[0]), "sign": o[1:4].copy(), "planes": o[4:4 + npl].copy(),
                         ("hits" if ph == "iso" else "removed"): int(o[4 + npl])})
        return rows

    def trace_ahead(self, ids, dt, k_passes, phases, record_phase=0, scatter=None, delete=None, seed=0, step=0, defer=False):
        """Where the particles with the (ascending) ids ``ids`` will be when a trace step behind phase ``record_phase`` runs
        in each of the next ``k_passes`` passes of a loop with the phases ``phases`` ("iso" / "delete"; arguments as
        step_mixed_multi's) -- worked out from the store as it stands, which is NOT changed (pcl_store_trace_ahead): call it
        with the arguments of the K-pass launch, before that launch.  Returns (k_passes, len(ids), 4) float64:
        r0, r1, r2, moved (dv != 0); NaN rows where the particle is not in the store at that point.
        ``defer=True``: the kernel is only enqueued and a function is returned that hands the rows out -- call it behind the
        K-pass launch (same stream, in order: the rows are there when the launch's own counter rows are, no extra wait)."""
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        kinds = np.array([{"iso": PHASE_ISOTROPIC, "delete": PHASE_DELETE}[p] for p in phases], dtype=np.int32)
        sc = scatter or {}
        A_d, n_d = delete if delete is not None else (0.0, 0.0)
        out = np.empty((int(k_passes), len(ids), 4), dtype=np.float64)
        expr = sc.get("n_expr")
        check(self.lib.pcl_store_trace_ahead(
            self.ctx, ids.ctypes.data_as(c_void_p), len(ids), float(dt), int(k_passes), len(kinds), kinds.ctypes.data_as(c_void_p),
            int(record_phase), float(sc.get("A", 0.0)), float(sc.get("n", 0.0)), int(sc.get("flags", 0)), float(sc.get("c", 0.0)),
            float(sc.get("h", 0.0)), expr.encode() if expr is not None else None, float(A_d), float(n_d), int(seed),
            int(step) & 0xFFFFFFFF, None if (defer and out.size) else out.ctypes.data_as(c_void_p)))
        if not defer:
            return out

        def read():
            if out.size:
                check(self.lib.pcl_store_trace_read(self.ctx, out.ctypes.data_as(c_void_p), out.size))
            return out
        return read

    def step_fused_read(self, n_planes=0):
        """Counters of the last ``step_fused(..., sync=False)``: same dict as the synchronous call."""
        out = np.zeros(5 + n_planes, dtype=np.int64)
        check(self.lib.pcl_step_fused_read(self.ctx, n_planes, out.ctypes.data_as(c_void_p)))
        return {"N": int(out[0]), "sign": out[1:4].copy(), "planes": out[4:4 + n_planes].copy(),
                "hits": int(out[4 + n_planes])}

    def last_scatter_hits(self):
        h = c_int64()
        check(self.lib.pcl_store_last_scatter_hits(self.ctx, byref(h)))
        return h.value

    def step_scatter_delete(self, A, n, rng_mode=RNG_PHILOX, seed=0, step=0):
        alive, removed = c_int64(), c_int64()
        check(self.lib.pcl_step_scatter_delete(self.ctx, float(A), float(n), int(rng_mode), int(seed),
                                               int(step) & 0xFFFFFFFF, byref(alive), byref(removed)))
        return alive.value, removed.value

    def step_fused_delete(self, dt, A, n, rng_mode=RNG_PHILOX, seed=0, step=0, planes=None, lazy=False):
        """Newton + ScatterDelete (+ counters on the survivors if ``planes`` is not None) as one pipeline.
        Returns {'N' (alive), 'removed', 'sign', 'planes'}."""
        if planes is None:
            npl, pp = -1, None
        else:
            # (a loop calls this once per body with the same planes: a ready float64 array is taken as it is)
            pl = planes if (type(planes) is np.ndarray and planes.dtype == np.float64 and planes.ndim == 2 and
                            planes.flags.c_contiguous) else np.ascontiguousarray(np.asarray(planes, dtype=np.float64).reshape(-1, 3))
            npl = len(pl)
            pp = pl.ctypes.data if npl else None
        k = max(npl, 0)
        out = np.zeros(5 + k, dtype=np.int64)
        rc = self.lib.pcl_step_fused_delete(self.ctx, float(dt), float(A), float(n), FUSED_LAZY if lazy else 0,
                                            int(rng_mode), int(seed), int(step) & 0xFFFFFFFF, pp, npl, out.ctypes.data)
        if rc != 0:
            check(rc)
        # (``out`` is this call's own array: the slices need no copy)
        return {"N": int(out[0]), "sign": out[1:4], "planes": out[4:4 + k], "removed": int(out[4 + k])}

    def step_fused_delete_multi(self, dt, k_steps, A, n, seed=0, step=0, planes=None, raw=False):
        """``k_steps`` delete loop bodies (Newton + ScatterDelete + counters on the survivors) in one pass and one
        compaction.  Returns a list of k_steps dicts {'N','removed','sign','planes'}."""
        if planes is None:
            npl, pp = -1, None
        else:
            pl = np.ascontiguousarray(np.asarray(planes, dtype=np.float64).reshape(-1, 3))
            npl = len(pl)
            pp = pl.ctypes.data_as(c_void_p) if npl else None
        k = max(npl, 0)
        out = np.zeros((k_steps, 5 + k), dtype=np.int64)
        check(self.lib.pcl_step_fused_delete_multi(self.ctx, float(dt), int(k_steps), float(A), float(n), int(seed),
                                                   int(step) & 0xFFFFFFFF, pp, npl, out.ctypes.data_as(c_void_p)))
        if raw:                   # (k_steps, 5 + n_planes) int64: [N, sign x 3, planes ..., removed] per body
            return out
        return [{"N": int(o[0]), "sign": o[1:4].copy(), "planes": o[4:4 + k].copy(), "removed": int(o[4 + k])} for o in out]

    def last_delete_flags(self, n):
        out = np.empty(n, dtype=np.int32)
        check(self.lib.pcl_store_last_delete_flags(self.ctx, out.ctypes.data_as(c_void_p), n))
        return out

    def plane_energies(self, plane, n_hint=None):
        """Energies of the photons that crossed ``plane`` ([x, y, z] with NaN in the free coordinates) in the last
        move, in particle order (ScatterMeasureStep(measure_E=True)).  ``n_hint``: the crossing count if the caller
        already has it from step_counters (sizes the host buffer)."""
        pl = np.ascontiguousarray(np.asarray(plane, dtype=np.float64).reshape(3))
        n = c_int64()
        cap = self.count if n_hint is None else min(int(n_hint), self.count)
        out = np.empty(max(cap, 1), dtype=self.np_dtype)
        check(self.lib.pcl_step_plane_energies(self.ctx, pl.ctypes.data_as(c_void_p), out.ctypes.data_as(c_void_p), cap, byref(n)))
        if n.value > cap:
            raise HipError(-2, "plane_energies: %d photons crossed, n_hint was %d" % (n.value, cap))
        return out[:n.value].copy()

    def step_counters(self, planes=()):
        planes = np.ascontiguousarray(np.asarray(planes, dtype=np.float64).reshape(-1, 3))
        out = np.zeros(CNT_PLANE0 + len(planes), dtype=np.int64)
        check(self.lib.pcl_step_counters(self.ctx, planes.ctypes.data_as(c_void_p) if len(planes) else None,
                                         len(planes), out.ctypes.data_as(c_void_p)))
        return out


class DeviceGroup:
    """``pcl_group_*``: several contexts in one process, sharded by global index, behind the C ABI (the shim owns the
    contexts and one worker thread per context, fans every step out and sums the counter rows).  The Python host layer
    uses physicl_amd.multidev.MultiDevice, which does the same over ``Device`` objects; this class binds the C-level
    form a non-Python host would use, for tests and scripts."""

    def __init__(self, devices):
        self.lib = load()
        ids = (c_int * len(devices))(*[int(d) for d in devices])
        g = c_void_p()
        check(self.lib.pcl_group_create(len(devices), ids, byref(g)))
        self.g, self.n = g, len(devices)

    def close(self):
        if getattr(self, "g", None):
            self.lib.pcl_group_destroy(self.g)
            self.g = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def shard(self, n_global, i):
        lo, hi = c_int64(), c_int64()
        check(self.lib.pcl_group_shard(self.g, int(n_global), int(i), byref(lo), byref(hi)))
        return lo.value, hi.value

    def store_alloc(self, capacity, dtype="f64"):
        check(self.lib.pcl_group_store_alloc(self.g, int(capacity), DTYPE_F64 if dtype == "f64" else DTYPE_F32))

    def fill_photons(self, n, id_base, c, e_min, e_max, seed):
        check(self.lib.pcl_group_fill_photons(self.g, int(n), int(id_base), c, e_min, e_max, int(seed)))

    @property
    def count(self):
        v = c_int64()
        check(self.lib.pcl_group_count(self.g, byref(v)))
        return v.value

    def sync(self):
        check(self.lib.pcl_group_sync(self.g))

    def step_fused_multi(self, dt, k_steps, sc, planes=()):
        pl = np.ascontiguousarray(np.asarray(planes, dtype=np.float64).reshape(-1, 3))
        out = np.zeros((k_steps, 5 + len(pl)), dtype=np.int64)
        expr = sc.get("n_expr")
        check(self.lib.pcl_group_step_fused_multi(self.g, float(dt), int(k_steps), float(sc["A"]), float(sc["n"]), int(sc.get("flags", 0)),
                                                  float(sc.get("c", 0.0)), float(sc.get("h", 0.0)), expr.encode() if expr else None,
                                                  int(sc.get("seed", 0)), int(sc.get("step", 0)) & 0xFFFFFFFF,
                                                  pl.ctypes.data if len(pl) else None, len(pl), out.ctypes.data))
        return out

    def step_fused_delete(self, dt, A, n, seed, step, planes=(), lazy=True):
        pl = np.ascontiguousarray(np.asarray(planes, dtype=np.float64).reshape(-1, 3))
        out = np.zeros(5 + len(pl), dtype=np.int64)
        check(self.lib.pcl_group_step_fused_delete(self.g, float(dt), float(A), float(n), FUSED_LAZY if lazy else 0, RNG_PHILOX, int(seed),
                                                   int(step) & 0xFFFFFFFF, pl.ctypes.data if len(pl) else None, len(pl), out.ctypes.data))
        return out

    def step_fused_delete_multi(self, dt, k_steps, A, n, seed, step, planes=()):
        pl = np.ascontiguousarray(np.asarray(planes, dtype=np.float64).reshape(-1, 3))
        out = np.zeros((k_steps, 5 + len(pl)), dtype=np.int64)
        check(self.lib.pcl_group_step_fused_delete_multi(self.g, float(dt), int(k_steps), float(A), float(n), int(seed), int(step) & 0xFFFFFFFF,
                                                         pl.ctypes.data if len(pl) else None, len(pl), out.ctypes.data))
        return out

    def step_mixed_multi(self, dt, k_passes, phases, sc, delete, seed, step, planes=()):
        kinds = np.array([{"iso": PHASE_ISOTROPIC, "delete": PHASE_DELETE}[p] for p in phases], dtype=np.int32)
        pl = np.ascontiguousarray(np.asarray(planes, dtype=np.float64).reshape(-1, 3))
        out = np.zeros((k_passes * len(kinds), 5 + len(pl)), dtype=np.int64)
        expr = sc.get("n_expr")
        check(self.lib.pcl_group_step_mixed_multi(self.g, float(dt), int(k_passes), len(kinds), kinds.ctypes.data, float(sc.get("A", 0.0)),
                                                  float(sc.get("n", 0.0)), int(sc.get("flags", 0)), float(sc.get("c", 0.0)), float(sc.get("h", 0.0)),
                                                  expr.encode() if expr else None, float(delete[0]), float(delete[1]), int(seed),
                                                  int(step) & 0xFFFFFFFF, pl.ctypes.data if len(pl) else None, len(pl), out.ctypes.data))
        return out

    def download(self, field, n=None, offset=0, dtype=None):
        n = self.count - offset if n is None else n
        if dtype is None:                         # the store's own element type (pcl_group_store_dtype)
            d = c_int()
            check(self.lib.pcl_group_store_dtype(self.g, byref(d)))
            dtype = np.float32 if d.value == DTYPE_F32 else np.float64
        out = np.empty(n, dtype=dtype)
        check(self.lib.pcl_group_download(self.g, int(field), out.ctypes.data, int(offset), int(n)))
        return out

    def download_ids(self, n=None, offset=0):
        n = self.count - offset if n is None else n
        out = np.empty(n, dtype=np.int64)
        check(self.lib.pcl_group_download_ids(self.g, out.ctypes.data, int(offset), int(n)))
        return out
