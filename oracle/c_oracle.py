"""ctypes wrapper of oracle/_build/liboracle.so (oracle/c/physicl_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/ and by the cpu_baseline leg of bench.py, never by physicl_amd."""
import ctypes
import os
import subprocess
from ctypes import c_double, c_int, c_int64, c_uint32, c_uint64, c_void_p

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "liboracle.so")
_lib = None


def load(build=True):
    global _lib
    if _lib is None:
        src = os.path.join(HERE, "c", "physicl_oracle.c")
        if build and (not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src)):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _lib = ctypes.CDLL(LIB)
        _lib.orc_scatter_isotropic.restype = c_int64
        _lib.orc_compact_indices.restype = c_int64
        _lib.orc_threads.restype = c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(c_void_p) if a is not None else None


def usable_cores():
    """CPUs this process may actually use: min(affinity mask, cgroup v2 cpu.max quota)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def threads():
    return load().orc_threads()


def set_threads(n):
    load().orc_set_threads(int(n))


def newton(st, dt):
    """In place on st = {'r':[3], 'v':[3], 'dr':[3]} of contiguous float64 arrays."""
    n = len(st["r"][0])
    load().orc_newton(*[_p(a) for a in st["r"]], *[_p(a) for a in st["v"]], *[_p(a) for a in st["dr"]],
                      c_double(dt), c_int64(n))


def philox_words(ids, step, block, seed):
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    out = np.empty((len(ids), 4), dtype=np.uint32)
    load().orc_philox_words(_p(ids), c_int64(len(ids)), c_uint32(step), c_uint32(block), c_uint64(seed), _p(out))
    return out


def delete_flags(d, rand, A, n):
    N = len(rand)
    flags = np.empty(N, dtype=np.int32)
    load().orc_delete_flags(_p(d[0]), _p(d[1]), _p(d[2]), _p(rand), c_double(A), c_double(n), _p(flags), c_int64(N))
    return flags


def delete_chain(v, dt, A, n, seed, step0, K, ids=None, id_base=0):
    """The body (0 .. K-1) that removes each photon of a delete run, K if none of the K bodies does (orc_delete_chain)."""
    N = len(v[0])
    death = np.empty(N, dtype=np.int32)
    if ids is not None:
        ids = np.ascontiguousarray(ids, dtype=np.int64)
    load().orc_delete_chain(_p(v[0]), _p(v[1]), _p(v[2]), _p(ids), c_int64(id_base), c_int64(N), c_double(dt), c_double(A), c_double(n),
                            c_uint64(seed), c_uint32(step0), c_int(K), _p(death))
    return death


def order_checksum(ids):
    """Order-sensitive checksum of a survivor list: sum of id * (position + 1) mod 2**64."""
    ids = np.asarray(ids).astype(np.uint64)
    with np.errstate(over="ignore"):
        return int((ids * (np.arange(len(ids), dtype=np.uint64) + np.uint64(1))).sum(dtype=np.uint64))


def compact_indices(flags):
    flags = np.ascontiguousarray(flags, dtype=np.int32)
    idx = np.empty(len(flags), dtype=np.int64)
    k = load().orc_compact_indices(_p(flags), c_int64(len(flags)), _p(idx))
    return idx[:k]


def scatter_isotropic(st, A, n, c, h, use_E, profile=0, prof_k=0.0, prof_off=0.0, seed=0, step=0, ids=None,
                      id_base=0, draws=None):
    """In place on st['v'], st['dv'].  draws = (rtheta, rphi, rand) to use input randoms."""
    N = len(st["E"])
    rt, rp, ra = draws if draws is not None else (None, None, None)
    return load().orc_scatter_isotropic(
        *[_p(a) for a in st["dr"]], _p(st["E"]), _p(st["r"][0]), *[_p(a) for a in st["v"]], *[_p(a) for a in st["dv"]],
        _p(ids), c_int64(id_base), c_int64(N), c_double(A), c_double(n), c_double(c), c_double(h), c_int(use_E),
        c_int(profile), c_double(prof_k), c_double(prof_off), c_uint64(seed), c_uint32(step), _p(rt), _p(rp), _p(ra))


def counters(st, plane_axis=None, L=0.0):
    out = np.zeros(4, dtype=np.int64)
    x = st["r"][plane_axis] if plane_axis is not None else None
    dx = st["dr"][plane_axis] if plane_axis is not None else None
    load().orc_counters(*[_p(a) for a in st["v"]], _p(x), _p(dx), c_double(L), c_int64(len(st["v"][0])), _p(out))
    return out
