#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Run in the build container only (``/root/reference`` does not travel to the GPU box):

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

What is real reference code and what is a stand-in
--------------------------------------------------
* REAL: every line of ``/root/reference/physicl/{__init__,light,newton}.py`` that runs here is the
  reference's own code, imported from where it lies.  That covers the CPU paths
  (``NewtonianKinematicsStep.run`` newton.py:10-16, all measure steps light.py:361-431,
  ``Measurement`` __init__.py:18-291) and the complete HOST side of the OpenCL paths
  (``CLProgram.build_kernel``/``run`` __init__.py:583-664, ``ScatterIsotropicStep.__run_cl``
  light.py:281-331, ``ScatterDeleteStep.run`` light.py:231-260,
  ``ScatterDeleteStepReference.__run_cl`` light.py:164-205): gather order, RNG consumption order,
  the A/n swap, argument order, write-back of v/dv, list removal.
* STAND-IN: ``pyopencl`` is not installed and there is no OpenCL device.  ``import physicl`` needs
  the module names, so two empty modules ``pyopencl`` / ``pyopencl.array`` are registered with the
  five entry points the reference calls (``create_some_context``, ``CommandQueue``, ``Program``,
  ``array.to_device``, ``array.empty``), backed by numpy arrays.
* KERNEL MATHS: the OpenCL C kernel *text the reference generates at run time* is handed to
  ``Program(ctx, src).build()``.  The stand-in compiles that text, unmodified, as C99 with gcc
  (a 6-line prelude maps ``__kernel``/``__global``/``get_global_id`` to plain C) and executes it
  once per work-item.  So the arithmetic of the fixtures is the reference's own kernel source,
  with glibc libm in place of an OpenCL device maths library (gcc -O2 folds ``pow(x, 2)`` to
  ``x*x`` exactly as OpenCL compilers do; verified).  No kernel text is stored in the fixtures:
  only its sha256 and the parsed argument-name order.
* ``np.int`` (removed in numpy >= 1.24) is used by the reference for the delete flags
  (``dtype=np.int`` __init__.py:653, light.py:198).  The published results were produced on
  Windows, where ``np.int`` is a 32-bit C long and therefore matches the kernel's ``int`` output;
  the generator restores that meaning (``np.int = np.int32``).

The fixtures are DATA (inputs, outputs, seeds).  Nothing under tests/golden/ contains reference
source text.
"""
import ctypes
import hashlib
import os
import re
import subprocess
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

# ----------------------------------------------------------------------------------------------
# stand-in pyopencl
# ----------------------------------------------------------------------------------------------
_PRELUDE = r"""
#include <math.h>
#define __kernel
#define __global
static long pcl_gid__;
static inline long get_global_id(int d) { (void)d; return pcl_gid__; }
"""

KERNEL_LOG = []  # one dict per launch: name, arg names, inputs (copies), outputs (after launch)


class _Holder:
    """numpy 'device array': what cl_array.to_device / cl_array.empty return."""

    def __init__(self, arr):
        self.data = arr
        self.shape = arr.shape

    def get(self):
        return self.data.copy()


class _Kernel:
    def __init__(self, lib, name, params):
        self.lib, self.name, self.params = lib, name, params

    def __call__(self, queue, gshape, lshape, *args):
        assert lshape is None, "reference launches with local size None"
        n = int(gshape[0])
        assert len(args) == len(self.params), (len(args), self.params)
        cargs, rec = [], {"name": self.name, "argnames": [p[1] for p in self.params], "N": n, "args": {}}
        for (ctype, pname, is_ptr), a in zip(self.params, args):
            if is_ptr:
                assert isinstance(a, np.ndarray) and a.flags["C_CONTIGUOUS"]
                if ctype == "double":
                    assert a.dtype == np.float64
                elif ctype == "int":
                    assert a.dtype == np.int32, a.dtype
                cargs.append(a.ctypes.data_as(ctypes.c_void_p))
                rec["args"][pname] = a  # reference to the live buffer; snapshot after launch
            else:
                assert ctype == "double"
                cargs.append(ctypes.c_double(float(a)))
                rec["args"][pname] = np.float64(a)
        fn = getattr(self.lib, "drive_" + self.name)
        fn.restype = None
        fn(ctypes.c_long(n), *cargs)
        rec["args"] = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in rec["args"].items()}
        KERNEL_LOG.append(rec)


class _Built:
    pass


class _Program:
    sources = {}  # kernel name -> sha256 of the text the reference generated

    def __init__(self, ctx, src):
        self.src = src

    def build(self):
        m = re.search(r"__kernel\s+void\s+(\w+)\s*\((.*?)\)\s*\{", self.src, re.S)
        name, plist = m.group(1), m.group(2)
        params = []
        for p in plist.split(","):
            p = p.replace("__global", "").strip()
            is_ptr = "*" in p
            ctype, pname = p.replace("*", " ").split()
            params.append((ctype, pname, is_ptr))
        sig = ", ".join(("%s *%s" if ip else "%s %s") % (ct, pn) for ct, pn, ip in params)
        call = ", ".join(pn for _, pn, _ in params)
        drv = "\nvoid drive_%s(long N, %s) { for (pcl_gid__ = 0; pcl_gid__ < N; ++pcl_gid__) %s(%s); }\n" % (
            name, sig, name, call)
        d = tempfile.mkdtemp(prefix="pcl_golden_")
        cfile, sofile = os.path.join(d, name + ".c"), os.path.join(d, name + ".so")
        with open(cfile, "w") as f:
            f.write(_PRELUDE + self.src + drv)
        subprocess.check_call(["gcc", "-std=gnu99", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC",
                               "-shared", cfile, "-o", sofile, "-lm"])
        lib = ctypes.CDLL(sofile)
        _Program.sources[name + ":" + ",".join(pn for _, pn, _ in params)] = hashlib.sha256(
            self.src.encode()).hexdigest()
        b = _Built()
        setattr(b, name, _Kernel(lib, name, params))
        return b


def install_standins():
    cl = types.ModuleType("pyopencl")
    cla = types.ModuleType("pyopencl.array")
    cl.array = cla
    cl.create_some_context = lambda *a, **k: object()
    cl.CommandQueue = lambda ctx, *a, **k: object()
    cl.Program = _Program
    cla.to_device = lambda q, arr: _Holder(np.ascontiguousarray(arr).copy())
    cla.empty = lambda q, shape, dtype=np.float64: _Holder(np.full(shape, -7, dtype=dtype))
    sys.modules["pyopencl"] = cl
    sys.modules["pyopencl.array"] = cla
    if not hasattr(np, "int"):
        np.int = np.int32  # Windows meaning of np.int; see module docstring
    sys.path.insert(0, REF)


install_standins()
import physicl  # noqa: E402
import physicl.light as light  # noqa: E402
import physicl.newton as newton  # noqa: E402

C = float(light.c)


def _state(objs):
    """SoA snapshot of a list of reference objects."""
    out = {}
    for f in ("r", "v", "dr", "dv"):
        out[f] = np.array([np.asarray(getattr(o, f), dtype=np.float64).reshape(3) for o in objs],
                          dtype=np.float64).reshape(len(objs), 3)
    out["E"] = np.array([float(np.asarray(getattr(o, "E", np.nan))) for o in objs], dtype=np.float64)
    out["uid"] = np.array([getattr(o, "uid", -1) for o in objs], dtype=np.int64)
    return out


def _save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %7.1f KiB  (%d arrays)" % (name + ".npz", os.path.getsize(path) / 1024, len(arrs)))


# ----------------------------------------------------------------------------------------------
# G1  NewtonianKinematicsStep (newton.py:10-16), reference CPU path unmodified
# ----------------------------------------------------------------------------------------------
def g1_newton():
    out = {}
    rng = np.random.RandomState(101)
    N = 2048
    r0 = rng.uniform(-1e6, 1e6, (N, 3))
    d = rng.normal(size=(N, 3))
    v0 = C * d / np.linalg.norm(d, axis=1)[:, None]
    v0[:64] *= rng.uniform(1e-12, 1e3, (64, 1))  # generic Objects need not move at c
    v0[64:72] = 0.0
    out["r_init"], out["v_init"] = r0, v0
    for ci, dt in enumerate((1e-3, 5e-3, 1e-5)):
        sim = physicl.Simulation(cl_on=False)
        for i in range(N):
            sim.add_obj(physicl.Object(r=physicl.Measurement(r0[i].copy(), "m**1"),
                                       v=physicl.Measurement(v0[i].copy(), "m**1 s**-1")))
        sim.dt = np.double(dt)
        st = newton.NewtonianKinematicsStep()
        for k in range(1, 11):
            st.run(sim)
            if k in (1, 10):
                s = _state(sim.objects)
                out["c%d_dt" % ci] = np.float64(dt)
                out["c%d_r_after%d" % (ci, k)] = s["r"]
                out["c%d_dr_after%d" % (ci, k)] = s["dr"]
    # config 1 of BASELINE.json: 1e4 photons, v=(c,0,0), dt=1e-3, 100 steps (test_light.py:19-24,32-33)
    sim = physicl.Simulation(cl_on=False)
    for i in range(10000):
        sim.add_obj(light.PhotonObject(v=np.array([light.c, 0, 0], dtype=np.double), E=np.double(1)))
    sim.dt = np.double(0.001)
    st = newton.NewtonianKinematicsStep()
    for k in range(100):
        st.run(sim)
    s = _state(sim.objects)
    assert (s["r"] == s["r"][0]).all() and (s["dr"] == s["dr"][0]).all()
    out["cfg1_r_after100"] = s["r"][0]
    out["cfg1_dr_after100"] = s["dr"][0]
    _save("g1_newton", **out)


# ----------------------------------------------------------------------------------------------
# G2  ScatterIsotropicStep, OpenCL path (light.py:281-331 + __init__.py:602-664)
# ----------------------------------------------------------------------------------------------
def _photons(N, rng, E_lo=None, E_hi=None, r_box=None):
    objs = []
    for i in range(N):
        E = np.double(1.0) if E_lo is None else np.double(E_lo + (E_hi - E_lo) * rng.power(3))
        p = light.PhotonObject(v=np.array([light.c, 0, 0], dtype=np.double), E=E, uid=i)
        if r_box is not None:
            p.r = physicl.Measurement(rng.uniform(r_box[0], r_box[1], 3), "m**1")
        objs.append(p)
    return objs


def _run_iso(tag, N, K, dt, seed, step_kwargs, E_range=None, r_box=None, planes=()):
    out = {}
    rng = np.random.RandomState(seed + 1000)
    sim = physicl.Simulation(cl_on=True)
    sim.add_objs(_photons(N, rng, *(E_range or (None, None)), r_box=r_box))
    out["init_r"] = _state(sim.objects)["r"]
    out["init_E"] = _state(sim.objects)["E"]
    upd = physicl.UpdateTimeStep(lambda s: np.double(dt))
    nk = newton.NewtonianKinematicsStep()
    sc = light.ScatterIsotropicStep(**step_kwargs)
    sign = light.ScatterSignMeasureStep(None, True)
    meas = light.ScatterMeasureStep(None, True, [np.array(p, dtype=np.double) for p in planes])
    sim.t, sim.dt, sim.ts = 0, 0, []
    np.random.seed(seed)
    out["seed"], out["dt"], out["K"] = np.int64(seed), np.float64(dt), np.int64(K)
    out["A_user"], out["n_user"] = np.float64(sc.A), np.float64(sc.n)
    for k in range(K):
        del KERNEL_LOG[:]
        upd.run(sim)
        nk.run(sim)
        sc.run(sim)
        sign.run(sim)
        meas.run(sim)
        (rec,) = KERNEL_LOG
        if k == 0:
            out["argnames"] = np.array(rec["argnames"])
        for an, av in rec["args"].items():
            out["k%d_%s" % (k, an)] = av
        s = _state(sim.objects)
        for f in ("r", "v", "dr", "dv"):
            out["k%d_post_%s" % (k, f)] = s[f]
    out["sign_rows"] = np.array(sign.data, dtype=np.float64)
    out["measure_rows"] = np.array(meas.data, dtype=np.float64)
    out["planes"] = np.array(planes, dtype=np.float64).reshape(-1, 3)
    _save("g2_iso_" + tag, **out)


def g2_iso():
    E_lo = float(light.E_from_wavelength(700e-9))
    E_hi = float(light.E_from_wavelength(200e-9))
    # base: constants of test/test_light.py:34 (pcoll ~ 0.2998)
    _run_iso("base", 4096, 4, 1e-3, 7, dict(A=np.double(0.001), n=np.double(0.001)),
             planes=[[3e5, np.nan, np.nan], [np.nan, 0.0, np.nan], [np.nan, np.nan, -1e5]])
    # wavelength-dependent: constants of examples/variable_n_scattering.ipynb:56, constant n
    _run_iso("lambda", 4096, 3, 5e-3, 8,
             dict(n=0.000000000000001, A=0.0000000000000000001, wavelength_dep_scattering=True),
             E_range=(E_lo, E_hi))
    # variable n + wavelength: literal expression of examples/variable_n_scattering.ipynb:30
    _run_iso("varn", 4096, 3, 1e-9, 9,
             dict(n=0.000000000000001, A=0.0000000000000000001, wavelength_dep_scattering=True,
                  variable_n=True, variable_n_fn="0.000000001 * exp(r0[gid] - 5)"),
             E_range=(E_lo, E_hi), r_box=(-10.0, 10.0))
    # variable n, radial profile in the style of examples/presentation_example_2.ipynb:41 (sqrt/pow of r0..r2)
    _run_iso("varn_radial", 2048, 2, 1e-9, 10,
             dict(n=0.5, A=123.0, variable_n=True,
                  variable_n_fn="2.5 * exp(-1 * (sqrt(pow(r0[gid], 2) + pow(r1[gid], 2) + pow(r2[gid], 2)) - 6.0)/(3.5))"),
             r_box=(-8.0, 8.0))
    # config-3 literal regime: exp() overflows to +inf / underflows to 0 (SURVEY 8(d) caveat)
    _run_iso("varn_overflow", 1024, 3, 5e-3, 11,
             dict(n=0.000000000000001, A=0.0000000000000000001, wavelength_dep_scattering=True,
                  variable_n=True, variable_n_fn="0.000000001 * exp(r0[gid] - 5)"),
             E_range=(E_lo, E_hi))


# ----------------------------------------------------------------------------------------------
# G4  ScatterDeleteStep (light.py:231-260) and ScatterDeleteStepReference (light.py:164-205)
# ----------------------------------------------------------------------------------------------
def _run_delete(step_cls, N, dt, seed, A, n, planes):
    rng = np.random.RandomState(seed + 1000)
    sim = physicl.Simulation(cl_on=True)
    sim.add_objs(_photons(N, rng))
    upd = physicl.UpdateTimeStep(lambda s: np.double(dt))
    nk = newton.NewtonianKinematicsStep()
    sc = step_cls(np.double(n), np.double(A))
    meas = light.ScatterMeasureStep(None, True, [np.array(p, dtype=np.double) for p in planes])
    sign = light.ScatterSignMeasureStep(None, True)
    sim.t, sim.dt, sim.ts = 0, 0, []
    np.random.seed(seed)
    per_step = []
    k = 0
    while len(sim.objects) > 0:
        del KERNEL_LOG[:]
        upd.run(sim)
        nk.run(sim)
        sc.run(sim)
        meas.run(sim)
        sign.run(sim)
        (rec,) = KERNEL_LOG
        per_step.append((rec, _state(sim.objects)))
        k += 1
    return per_step, np.array(meas.data, dtype=np.float64), np.array(sign.data, dtype=np.float64)


def g4_delete():
    N, dt, seed, A, n = 4096, 1e-3, 21, 0.001, 0.001
    planes = [[1.0 / (n * A), np.nan, np.nan], [3e5, np.nan, np.nan]]
    a, meas_a, sign_a = _run_delete(light.ScatterDeleteStep, N, dt, seed, A, n, planes)
    b, meas_b, sign_b = _run_delete(light.ScatterDeleteStepReference, N, dt, seed, A, n, planes)
    assert len(a) == len(b) and (meas_a == meas_b).all()
    out = {"seed": np.int64(seed), "dt": np.float64(dt), "A_user": np.float64(A), "n_user": np.float64(n),
           "N": np.int64(N), "K": np.int64(len(a)), "measure_rows": meas_a, "sign_rows": sign_a,
           "planes": np.array(planes, dtype=np.float64),
           "argnames_clprogram": np.array(a[0][0]["argnames"]),
           "argnames_reference": np.array(b[0][0]["argnames"])}
    for k, ((ra, sa), (rb, sb)) in enumerate(zip(a, b)):
        fa = ra["args"]["res"]
        fb = rb["args"]["result"]
        assert fa.dtype == np.int32 and (fa == fb).all() and (sa["uid"] == sb["uid"]).all()
        for nm in ("d0", "d1", "d2", "rand"):
            out["k%d_%s" % (k, nm)] = ra["args"][nm]
        out["k%d_flags" % k] = fa
        out["k%d_survivor_uid" % k] = sa["uid"]
        if k < 3:
            out["k%d_post_r" % k] = sa["r"]
    _save("g4_delete", **out)


# ----------------------------------------------------------------------------------------------
# G3  the reference's CPU paths of the light steps, cl_on=False, unmodified reference code:
#     ScatterIsotropicStep.__run_py (light.py:335-350): RNG order rand, then -- only on a hit -- phi, then theta;
#     dv = v_old on a hit; variable_n ignored.   ScatterDeleteStepReference.__run_py (light.py:216-223): removes from
#     the list it is iterating over, so the element after every removal is skipped (and draws no random number).
# ----------------------------------------------------------------------------------------------
def _num(x, shape=None):
    a = np.asarray(x, dtype=np.float64)
    return a.reshape(shape) if shape is not None else a


def _run_iso_py(tag, N, K, dt, seed, step_kwargs, E_range=None, with_objects=False):
    out = {}
    rng = np.random.RandomState(seed + 1000)
    sim = physicl.Simulation(cl_on=False)
    objs = _photons(N, rng, *(E_range or (None, None)))
    if with_objects:                       # plain Objects in between: moved by Newton, skipped by the light step (light.py:337)
        for i in range(0, N, 5):
            objs[i] = physicl.Object(v=physicl.Measurement(rng.normal(size=3) * 1e8, "m**1 s**-1"), uid=i)
    sim.add_objs(objs)
    out["init_E"] = _state(sim.objects)["E"]
    out["init_v"] = _state(sim.objects)["v"]
    out["is_photon"] = np.array([type(o) is light.PhotonObject for o in sim.objects])
    upd = physicl.UpdateTimeStep(lambda s: np.double(dt))
    nk = newton.NewtonianKinematicsStep()
    sc = light.ScatterIsotropicStep(**step_kwargs)
    sign = light.ScatterSignMeasureStep(None, True)
    sim.t, sim.dt, sim.ts = 0, 0, []
    np.random.seed(seed)
    out["seed"], out["dt"], out["K"] = np.int64(seed), np.float64(dt), np.int64(K)
    out["A_user"], out["n_user"] = np.float64(sc.A), np.float64(sc.n)
    for k in range(K):
        upd.run(sim)
        nk.run(sim)
        sc.run(sim)
        sign.run(sim)
        for f in ("r", "v", "dr", "dv"):
            out["k%d_post_%s" % (k, f)] = np.array([_num(getattr(o, f), 3) for o in sim.objects])
    out["sign_rows"] = np.array(sign.data, dtype=np.float64)
    out["next_draw"] = np.float64(np.random.random())      # where the global MT19937 stream stands after K steps
    _save("g3_iso_py_" + tag, **out)


def g3_iso_py():
    E_lo = float(light.E_from_wavelength(700e-9))
    E_hi = float(light.E_from_wavelength(200e-9))
    _run_iso_py("base", 2048, 4, 1e-3, 31, dict(A=np.double(0.001), n=np.double(0.001)))
    _run_iso_py("lambda", 1024, 3, 5e-3, 32,
                dict(n=0.000000000000001, A=0.0000000000000000001, wavelength_dep_scattering=True), E_range=(E_lo, E_hi))
    # variable_n is silently ignored by the CPU path ("this does not support variable n scattering", light.py:334)
    _run_iso_py("varn_ignored", 1024, 2, 1e-3, 33,
                dict(A=np.double(0.001), n=np.double(0.0007), variable_n=True, variable_n_fn="0.000000001 * exp(r0[gid] - 5)"),
                with_objects=True)


def g3_delete_py():
    N, dt, seed, A, n = 2000, 1e-3, 41, 0.001, 0.001
    rng = np.random.RandomState(seed + 1000)
    sim = physicl.Simulation(cl_on=False)
    objs = _photons(N, rng)
    for i in range(7, N, 11):
        objs[i] = physicl.Object(v=physicl.Measurement(rng.normal(size=3) * 1e8, "m**1 s**-1"), uid=i)
    sim.add_objs(objs)
    out = {"seed": np.int64(seed), "dt": np.float64(dt), "A_user": np.float64(A), "n_user": np.float64(n), "N": np.int64(N),
           "is_photon": np.array([type(o) is light.PhotonObject for o in sim.objects]),
           "init_v": _state(sim.objects)["v"]}
    upd = physicl.UpdateTimeStep(lambda s: np.double(dt))
    nk = newton.NewtonianKinematicsStep()
    sc = light.ScatterDeleteStepReference(np.double(n), np.double(A))
    sim.t, sim.dt, sim.ts = 0, 0, []
    np.random.seed(seed)
    K = 6
    for k in range(K):
        upd.run(sim)
        nk.run(sim)
        sc.run(sim)
        out["k%d_survivor_uid" % k] = _state(sim.objects)["uid"]
        out["k%d_post_r" % k] = _state(sim.objects)["r"]
    out["K"] = np.int64(K)
    out["next_draw"] = np.float64(np.random.random())
    _save("g3_delete_py", **out)


# ----------------------------------------------------------------------------------------------
# G5  TracePathMeasureStep (light.py:433-483), the reference's own run / terminate, behind the OpenCL light steps:
#     (a) [UpdateTimeStep, Newton, ScatterIsotropic, TracePath(trace_dv=True)]  K passes, nobody leaves
#     (b) [UpdateTimeStep, Newton, ScatterDelete, TracePath(id_info_fn=uid)]     until the list is empty
#     The table terminate() builds is ragged (a row = id info [, freq], 3 NaN SCALARS per pass before the first sight, one
#     position VECTOR per pass seen, 3 NaN scalars per pass after the last): stored per row as (info, freq, scalars in front,
#     positions, scalars behind).
# ----------------------------------------------------------------------------------------------
def _trace_table(tr, trace_dv):
    rows = tr.data
    out = {"t_row": np.array(rows[0][1:], dtype=np.float64), "label": np.array(rows[0][0])}
    info, freq, lead, trail, pos_len, pos = [], [], [], [], [], []
    for row in rows[1:]:
        info.append(str(row[0]))
        k = 1
        if trace_dv:
            freq.append(int(row[1]))
            k = 2
        body = row[k:]
        a = 0
        while a < len(body) and np.ndim(body[a]) == 0:
            assert np.isnan(body[a])
            a += 1
        b = len(body)
        while b > a and np.ndim(body[b - 1]) == 0:
            assert np.isnan(body[b - 1])
            b -= 1
        vec = [np.asarray(x, dtype=np.float64).reshape(3) for x in body[a:b]]
        lead.append(a)
        trail.append(len(body) - b)
        pos_len.append(len(vec))
        pos.extend(vec)
    out.update(info=np.array(info), freq=np.array(freq, dtype=np.int64), lead_scalars=np.array(lead, dtype=np.int64),
               trail_scalars=np.array(trail, dtype=np.int64), pos_len=np.array(pos_len, dtype=np.int64),
               pos=np.array(pos, dtype=np.float64).reshape(-1, 3))
    return out


def g5_trace():
    out = {}
    # (a) isotropic, trace_dv: constants of test/test_light.py:34
    N, K, dt, seed = 96, 6, 1e-3, 31
    rng = np.random.RandomState(seed + 1000)
    sim = physicl.Simulation(cl_on=True)
    sim.add_objs(_photons(N, rng))
    upd = physicl.UpdateTimeStep(lambda s: np.double(dt))
    nk = newton.NewtonianKinematicsStep()
    sc = light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001))
    tr = light.TracePathMeasureStep(None, trace_dv=True)
    sim.t, sim.dt, sim.ts = 0, 0, []
    np.random.seed(seed)
    for k in range(K):
        for st in (upd, nk, sc, tr):
            st.run(sim)
    tr.terminate(sim)
    for key, v in _trace_table(tr, True).items():
        out["iso_" + key] = v
    out.update(iso_N=np.int64(N), iso_K=np.int64(K), iso_dt=np.float64(dt), iso_seed=np.int64(seed), iso_A_user=np.float64(0.001),
               iso_n_user=np.float64(0.001))
    # (b) delete until empty, ids named by a user function
    N, dt, seed, A, n = 96, 1e-3, 32, 0.001, 0.001
    rng = np.random.RandomState(seed + 1000)
    sim = physicl.Simulation(cl_on=True)
    sim.add_objs(_photons(N, rng))
    upd = physicl.UpdateTimeStep(lambda s: np.double(dt))
    nk = newton.NewtonianKinematicsStep()
    de = light.ScatterDeleteStep(np.double(n), np.double(A))
    tr = light.TracePathMeasureStep(None, id_info_fn=lambda o: "photon %d" % o.uid)
    sim.t, sim.dt, sim.ts = 0, 0, []
    np.random.seed(seed)
    alive = []
    while len(sim.objects) > 0:
        for st in (upd, nk, de, tr):
            st.run(sim)
        alive.append(len(sim.objects))
    tr.terminate(sim)
    for key, v in _trace_table(tr, False).items():
        out["del_" + key] = v
    out.update(del_N=np.int64(N), del_dt=np.float64(dt), del_seed=np.int64(seed), del_A_user=np.float64(A), del_n_user=np.float64(n),
               del_alive=np.array(alive, dtype=np.int64))
    # (c) photons that join in the middle of the run (a user Step adds one in pass 2 and two in pass 4, in front of the trace
    #     step): their rows start with 3 NaN scalars per pass they missed
    N, K, dt, seed = 48, 6, 1e-3, 33
    rng = np.random.RandomState(seed + 1000)
    sim = physicl.Simulation(cl_on=True)
    sim.add_objs(_photons(N, rng))

    class Joiner(physicl.Step):
        def __init__(self):
            self.k = 0

        def run(self, sim):
            for j in range({2: 1, 4: 2}.get(self.k, 0)):
                sim.add_obj(light.PhotonObject(v=np.array([0, light.c, 0], dtype=np.double), E=np.double(1.0), uid=900 + 10 * self.k + j))
            self.k += 1

    upd = physicl.UpdateTimeStep(lambda s: np.double(dt))
    nk = newton.NewtonianKinematicsStep()
    sc = light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001))
    jn = Joiner()
    tr = light.TracePathMeasureStep(None, id_info_fn=lambda o: "uid %d" % o.uid)
    sim.t, sim.dt, sim.ts = 0, 0, []
    np.random.seed(seed)
    for k in range(K):
        for st in (upd, nk, sc, jn, tr):
            st.run(sim)
    tr.terminate(sim)
    for key, v in _trace_table(tr, False).items():
        out["join_" + key] = v
    out.update(join_N=np.int64(N), join_K=np.int64(K), join_dt=np.float64(dt), join_seed=np.int64(seed))
    _save("g5_trace", **out)


# ----------------------------------------------------------------------------------------------
# G7  set-up functions of physicl/light.py: E_from_wavelength / wavelength_from_E (41-51), planck_distribution (53-60),
#     planck_probability (63-64), planck_phot_distribution (73-104; returns None where the draw lies under the first
#     bin's mass), generate_photons (112-128) and generate_photons_from_E (109-110) under recorded seeds
# ----------------------------------------------------------------------------------------------
def g7_setup():
    out = {}
    lam = np.array([200e-9, 350e-9, 700e-9, 2500e-9])
    out["lam"] = lam
    out["E_of_lam"] = np.array([_num(light.E_from_wavelength(x)) for x in lam])
    out["lam_of_E"] = np.array([_num(light.wavelength_from_E(x)) for x in out["E_of_lam"]])
    Es, Ts = np.array([7.9e-20, 3e-19, 5.5e-19, 9.9e-19]), np.array([300.0, 5778.0, 12000.0])
    out["pd_E"], out["pd_T"] = Es, Ts
    out["pd"] = np.array([[_num(light.planck_distribution(np.double(e), np.double(t))) for t in Ts] for e in Es])
    out["pd_measurement_args"] = np.array(_num(light.planck_distribution(physicl.Measurement(np.double(3e-19), "J**1"), physicl.Measurement(np.double(5778.0), "K**1"))))
    lo, hi, T, bins = float(_num(light.E_from_wavelength(2500e-9))), float(_num(light.E_from_wavelength(200e-9))), 5778.0, 40
    out.update(pp_lo=np.float64(lo), pp_hi=np.float64(hi), pp_T=np.float64(T), pp_bins=np.int64(bins))
    grid = np.linspace(lo, hi, bins)
    out["pp_mass"] = np.array([light.planck_probability(grid[k], grid[k + 1], T)[0] for k in range(bins - 1)])
    np.random.seed(41)
    draws = [light.planck_phot_distribution(lo, hi, T, bins) for _ in range(400)]
    out["ppd_seed"] = np.int64(41)
    out["ppd_none"] = np.array([d is None for d in draws])
    out["ppd_E"] = np.array([np.nan if d is None else _num(d) for d in draws])
    out["ppd_next_random"] = np.float64(np.random.random())
    out["ppd_cdf"] = np.array(light.last_planck_cdf)
    # generate_photons: default sampler, and a user sampler
    np.random.seed(42)
    ph = light.generate_photons(64, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9))
    out["gp_seed"] = np.int64(42)
    s = _state(ph)
    out["gp_E"], out["gp_v"], out["gp_r"] = s["E"], s["v"], s["r"]
    out["gp_next_random"] = np.float64(np.random.random())
    np.random.seed(43)
    ph = light.generate_photons(32, fn=lambda: np.random.random() ** 2, min=1.0, max=3.0)
    out["gp_user_seed"], out["gp_user_E"] = np.int64(43), _state(ph)["E"]
    ph = light.generate_photons_from_E([np.double(1.5), np.double(2.5e-19)])
    s = _state(ph)
    out["gpe_E"], out["gpe_v"] = s["E"], s["v"]
    _save("g7_setup", **out)


# ----------------------------------------------------------------------------------------------
# G8  the files MeasureStep.terminate writes (__init__.py:360-378): the text of the counting measures' CSVs after a seeded
#     isotropic run and a delete run until empty (OUTPUT of the reference: rows "t, N, counts...", str() of numpy scalars)
# ----------------------------------------------------------------------------------------------
def g8_csv():
    out = {}
    tmp = tempfile.mkdtemp(prefix="pcl_g8_")
    planes = [[3e5, np.nan, np.nan], [np.nan, 0.0, np.nan]]
    for tag, seed in (("iso", 51), ("del", 52)):
        N, dt = 256, 1e-3
        rng = np.random.RandomState(seed + 1000)
        sim = physicl.Simulation(cl_on=True)
        sim.add_objs(_photons(N, rng))
        upd = physicl.UpdateTimeStep(lambda s: np.double(dt))
        nk = newton.NewtonianKinematicsStep()
        sc = (light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)) if tag == "iso"
              else light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
        f_sign, f_meas = os.path.join(tmp, tag + "_sign.csv"), os.path.join(tmp, tag + "_meas.csv")
        sign = light.ScatterSignMeasureStep(f_sign, True)
        meas = light.ScatterMeasureStep(f_meas, True, [np.array(p, dtype=np.double) for p in planes])
        meas_no_n = light.ScatterMeasureStep(os.path.join(tmp, tag + "_meas_no_n.csv"), False, [np.array(planes[0], dtype=np.double)])
        sim.t, sim.dt, sim.ts = 0, 0, []
        np.random.seed(seed)
        k = 0
        while (k < 5) if tag == "iso" else (len(sim.objects) > 0):
            for st in (upd, nk, sc, sign, meas, meas_no_n):
                st.run(sim)
            k += 1
        for st in (sign, meas, meas_no_n):
            st.terminate(sim)
        out[tag + "_sign_csv"] = np.array(open(f_sign).read())
        out[tag + "_meas_csv"] = np.array(open(f_meas).read())
        out[tag + "_meas_no_n_csv"] = np.array(open(os.path.join(tmp, tag + "_meas_no_n.csv")).read())
        out.update({tag + "_N": np.int64(N), tag + "_dt": np.float64(dt), tag + "_seed": np.int64(seed), tag + "_passes": np.int64(k)})
    out["planes"] = np.array(planes, dtype=np.float64)
    _save("g8_csv", **out)


# ----------------------------------------------------------------------------------------------
# G9  Simulation.run itself (__init__.py:501-524), start() / join() on its own thread: steps registered out of index order
#     run in REGISTRATION order (the dict's), a user Step that removes an object and later adds one through the simulation's
#     own methods, the clock (t, dt reset to int 0; ts), get_state() after the run
# ----------------------------------------------------------------------------------------------
def g9_run():
    out = {}
    N, dt, seed = 64, 1e-3, 61
    rng = np.random.RandomState(seed + 1000)
    sim = physicl.Simulation(cl_on=True, exit=lambda s: s.t >= 0.0075)
    sim.add_objs(_photons(N, rng))
    probe_rows = []

    class Probe(physicl.Step):
        def __init__(self):
            self.passes = 0
            self.terminated_at = None

        def run(self, sim):
            if self.passes == 2:
                sim.remove_obj(sim.objects[0])
            if self.passes == 4:
                sim.add_obj(light.PhotonObject(v=np.array([light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=1000))
            probe_rows.append((float(sim.t), float(sim.dt), [o.uid for o in sim.objects],
                               [float(np.asarray(o.r)[0]) for o in sim.objects]))
            self.passes += 1

        def terminate(self, sim):
            self.terminated_at = float(sim.t)

    sign = light.ScatterSignMeasureStep(None, True)
    probe = Probe()
    sim.add_step(5, sign)                                             # registered first: runs first in every pass
    sim.add_step(0, physicl.UpdateTimeStep(lambda s: np.double(dt)))
    sim.add_step(1, newton.NewtonianKinematicsStep())
    sim.add_step(2, light.ScatterDeleteStep(np.double(0.0005), np.double(0.001)))
    sim.add_step(7, probe)
    try:
        sim.add_step(7, probe)
        out["duplicate_index_error"] = np.array("none")
    except BaseException as e:                                        # (the reference raises a NameError here: IndexException is undefined)
        out["duplicate_index_error"] = np.array(type(e).__name__)
    np.random.seed(seed)
    sim.start()
    sim.join()
    out["sign_rows"] = np.array(sign.data, dtype=np.float64)
    out["ts"] = np.array(sim.ts, dtype=np.float64)
    out["probe_t"] = np.array([p[0] for p in probe_rows])
    out["probe_dt"] = np.array([p[1] for p in probe_rows])
    out["probe_n"] = np.array([len(p[2]) for p in probe_rows], dtype=np.int64)
    out["probe_uids"] = np.array([u for p in probe_rows for u in p[2]], dtype=np.int64)
    out["probe_r0"] = np.array([x for p in probe_rows for x in p[3]], dtype=np.float64)
    out["terminated_at"] = np.float64(probe.terminated_at)
    st = sim.get_state()
    out["state_keys"] = np.array(sorted(st.keys()))
    out["state_objects"], out["state_t"], out["state_dt"] = np.int64(st["objects"]), np.float64(st["t"]), np.float64(st["dt"])
    out["running_after"] = np.bool_(sim.running)
    out["next_random"] = np.float64(np.random.random())
    try:
        sim.remove_step(7)
        out["steps_after_remove"] = np.array(list(sim.steps.keys()), dtype=np.int64)
    except BaseException as e:
        out["steps_after_remove"] = np.array([-1], dtype=np.int64)
    out.update(N=np.int64(N), dt=np.float64(dt), seed=np.int64(seed))
    _save("g9_run", **out)


# ----------------------------------------------------------------------------------------------
# G10  the kernel-glue classes with a USER's kernel (CLInput / CLOutput / CLProgram, __init__.py:543-664): a Step of this
#      repo's own making (an absorber with its own kernel text; every input type: obj, obj_def, obj_action with a
#      ``continue``, obj_track, const; an int and a double output) run by the reference for three passes
# ----------------------------------------------------------------------------------------------
ABSORB_BODY = """
    int gid = get_global_id(0);
    double path = sqrt(d0[gid] * d0[gid] + d1[gid] * d1[gid] + d2[gid] * d2[gid]);
    gone[gid] = (sigma * path >= u[gid]) ? 1 : 0;
    depth[gid] = sigma * path + 0.25 * e2[gid];
"""


def g10_clprogram():
    out = {}
    N, seed, sigma = 300, 71, np.double(1e-6)

    class Absorber(physicl.Step):
        def __init__(self):
            self.prog, self.outs = None, []

        def run(self, sim):
            if self.prog is None:
                skip = physicl.CLInput(name="only_photons", type="obj_action",
                                       code="if type(obj) != physicl.light.PhotonObject:\n \t\t continue")
                d = [physicl.CLInput(name="d%d" % k, type="obj", obj_attr="dr[%d]" % k) for k in range(3)]
                u = physicl.CLInput(name="u", type="obj_def", obj_def="np.random.random()")
                e2 = physicl.CLInput(name="e2", type="obj_def", obj_def="obj.E * 2")
                sg = physicl.CLInput(name="sigma", type="const", const_value=str(sigma))
                who = physicl.CLInput(name="who", type="obj_track", obj_track="obj")
                self.prog = physicl.CLProgram(sim, "absorb", ABSORB_BODY)
                self.prog.prep_metadata = [skip] + d + [u, e2, who, sg]
                self.prog.output_metadata = [physicl.CLOutput(name="gone", ctype="int"), physicl.CLOutput(name="depth")]
                self.prog.build_kernel()
            res = self.prog.run()
            self.outs.append({k: np.array(v) for k, v in res.items()})
            for idx, x in enumerate(res["gone"]):
                if x == 1:
                    sim.remove_obj(self.prog.who[idx])

    rng = np.random.RandomState(seed + 1000)
    sim = physicl.Simulation(cl_on=True, exit=lambda s: s.t >= 0.0025)
    objs = _photons(N, rng, 1.0, 3.0)
    objs.insert(10, physicl.Object(v=physicl.Measurement([5.0, 0, 0], "m**1 s**-1"), uid=-1))
    sim.add_objs(objs)
    out["init_E"] = np.array([float(np.asarray(getattr(o, "E", np.nan))) for o in objs])
    ab = Absorber()
    sim.add_step(0, physicl.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, newton.NewtonianKinematicsStep())
    sim.add_step(2, ab)
    np.random.seed(seed)
    sim.start()
    sim.join()
    out["passes"] = np.int64(len(ab.outs))
    for k, o in enumerate(ab.outs):
        out["k%d_gone" % k], out["k%d_depth" % k] = o["gone"], o["depth"]
        out["k%d_keys" % k] = np.array(sorted(o.keys()))
    out["survivor_uid"] = np.array([o.uid for o in sim.objects], dtype=np.int64)
    out["next_random"] = np.float64(np.random.random())
    out.update(N=np.int64(N), seed=np.int64(seed), sigma=np.float64(sigma))
    _save("g10_clprogram", **out)


# ----------------------------------------------------------------------------------------------
# G11  BASELINE configs[4]'s loop by the reference: [UpdateTimeStep, Newton, ScatterIsotropic, sign measure, Newton,
#      ScatterDelete, plane measure] -- one np.random stream feeds both light steps (three draws per photon in the
#      isotropic step, then one per photon in the delete step, every pass)
# ----------------------------------------------------------------------------------------------
def g11_mixed():
    out = {}
    N, dt, seed, K = 512, 1e-3, 81, 7
    planes = [[3e5, np.nan, np.nan], [np.nan, 0.0, np.nan]]
    rng = np.random.RandomState(seed + 1000)
    sim = physicl.Simulation(cl_on=True)
    sim.add_objs(_photons(N, rng))
    steps = [physicl.UpdateTimeStep(lambda s: np.double(dt)), newton.NewtonianKinematicsStep(),
             light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)), light.ScatterSignMeasureStep(None, True),
             newton.NewtonianKinematicsStep(), light.ScatterDeleteStep(np.double(0.0004), np.double(0.001)),
             light.ScatterMeasureStep(None, True, [np.array(p, dtype=np.double) for p in planes])]
    sim.t, sim.dt, sim.ts = 0, 0, []
    np.random.seed(seed)
    uids = []
    for k in range(K):
        for st in steps:
            st.run(sim)
        uids.append(np.array([o.uid for o in sim.objects], dtype=np.int64))
    s = _state(sim.objects)
    out.update(sign_rows=np.array(steps[3].data, dtype=np.float64), measure_rows=np.array(steps[6].data, dtype=np.float64),
               planes=np.array(planes, dtype=np.float64), final_uid=s["uid"], final_r=s["r"], final_v=s["v"],
               alive=np.array([len(u) for u in uids], dtype=np.int64), uid_after_pass_2=uids[2],
               next_random=np.float64(np.random.random()), N=np.int64(N), dt=np.float64(dt), seed=np.int64(seed), K=np.int64(K))
    _save("g11_mixed", **out)


# ----------------------------------------------------------------------------------------------
# G12  plain Objects among the photons (OpenCL paths): the light steps skip them -- no random number drawn for them
#      (light.py:233, 283) --, Newton moves them, the measures count them
# ----------------------------------------------------------------------------------------------
def g12_kinds():
    out = {}
    planes = [[3e5, np.nan, np.nan], [np.nan, -0.004, np.nan]]
    for tag, seed in (("iso", 91), ("del", 92)):
        N, dt = 210, 1e-3
        rng = np.random.RandomState(seed + 1000)
        objs = _photons(N, rng)
        is_obj = np.zeros(N, dtype=bool)
        for i in range(3, N, 7):
            objs[i] = physicl.Object(v=physicl.Measurement(np.array([5.0, -3.0, 2.0]) * (1 + i % 3), "m**1 s**-1"), uid=i)
            is_obj[i] = True
        sim = physicl.Simulation(cl_on=True)
        sim.add_objs(objs)
        upd = physicl.UpdateTimeStep(lambda s: np.double(dt))
        nk = newton.NewtonianKinematicsStep()
        sc = (light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)) if tag == "iso"
              else light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
        sign = light.ScatterSignMeasureStep(None, True)
        meas = light.ScatterMeasureStep(None, True, [np.array(p, dtype=np.double) for p in planes])
        sim.t, sim.dt, sim.ts = 0, 0, []
        np.random.seed(seed)
        for k in range(6):
            for st in (upd, nk, sc, sign, meas):
                st.run(sim)
        s = _state(sim.objects)
        out.update({tag + "_is_obj": is_obj, tag + "_sign_rows": np.array(sign.data, dtype=np.float64),
                    tag + "_measure_rows": np.array(meas.data, dtype=np.float64), tag + "_final_uid": s["uid"],
                    tag + "_final_r": s["r"], tag + "_final_v": s["v"], tag + "_next_random": np.float64(np.random.random()),
                    tag + "_seed": np.int64(seed), tag + "_N": np.int64(N), tag + "_dt": np.float64(dt)})
    out["planes"] = np.array(planes, dtype=np.float64)
    _save("g12_kinds", **out)


# ----------------------------------------------------------------------------------------------
# G13  planes that lie EXACTLY where unscattered photons stop after a pass (x = k * fl(c * dt)): the crossing test is
#      ``r - dr <= loc <= r`` with both ends included (light.py:385-399), so such a photon is counted when it arrives and --
#      where fl(r - dr) gives the plane back -- again when it leaves; delete flow (nobody turns), 7 passes
# ----------------------------------------------------------------------------------------------
def g13_plane_edges():
    out = {}
    N, dt, seed = 160, 1e-3, 95
    d = np.double(299792458.0) * np.double(dt)
    planes = [[d * k, np.nan, np.nan] for k in (1, 2, 3, 5)] + [[-d, np.nan, np.nan], [np.nan, 0.0, np.nan], [np.nan, np.nan, 1.0]]
    rng = np.random.RandomState(seed + 1000)
    sim = physicl.Simulation(cl_on=True)
    sim.add_objs(_photons(N, rng))
    upd = physicl.UpdateTimeStep(lambda s: np.double(dt))
    nk = newton.NewtonianKinematicsStep()
    sc = light.ScatterDeleteStep(np.double(0.0003), np.double(0.001))
    meas = light.ScatterMeasureStep(None, True, [np.array(p, dtype=np.double) for p in planes])
    sim.t, sim.dt, sim.ts = 0, 0, []
    np.random.seed(seed)
    for k in range(7):
        for st in (upd, nk, sc, meas):
            st.run(sim)
    out.update(planes=np.array(planes, dtype=np.float64), measure_rows=np.array(meas.data, dtype=np.float64), N=np.int64(N),
               dt=np.float64(dt), seed=np.int64(seed), final_uid=_state(sim.objects)["uid"])
    _save("g13_plane_edges", **out)


# ----------------------------------------------------------------------------------------------
# G6  Measurement / code units (test/test_units.py:25-78, code_unit_scale_test.ipynb:55)
# ----------------------------------------------------------------------------------------------
def g6_units():
    M = physicl.Measurement
    out = {}

    def rec(key, m):
        out[key + "_code"] = np.asarray(m.view(np.ndarray), dtype=np.float64)
        out[key + "_scale"] = np.float64(m.scale)
        out[key + "_units"] = np.array(sorted("%s:%g" % kv for kv in m.units.items() if kv[1] != 0))
        out[key + "_value"] = np.asarray(m.value(), dtype=np.float64)

    rec("N5", M(5, "N**1"))
    rec("kgms2", M(5, "kg**1 m**1 s**-2"))
    rec("au1", M(1, "au**1"))
    rec("au_plus_m", M(1, "au**1") + M(149597870700, "m**1"))
    rec("m_plus_au", M(149597870700, "m**1") + M(1, "au**1"))
    rec("E633", light.E_from_wavelength(M(633e-9, "m**1")))
    rec("wl633", light.wavelength_from_E(light.E_from_wavelength(M(633e-9, "m**1"))))
    Eg = M(0, "J**1") + M(13.6, "eV**1")
    rec("Eg", Eg)
    rec("f", Eg / light.h)
    rec("l", light.c / (Eg / light.h))
    a, l, t = M(5, "kg**1 m**1 s**-2"), M(5, "au**1"), M(10, "min**2")
    rec("a_times_t", a * t)
    rec("a_times_l", a * l)
    rec("a_div_l", a / l)
    rec("a_sq", a ** 2)
    rec("min2", t)
    rec("eV", M(1, "eV**1"))
    rec("vec", M([1.5, -2.0, 3.25], "m**1 s**-1"))
    rec("c", light.c)
    rec("h", light.h)
    rec("kB", light.kB)
    out["str_c"] = np.array(str(light.c))
    out["str_h_upper"] = np.array(str(light.h).upper())
    out["fmt_n0"] = np.array("{}".format(M(2.5e25, "m**-3")))
    out["repr_vec"] = np.array(repr(M([1.5, -2.0, 3.25], "m**1 s**-1")))
    # class-global code scale: m -> 1e-3 (code_unit_scale_test.ipynb:55)
    M.set_code_scale("m", 0.001)
    try:
        c2 = M(np.double(299792458), "m**1 s**-1")
        h2 = M(np.double(6.62607015e-34), "J**1 s**1")
        rec("c_mscale", c2)
        rec("h_mscale", h2)
        out["str_c_mscale"] = np.array(str(c2))
        out["str_h_mscale_upper"] = np.array(str(h2).upper())
        rec("E200_mscale", (h2 * c2) / M(200e-9, "m**1"))
        nA = M(2.0e25, "m**-3") * M(5.1e-31, "m**2")
        rec("nA_mscale", nA)
        rec("inv_nA_mscale", 1 / nA)
    finally:
        M.reset_code_scale("m")
    _save("g6_units", **out)


# Expressions evaluated with the reference's Measurement (M), light module (light) and numpy (np);
# tests/test_units_parity.py evaluates the same strings with physicl_amd and compares.
UNIT_OPS = [
    "M(5, 'kg**1 m**1 s**-2')", "M(5, 'N**1')", "M(1, 'au**1')", "M(3, 'eV**1')", "M(10, 'min**2')",
    "M([1.5, -2.0, 3.25], 'm**1 s**-1')", "M(8.6e3, 'm')", "M(2, 'km**1')" if False else "M(2, 'L**1')",
    "M(1, 'au**1') + M(149597870700, 'm**1')", "M(149597870700, 'm**1') + M(1, 'au**1')",
    "M(0, 'J**1') + M(13.6, 'eV**1')", "(M(0, 'J**1') + M(13.6, 'eV**1')) / light.h",
    "light.c / ((M(0, 'J**1') + M(13.6, 'eV**1')) / light.h)",
    "light.E_from_wavelength(M(633e-9, 'm**1'))", "light.wavelength_from_E(light.E_from_wavelength(M(633e-9, 'm**1')))",
    "light.E_from_wavelength(200e-9)", "(light.h * light.c) / M(1e-19, 'J**1')",
    "((light.h * light.c) / M(1e-19, 'J**1')) ** -4",
    "M(5, 'N**1') * M(10, 'min**2')", "M(5, 'N**1') * M(5, 'au**1')", "M(5, 'N**1') / M(5, 'au**1')",
    "M(5, 'N**1') ** 2", "M(5, 'au**1') ** 2", "M(5, 'au**1') * M(5, 'au**1')", "np.sqrt(M(5, 'au**1'))",
    "np.sqrt(M(4, 'm**2'))", "M(5, 'au**1') / M(5, 'au**1')", "1 / M(5, 'au**1')", "M(5, 'au**1') + 1", "1 + M(5, 'au**1')",
    "M(5, 'au**1') - M(1, 'm**1')", "M([1, 2, 3], 'au**1').sum()", "np.sum(M([1, 2, 3], 'au**1'))",
    "M([light.c, 0, 0], 'm**1 s**-1') * 2", "2 * M([light.c, 0, 0], 'm**1 s**-1')",
    "M([light.c, 0, 0], 'm**1 s**-1') * np.double(1e-3)", "light.c * [1, 0, 0]", "-M([1.0, -2.0], 'm**1')",
    "abs(M(-5, 'm**1'))", "np.exp(M(1, 'm**1'))", "M(5, 'au**1') < 3", "M(5, 'au**1') == 5",
    "M(5, 'N**1') == M(5, 'kg**1 m**1 s**-2')", "M([1., 2., 3.], 'm**1')[1]", "M([1., 2., 3.], 'm**1')[0:2]",
    "M([1., 2., 3.], 'au**1').copy()", "M([M(1, 'au**1'), 2], 'm**1')", "M(M([1., 2.], 'm**1'), 'm**1')",
    "float(M(5, 'au**1'))", "M(5, 'au**1').value()", "M(5, 'au**1').valstr()", "M(5, 'au**1').fstr()",
    "M(5, 'au**1').unitstr()", "str(M(2.5e-7, 'm**1'))", "str(M([1e-9, 2.0], 'm**1'))", "repr(M(5, 'au**1') * M(5, 'au**1'))",
    "'{}'.format(M(2.5e25, 'm**-3'))", "'{:.3e}'.format(M(2.5e25, 'm**-3'))", "'{:.3f}'.format(M(2.5, 'm**1'))",
    "__import__('copy').deepcopy(M([1., 2.], 'au**1'))", "np.linalg.norm(M([light.c, 0, 0], 'm**1 s**-1'))",
    "np.linalg.norm(light.c)", "np.isnan(M([1., np.nan], 'm**1'))", "np.array_equal(M([0., 0.], 'm**1'), np.array([0, 0]))",
    "M(3.0, 'm**1 s**-1 s**-1')", "M(1, 'Pa**1')", "M(1, 'W**1')", "M(1, 'V**1')", "M(1, 'F**1')", "M(1, 'Ohm**1')",
    "M(1, 'T**1')", "M(1, 'H**1')", "M(1, 'd**1')", "M(1, 'ha**1')", "M(1, 't**1')", "M(1, 'Da**1')", "M(2, 'h**1')",
    "M(7, 'kat**1')", "M(7, 'Sv**1')", "M(1, 'lm**1')", "M(4, 'm ^ 2')", "M(4, 'm**2 kg**1')",
    "np.multiply(M(2, 'm**1'), M(3, 's**1'))", "np.add.reduce(M([1., 2.], 'min**1'))",
    "M(2, 'm**1') * M([1., 2.], 's**-1') + M([1., 1.], 'm**1 s**-1')",
    # Planck helpers (set-up time): density values and one bin mass, the inputs of the tabulated sampler
    "light.planck_distribution(M(3e-19, 'J**1'), 5778)", "light.planck_distribution(2.5e-19, M(5778, 'K**1'))",
    "light.planck_distribution(np.linspace(8e-20, 9e-19, 5), 5778)",
    "float(light.planck_probability(1e-19, 2e-19, 5778)[0])",
    # the comparisons the reference's own unit tests make (test/test_units.py), as recorded outcomes: whatever the
    # reference answers here -- including the comparison its last test gets wrong -- is what the build must answer
    "M(5, 'kg**1 m**1 s**-2') == M(5, 'N**1')", "M(5, 'kg**1 m**1 s**-2').units == M(5, 'N**1').units",
    "M(1, 'au**1') + M(149597870700 * 1, 'm**1') == M(2, 'au**1')",
    "M(149597870700 * 1, 'm**1') + M(1, 'au**1') == M(149597870700 * 2, 'm**1')",
    "light.PhotonObject(E=M(5, 'J**1'), v=M([light.c, 0, 0], 'm**1 s**-1')).E.units == {'L': 2, 'T': -2, 'M': 1}",
    "light.PhotonObject(E=M(5, 'J**1'), v=M([light.c, 0, 0], 'm**1 s**-1')).v.units == {'L': 1, 'T': -1}",
    "np.linalg.norm(light.PhotonObject(E=M(5, 'J**1'), v=M([light.c, 0, 0], 'm**1 s**-1')).v) == light.c",
    "light.E_from_wavelength(M(633e-9, 'm**1')) == (299792458 * 6.62607015e-34) / (633e-9)",
    "light.E_from_wavelength(M(633e-9, 'm**1')).units == {'L': 2, 'T': -2, 'M': 1}",
    "light.wavelength_from_E(light.E_from_wavelength(M(633e-9, 'm**1'))) == 633e-9",
    "sorted((k, v) for k, v in light.wavelength_from_E(light.E_from_wavelength(M(633e-9, 'm**1'))).units.items() if v != 0)",
    "(M(0, 'J**1') + M(13.6, 'eV**1')) == 1.602176634e-19 * 13.6",
    "((M(0, 'J**1') + M(13.6, 'eV**1')) / light.h) == (1.602176634e-19 * 13.6) / 6.62607015e-34",
    "sorted((k, v) for k, v in ((M(0, 'J**1') + M(13.6, 'eV**1')) / light.h).units.items() if v != 0)",
    "(light.c / ((M(0, 'J**1') + M(13.6, 'eV**1')) / light.h)) == 299792458 / ((1.602176634e-19 * 13.6) / 6.62607015e-34)",
    "sorted((k, v) for k, v in (light.c / ((M(0, 'J**1') + M(13.6, 'eV**1')) / light.h)).units.items() if v != 0)",
    "M(5, 'kg**1 m**1 s**-2') * M(10, 'min**2') == 50",
    "M(0, 'kg**1 m**1') + (M(5, 'kg**1 m**1 s**-2') * M(10, 'min**2')) == (60 ** 2) * 10 * 5",
    "M(5, 'kg**1 m**1 s**-2') * M(5, 'au**1') == 25",
    "(M(5, 'kg**1 m**1 s**-2') / M(5, 'au**1')).flat[0] == 5 / (5 * 149597870700)",
    "M(5, 'kg**1 m**1 s**-2') ** 2 == 25",
    "sorted((k, v) for k, v in (M(5, 'kg**1 m**1 s**-2') ** 2).units.items() if v != 0)",
    "np.sqrt(M(5, 'au**1')) == np.sqrt(5)",
    "M(0, 'm**1') + np.sqrt(M(5, 'au**1')) == np.sqrt(149597870700 * 5)",
]


def _describe(x):
    M = physicl.Measurement
    d = {"type": type(x).__name__}
    if isinstance(x, M):
        d["code"] = np.asarray(x.view(np.ndarray)).astype(np.float64).tolist()
        d["has_units"] = hasattr(x, "units")
        if hasattr(x, "units"):
            d["scale"] = float(np.asarray(x.scale))
            d["units"] = {k: float(np.asarray(v)) for k, v in x.units.items()}
            d["original_units"] = {k: float(np.asarray(v)) for k, v in x.original_units.items()}
            d["unitstr"] = x.unitstr()
    elif isinstance(x, np.ndarray):
        d["code"] = x.astype(np.float64).tolist()
    elif isinstance(x, (float, int, np.floating, np.integer, bool, np.bool_)):
        d["code"] = float(x)
    else:
        d["text"] = str(x)
    return d


def g6_unit_ops():
    import json
    M = physicl.Measurement
    out = {"default": [], "m_scale_1e-3": []}
    for key in out:
        if key != "default":
            M.set_code_scale("m", 0.001)
        try:
            for expr in UNIT_OPS:
                try:
                    res = _describe(eval(expr, {"M": M, "light": light, "np": np, "__import__": __import__}))
                except Exception as e:  # the reference raises here; the build must raise too
                    res = {"raises": type(e).__name__}
                out[key].append({"expr": expr, "result": res})
        finally:
            M.reset_code_scale("m")
    path = os.path.join(OUT, "g6_unit_ops.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote g6_unit_ops.json  %d expressions x 2 code scales" % len(UNIT_OPS))


# ----------------------------------------------------------------------------------------------
# G14  the public surface: every class and function the three modules define, parameter names / kinds, the defaults that
#      are plain literals, base class names (NAMES only -- no source text)
# ----------------------------------------------------------------------------------------------
def g14_api():
    import inspect
    import json

    def params(f):
        out = []
        for p in inspect.signature(f).parameters.values():
            d = p.default
            lit = None if d is inspect._empty else (repr(d) if isinstance(d, (bool, int, float, str, type(None), list)) else "<object>")
            out.append([p.name, str(p.kind), lit])
        return out

    api = {}
    for modname, mod in (("physicl", physicl), ("physicl.light", light), ("physicl.newton", newton)):
        for name, obj in sorted(vars(mod).items()):
            if name.startswith("_") or getattr(obj, "__module__", None) != modname:
                continue
            if inspect.isclass(obj):
                entry = {"kind": "class", "bases": [b.__name__ for b in obj.__bases__], "methods": {}}
                for mn, m in sorted(vars(obj).items()):
                    if callable(m) and (not mn.startswith("_") or mn == "__init__"):
                        try:
                            entry["methods"][mn] = params(m)
                        except (TypeError, ValueError):
                            pass
                api[modname + "." + name] = entry
            elif inspect.isfunction(obj):
                api[modname + "." + name] = {"kind": "function", "params": params(obj)}
    consts = {k: float(np.asarray(getattr(light, k))) for k in ("c", "h", "kB")}
    path = os.path.join(OUT, "g14_api.json")
    with open(path, "w") as f:
        json.dump({"api": api, "light_constants": consts}, f, indent=1, sort_keys=True)
    print("wrote g14_api.json  %d names" % len(api))


def main():
    only = sys.argv[1:]
    if only:                      # e.g. `make_golden.py g6_unit_ops` regenerates one fixture
        for name in only:
            globals()[name]()
        return
    g1_newton()
    g2_iso()
    g4_delete()
    g3_iso_py()
    g3_delete_py()
    g5_trace()
    g7_setup()
    g8_csv()
    g9_run()
    g10_clprogram()
    g11_mixed()
    g12_kinds()
    g13_plane_edges()
    g14_api()
    g6_units()
    g6_unit_ops()
    # provenance: hashes of the kernel texts the reference generated (no text stored)
    with open(os.path.join(OUT, "kernel_sources.sha256"), "w") as f:
        for k in sorted(_Program.sources):
            f.write("%s  %s\n" % (_Program.sources[k], k))
    print("kernel signatures seen:")
    for k in sorted(_Program.sources):
        print("  ", k)


if __name__ == "__main__":
    main()
