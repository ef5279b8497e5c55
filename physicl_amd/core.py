"""Simulation runtime: the reference's plugin API (physicl/__init__.py:293-541) on top of a particle
store that lives in HBM.

Same public names and call signatures as the reference -- ``Step.run(sim)`` / ``terminate(sim)``,
``UpdateTimeStep(fn)``, ``MeasureStep(out_fn)``, ``Object(**kw)``, ``Simulation(**kw)`` with
``add_step / add_obj / add_objs / remove_obj / remove_step / start / join / get_state`` -- so
scripts written for PhysiCL run unchanged.  What differs is where the particles are:

* The reference keeps a Python list of objects and, every step, gathers attributes into numpy
  arrays, copies them to the device, launches, copies back and writes attributes again
  (physicl/__init__.py:602-664, physicl/light.py:325-331).
* Here the particles are uploaded ONCE into structure-of-arrays device memory and every
  device-native step (Newton, scatter, delete, the counting measure steps) is a kernel launch on
  that memory.  ``sim.objects`` is a list-like proxy: ``len()`` is answered from the alive count,
  anything else (indexing, iteration, a user-written Step that loops over objects) first brings
  the state back into the Python objects, and the next device step re-uploads it.  User plugins
  therefore keep working, at the reference's speed; all-native step lists run at HBM speed.

There is no CPU implementation of the device-native steps.  ``cl_on=False`` -- which in the reference selects its
Python paths -- selects those paths' SEMANTICS here (the data-dependent np.random order and ``dv = v_old`` of
ScatterIsotropicStep.__run_py, physicl/light.py:335-350; the skip-after-removal iteration of
ScatterDeleteStepReference.__run_py, light.py:216-223), still executed by the HIP kernels: the device context is then
created when the first device step needs it, and ``cl_ctx`` / ``cl_q`` stay None as in the reference.
"""
import copy
import collections
import os
import threading
import time

import numpy as np

from .units import Measurement
from .ahead import AheadView as _AheadView, NotAhead as _NotAhead, clock_only as _clock_only, count_use as _count_use

HOST, DEVICE, BOTH = "host", "device", "both"   # where the authoritative particle state is
_IMMUTABLE = (int, float, complex, np.generic)  # values later in-place arithmetic cannot change: no copy needed to keep them
_PLAIN_FLOATS = (float, np.float64, np.float32)


def _snap(x):
    """``x`` as it is now, safe from a later ``x += ...`` (physicl/__init__.py:343 deep-copies t into ts)."""
    return x if isinstance(x, _IMMUTABLE) else copy.deepcopy(x)


_MAX_PLANES = 12        # == _hip.MAX_PLANES == PCL_MAX_PLANES (include/physicl_hip.h); tests/test_host_api.py checks it


class Step:
    """Base class of every plugin (physicl/__init__.py:293-322)."""
    _device_native = False      # True: implements _device_run(sim) on the resident store
    _touches_objects = True     # False: only reads/writes scalars of the simulation (t, dt)
    _reads_only = False         # True: host step that never modifies objects (no re-upload needed)

    def __init__(self):
        pass

    def __compile_cl__(self, sim):
        pass

    def run(self, sim):
        pass

    def terminate(self, sim):
        pass


class DeviceStep(Step):
    """A step implemented as HIP kernels on the resident particle store (``_device_run``).  Called
    directly (``step.run(sim)``) it uploads the objects if need be and launches; inside
    ``Simulation.run`` the loop may fuse it with its neighbours."""
    _device_native = True

    def _device_run(self, sim):
        raise NotImplementedError

    def run(self, sim):
        sim._need_device(type(self).__name__)
        with sim._dev_lock:
            sim._to_device()
            self._device_run(sim)


class UpdateTimeStep(Step):
    """``dt = fn(sim); t += dt; ts.append(t)`` (physicl/__init__.py:324-343)."""
    _touches_objects = False

    def __init__(self, fn):
        super().__init__()
        self.fn = fn

    def run(self, sim):
        sim.dt = self.fn(sim)
        sim.t += sim.dt
        sim.ts.append(copy.deepcopy(sim.t))


class MeasureStep(Step):
    """Collects one row per step in ``self.data``; optional CSV at terminate
    (physicl/__init__.py:345-378)."""

    def __init__(self, out_fn=None):
        self.out_fn = out_fn
        self.data = []

    def terminate(self, sim):
        if self.out_fn is None:
            return
        rows = self.data.values() if isinstance(self.data, dict) else self.data
        with open(self.out_fn, "w") as f:
            for row in rows:
                f.write(", ".join(str(i) for i in list(row)) + "\n")


class Object:
    """Generic particle: r, dr, dv, v, a (3-vectors) + any keyword attributes
    (physicl/__init__.py:381-396)."""

    def __init__(self, **kwargs):
        self.r = Measurement([0] * 3, "m**1")
        self.dr = Measurement([0] * 3, "m**1")
        self.dv = Measurement([0] * 3, "m**1 s**-2")
        self.v = Measurement([0] * 3, "m**1 s**-1")
        self.a = Measurement([0] * 3, "m**1 s**-2")
        for attr, val in kwargs.items():
            setattr(self, attr, val)


class PhotonBatch:
    """``n`` photons described, not instantiated: r = 0, v = (c,0,0), E = min + (max-min) * U**(1/3)
    -- what ``light.generate_photons(n, min=, max=)`` makes one Python object at a time
    (physicl/light.py:112-128).  ``Simulation.add_objs(batch)`` creates them directly in device
    memory (1e8 photons take milliseconds instead of hours and ~100 GB of Python objects)."""

    def __init__(self, n, e_min, e_max, seed=0, table=None, fn_vec=None):
        self.n, self.e_min, self.e_max, self.seed = int(n), float(np.asarray(e_min)), float(np.asarray(e_max)), int(seed)
        self.table = table        # (cdf, grid): tabulated energy distribution instead of the power law
        self.fn_vec = fn_vec      # fn_vec(size) -> size numbers: the user's own sampler, evaluated on the host in chunks

    FN_CHUNK = 1 << 22

    def host_energies(self, lo, hi):
        """Chunks (offset in [lo, hi), energies) of ``e_min + (e_max - e_min) * fn_vec(...)`` for the photons [lo, hi) of the
        batch.  The sampler is called for EVERY photon of the batch, in order, whatever the shard: a sampler that walks a
        seeded stream (np.random) hands photon i the same number in a sharded run as in an unsharded one."""
        for at in range(0, self.n, self.FN_CHUNK):
            m = min(self.FN_CHUNK, self.n - at)
            u = np.asarray(self.fn_vec(m), dtype=np.float64).reshape(-1)
            if u.shape[0] != m:
                raise ValueError("fn_vec(%d) returned %d numbers" % (m, u.shape[0]))
            a, b = max(at, lo), min(at + m, hi)
            if b > a:
                yield a - lo, self.e_min + (self.e_max - self.e_min) * u[a - at:b - at]

    def __len__(self):
        return self.n


class ObjectList:
    """``sim.objects``: behaves like the reference's plain list (physicl/__init__.py:421)."""

    def __init__(self, sim, items=()):
        self._sim = sim
        self._items = list(items)

    # cheap: never moves data
    def __len__(self):
        return self._sim._object_count()

    def __bool__(self):
        return len(self) > 0

    # everything else needs real Python objects
    def _host(self, mutate):
        self._sim._to_host(mutate and not self._sim._readonly_scope)
        return self._items

    def __iter__(self):
        return iter(self._host(True))          # callers may modify the objects they iterate over

    def __getitem__(self, i):
        return self._host(True)[i]

    def __contains__(self, o):
        return o in self._host(False)

    def index(self, o):
        return self._host(False).index(o)

    def __setitem__(self, i, o):
        self._host(True)[i] = o

    def __delitem__(self, i):
        del self._host(True)[i]

    def append(self, o):
        self._host(True).append(o)

    def extend(self, it):
        self._host(True).extend(it)

    def insert(self, i, o):
        self._host(True).insert(i, o)

    def remove(self, o):
        self._host(True).remove(o)

    def pop(self, i=-1):
        return self._host(True).pop(i)

    def clear(self):
        self._host(True).clear()

    def __eq__(self, other):
        return list(self._host(False)) == list(other)

    def __repr__(self):
        if self._sim._residency == DEVICE:
            return "<ObjectList: %d particles resident on the HIP device>" % len(self)
        return repr(self._items)


class Simulation(threading.Thread):
    """Runs the steps, in the order they were added, until ``exit(sim)`` is true
    (physicl/__init__.py:400-541).

    Extra keyword attributes understood by this build (all optional):
      device   HIP device index (default 0, or LOCAL_RANK when ``comm`` is given)
      devices  a list of HIP device indices: the particles are sharded by index over that many GPUs INSIDE this
               process (one library context per entry, launches issued side by side from a thread pool, counters summed
               on the host: physicl_amd/multidev.py).  A script or a notebook scales over a node's GPUs this way without
               a launcher; every row, ``len(sim.objects)``, ``get_state()`` and the objects themselves are global, and
               the results are those of one device.  ``devices=[0, 0]`` makes two contexts on one GPU.
      rng      "numpy": the light steps draw their randoms from ``np.random`` on the host, in the
               reference's order (3 per photon per step: rtheta, rphi, rand; 1 for delete), so a
               seeded run reproduces the reference's OpenCL path;  "philox": drawn in-kernel,
               keyed by (seed, launch number, photon id) -- the fast mode.  Default "numpy",
               "philox" for simulations created from a PhotonBatch.
      seed     Philox seed
      fuse     True (default): consecutive Newton / ScatterIsotropic / counting-measure steps run
               as ONE kernel (bit-identical results)
      comm     a physicl_amd.dist.CounterComm (torch.distributed) or physicl_amd.comm.NativeCounterComm (the library's own
               RCCL entry points, no torch): this process owns one index shard of the particles;
               counters (alive, hits, measure rows) are all-reduced
      steps_per_launch
               How many passes of the loop one launch may carry.  Default (None): automatic -- when every pass is exactly
               [UpdateTimeStep] followed by one or two groups [NewtonianKinematicsStep][ScatterIsotropicStep |
               ScatterDeleteStep][counting measures], rng is "philox", and ``exit`` / the time-step function look at
               nothing but the clock (``t``, ``dt``, ``ts``) and the object count, up to 32 passes (64 in a loop whose only
               light step is a ScatterDeleteStep) run as ONE pass over the
               device store (photons do not interact; state, ``hits`` and every measure row are bit-identical to one
               launch per step).  The host side of those passes -- the time update and ``exit(sim)`` -- is evaluated
               ahead of the launch on a view of the simulation that exposes exactly those things; a function that
               touches anything else (measured rows, the objects, ``hits``), that reaches ``time``, ``random``,
               ``np.random``, ``os`` ... or keeps state of its own between calls (both functions are called exactly as
               often as the reference's loop calls them, but K of those calls come before the launch), or whose verdict
               depends on HOW MANY objects are left rather than on whether any are, silently gets one launch per light step instead
               (``sim.launch_note`` says why).  K > 1: the same with up to K passes.  1: always one launch per light step.
               For loops without a ScatterDeleteStep ``exit`` is evaluated ahead of the launch only; with one it is also
               replayed on the returned rows and the run is cut at the pass that emptied the store.

      rtc_background
               True (default): a ``variable_n_fn`` of one of the reference's example shapes does not wait for hipRTC
               (about 2 s the first time a text is seen on a machine): the run starts on ahead-of-time kernels and moves
               to the specialised ones when they are ready -- bit-identical results.  False: compile first.

    ``sim.schedule`` (a Counter) tells afterwards how the passes were launched: "fused", "fused_delete" (one launch per
    light step), "fused_multi", "fused_delete_multi", "mixed_multi" (K passes per launch).
    """

    def __init__(self, *args, **kwargs):
        threading.Thread.__init__(self)
        # older scripts: Simulation({"cl_on": ...}) or Simulation(params={...})  (examples/trace_ex.py:7, runtime1.py:21)
        if args and isinstance(args[0], dict):
            kwargs = dict(args[0], **kwargs)
        if isinstance(kwargs.get("params"), dict):
            kwargs = dict(kwargs.pop("params"), **kwargs)
        self.bounds = np.zeros(3)
        self.cl_on = True
        self.exit = lambda x: len(x.objects) == 0
        self.state_fn = lambda x: {"objects": len(x.objects), "t": x.t, "dt": x.dt,
                                   "run_time": time.time() - x.start_time}
        self.state_need_lock = False
        self.device = None
        self.devices = None
        self.rng = None
        self.seed = 0
        self.fuse = True
        self.comm = None
        self.steps_per_launch = None          # automatic (see the class docstring)
        for attr, val in kwargs.items():
            setattr(self, attr, val)
        self.dt = Measurement(np.double(0), "s**1")
        self.t = Measurement(np.double(0), "s**1")
        self._objects = ObjectList(self)
        self.steps = {}
        self._state_lock = threading.Lock()
        self.running = False
        self.ts = []                  # (the reference creates ``ts`` in run(), physicl/__init__.py:510: a script that polls it right after
                                      #  start() would race the simulation thread for the attribute)
        self.start_time = 0
        self.error = None
        # device-side state
        self._dev = None
        self._residency = HOST
        self._batch = None            # PhotonBatch not yet / already created on the device
        self._alive = 0               # host mirror of the (global) alive count while DEVICE-resident
        self._scattered = False       # a device scatter step has replaced velocities since the upload
        self._launch = 0              # Philox "step" word: one per light-step launch
        self._plan_key, self._plan = None, None
        self._upload_gen, self._multi_key, self._multi_ok = 0, None, False   # steps_per_launch eligibility cache
        self._ahead_ok = True         # exit / the time-step function can be evaluated ahead of a launch (until one cannot)
        self.launch_note = None       # why the run fell back to one launch per light step, if it did
        self._ahead_key = None
        self._count_key, self._count_why = None, None
        self._readonly_scope = False  # inside a host step that promises not to modify objects
        self._dev_lock = threading.RLock()   # one device call in flight per context (include/physicl_hip.h)
        self._uploaded, self._upload_lo = [], 0
        self._all_photons = True
        self.hits = 0                 # photons scattered by the most recent ScatterIsotropicStep
        self.schedule = collections.Counter()   # diagnostic: device launches by formulation ("fused", "fused_multi", ...)
        self._hip = None
        if self.cl_on:
            self._open_device()                     # raises if there is no GPU / no library: no fallback
        self.cl_ctx = self._dev                     # reference attribute names (physicl/__init__.py:427-432): None when
        self.cl_q = self._dev                       # cl_on is False

    # ------------------------------------------------------------------ objects
    @property
    def objects(self):
        return self._objects

    @objects.setter
    def objects(self, value):
        self._to_host(True)
        self._objects = ObjectList(self, value)

    def _object_count(self):
        if self._residency == DEVICE or self._residency == BOTH:
            return self._alive
        if self._batch is not None:
            return self._batch.n
        return len(self._objects._items)

    def add_obj(self, obj):
        if isinstance(obj, PhotonBatch):
            return self.add_objs(obj)
        if self._batch is not None:
            raise NotImplementedError("cannot mix explicit objects with a PhotonBatch")
        self._objects.append(obj)

    def add_objs(self, objs):
        if isinstance(objs, PhotonBatch):
            if self._batch is not None or self._objects._items:
                raise NotImplementedError("a PhotonBatch must be the simulation's only source of particles")
            self._batch = objs
            if self.rng is None:
                self.rng = "philox"
            return
        if self._batch is not None:
            raise NotImplementedError("cannot mix explicit objects with a PhotonBatch")
        self._objects.extend(objs)

    def remove_obj(self, obj):
        self._objects.remove(obj)

    # ------------------------------------------------------------------ steps
    def add_step(self, idx, step):
        if idx in self.steps:
            raise IndexError("Cannot add a step to an existing index.")
        self.steps[idx] = step

    def remove_step(self, idx):
        if self.running:
            raise RuntimeError("Cannot remove a Step while the simulation is running.")
        self.steps.pop(idx)

    # ------------------------------------------------------------------ device residency
    def _open_device(self):
        from . import _hip
        self._hip = _hip
        if self.devices is not None:
            if self.comm is not None:
                raise ValueError("devices=[...] shards inside this process; comm=... shards across processes: give one of them")
            from .multidev import MultiDevice
            self._dev = MultiDevice(self.devices, _hip)
        else:
            dev_index = self.device if self.device is not None else (self.comm.local_rank if self.comm else 0)
            self._dev = _hip.Device(dev_index)
            if hasattr(self.comm, "attach"):         # physicl_amd.comm.NativeCounterComm: its communicator lives on this context
                self.comm.attach(self._dev)
        # a variable_n_fn of one of the reference's example shapes starts at once on the ahead-of-time kernels while
        # hipRTC compiles its specialisation beside the run (~2 s; same bits, about the same speed)
        self._dev.set_rtc_background(bool(getattr(self, "rtc_background", True)))

    def _py_semantics(self):
        """``cl_on=False``: the light steps follow the reference's CPU paths (RNG order, write-back), on the device."""
        return self.cl_on == False      # noqa: E712 -- the reference's own test (light.py:207, 356)

    def _need_device(self, what):
        """The HIP device every device-native step runs on -- also under ``cl_on=False``, which only selects the
        reference's CPU-path semantics (there is no CPU implementation of the hot path to fall back to)."""
        if self._dev is None:
            self._open_device()
        return self._dev

    def _shard(self, n_global):
        return self.comm.shard(n_global) if self.comm is not None else (0, n_global)

    def _to_device(self):
        """Make the device store authoritative (upload if the Python objects are)."""
        with self._dev_lock:
            self._to_device_locked()

    def _to_device_locked(self):
        if self._residency in (DEVICE, BOTH):
            self._residency = DEVICE
            return
        dev = self._need_device("this step")
        self._upload_locked(dev)
        # a step list with a delete step will compact the store sooner or later: its second slab is allocated (and
        # chosen among candidates) now, with the upload, not inside the loop body that first needs it
        if self._residency == DEVICE and dev.capacity > 0 and \
                any(getattr(s, "_fuse_role", None) == "scatter_delete" for s in self.steps.values()):
            dev.reserve_compaction()

    def _upload_locked(self, dev):
        if self._batch is not None:
            b = self._batch
            lo, hi = self._shard(b.n)
            if dev.capacity < hi - lo or dev.capacity == 0:
                dev.store_alloc(max(hi - lo, 1))
            from .light import c as _c
            if b.table is not None:
                dev.fill_photons_table(hi - lo, lo, float(np.asarray(_c)), b.table[0], b.table[1], b.seed)
            else:
                dev.fill_photons(hi - lo, lo, float(np.asarray(_c)), b.e_min, b.e_max, b.seed)
                if b.fn_vec is not None:                  # the user's sampler: r, v, ids as filled, E from the host
                    for off, E in b.host_energies(lo, hi):
                        dev.upload(self._hip.E, E.astype(dev.np_dtype, copy=False), off)
            self._all_photons = True
            self._alive = b.n
            self._residency = DEVICE
            self._upload_gen += 1
            return
        items = self._objects._items
        lo, hi = self._shard(len(items))
        mine = items[lo:hi]
        n = len(mine)
        if dev.capacity < max(n, 1):
            dev.store_alloc(max(n, 1))
        from .light import PhotonObject
        cols = {}
        for g in ("r", "v", "dr", "dv"):
            try:                                 # one conversion per field: ~2x the per-object loop below
                cols[g] = np.asarray([getattr(o, g) for o in mine], dtype=np.float64).reshape(n, 3)
            except (ValueError, TypeError):      # ragged or odd-shaped attributes: object by object
                cols[g] = np.empty((n, 3))
                for k, o in enumerate(mine):
                    cols[g][k] = np.asarray(getattr(o, g), dtype=np.float64).reshape(3)
        # exact type, as the reference's ``type(obj) != PhotonObject`` (physicl/light.py:233, 283)
        kind = np.fromiter((type(o) is PhotonObject for o in mine), dtype=np.uint8, count=n)
        E = np.ones(n)
        if n:
            ph = np.flatnonzero(kind)
            E[ph] = [float(np.asarray(mine[k].E)) for k in ph]
        state = dict(cols, E=E, id_base=lo)
        self._all_photons = bool(kind.all())
        if not self._all_photons:
            state["kind"] = kind
        dev.upload_state(state)
        self._uploaded = mine                     # shells, index == device id - lo
        self._upload_lo = lo
        self._alive = len(items)
        self._scattered = False
        self._residency = DEVICE
        self._upload_gen += 1

    def _to_host(self, mutate=True):
        """Bring the state back into Python objects.  mutate=False keeps the device copy valid."""
        with self._dev_lock:
            self._to_host_locked(mutate)

    def _to_host_locked(self, mutate):
        if self._residency == HOST:
            return
        if self.comm is not None and self.comm.world > 1:
            raise NotImplementedError("per-object access to a sharded simulation: use the counters / "
                                      "Simulation.download() on each rank, or run on a single GPU")
        if self._residency == BOTH:
            if mutate:
                self._residency = HOST
            return
        dev = self._dev
        if self._batch is not None:
            self._materialise_batch()
        s = dev.download_state()
        shells = self._uploaded
        if isinstance(shells, dict):                          # materialised batch: keyed by photon id
            ids = s["id"]
            keep = [shells[int(i)] for i in ids]
        else:                                                 # uploaded objects: index == device id - lo
            ids = s["id"] - self._upload_lo
            keep = [shells[i] for i in ids]
        r, v, dr, dv = (np.stack(s[g], 1) if len(ids) else np.zeros((0, 3)) for g in ("r", "v", "dr", "dv"))
        for k, o in enumerate(keep):
            o.r = Measurement._from_code(r[k], like=o.r, units="m**1")
            o.dr = Measurement._from_code(dr[k], units="m**1")
            if self._scattered:
                o.v = np.array(v[k], dtype=np.double)         # the reference leaves a plain ndarray (light.py:328)
                o.dv = np.array(dv[k], dtype=np.double)
            else:
                o.v = Measurement._from_code(v[k], like=o.v, units="m**1 s**-1") if isinstance(o.v, Measurement) \
                    else np.array(v[k], dtype=np.double)
        self._objects._items[:] = keep
        # the device keeps its ids (a photon's random stream is keyed by them: looking at the objects must not change
        # the run) and ``_uploaded`` keeps every shell ever uploaded, removed ones included
        self._residency = HOST if mutate else BOTH

    def _materialise_batch(self):
        """Turn a device-created PhotonBatch into real PhotonObjects (only sensible for small n)."""
        n = self._dev.count
        if n > 5_000_000:
            raise MemoryError("refusing to create %d Python PhotonObjects; use the counters / download arrays" % n)
        from .light import PhotonObject, c as _c
        cval = np.asarray(_c)
        ids = self._dev.download_ids()
        E = self._dev.download(self._hip.E)
        shells = [PhotonObject.__new__(PhotonObject) for _ in range(n)]
        for o, e, i in zip(shells, E, ids):
            Object.__init__(o, E=np.double(e), v=Measurement._from_code([cval, 0, 0], units="m**1 s**-1"), uid=int(i))
        self._uploaded = {int(i): o for i, o in zip(ids, shells)}    # the photons keep their ids (and random streams)
        self._upload_lo = 0
        self._batch = None

    # ------------------------------------------------------------------ helpers for the device-native steps
    def _dt_code(self):
        return float(np.asarray(self.dt))

    def _next_launch(self):
        self._launch += 1
        return self._launch

    def _host_randoms(self, which):
        """Upload this step's host-drawn randoms (rng == 'numpy').  Reference order: per photon
        rtheta, rphi, rand (physicl/light.py:285, physicl/__init__.py:606-619); delete draws one."""
        dev = self._dev
        n = dev.count
        if self._all_photons:                      # the common case: no kind array to consult
            ph, m = slice(None), n
        else:
            ph = dev.download_kind(n) != 0 if n else np.zeros(0, bool)
            m = int(ph.sum())
        if which == "iso" and self._all_photons and os.environ.get("PCL_RAND3", "1") != "0":   # ("0": the three-array path, for A/B)
            # the raw uniforms go over in chunks, as they are drawn (consecutive draws are the one big draw's stream);
            # the copy of a chunk runs while the next one is drawn, and the split / scaling happens on the device
            ch = dev.RAND3_CHUNK
            for off in range(0, n, ch):
                dev.upload_rand3(np.random.random((min(ch, n - off), 3)), off)
        elif which == "iso":
            u = np.random.random((m, 3))
            full = np.zeros((n, 3))
            full[ph, 0] = u[:, 0] * 2 * np.pi
            full[ph, 1] = u[:, 1] * np.pi
            full[ph, 2] = u[:, 2]
            for w in range(3):
                dev.upload_rand(w, np.ascontiguousarray(full[:, w]))
        elif self._all_photons:
            dev.upload_rand(2, np.random.random(n))
        else:
            full = np.zeros(n)
            full[ph] = np.random.random(m)
            dev.upload_rand(2, full)

    def _rng_mode(self):
        if (self.rng or "numpy") == "philox" and not self._py_semantics():
            return self._hip.RNG_PHILOX
        if self.comm is not None and self.comm.world > 1:
            # every rank would draw the SAME np.random stream for its own shard (scripts seed it once): correlated
            # shards, and never the unsharded run's numbers.  Only the id-keyed device RNG is shard-independent.
            raise ValueError("a sharded Simulation (comm.world = %d) needs rng='philox'; rng=%r draws host randoms per "
                             "rank" % (self.comm.world, self.rng or "numpy"))
        return self._hip.RNG_INPUT

    def _global(self, values):
        return self.comm.allreduce_sum(values) if self.comm is not None else np.asarray(values, dtype=np.int64)

    # ------------------------------------------------------------------ the loop
    def _build_plan(self):
        """Group consecutive fusable device-native steps:
        [Newton][ScatterIsotropic | ScatterDelete]?[counting measures]*  -> one kernel (pipeline) per pass."""
        steps = list(self.steps.values())
        plan, i = [], 0
        while i < len(steps):
            s = steps[i]
            # (cl_on=False: the CPU paths' randoms depend on every photon's own collision probability, which the host has
            #  to see between the move and the scatter -- nothing is fused)
            if self.fuse and not self._py_semantics() and getattr(s, "_fuse_role", None) == "newton":
                group, j = [s], i + 1
                if j < len(steps) and getattr(steps[j], "_fuse_role", None) in ("scatter_iso", "scatter_delete"):
                    group.append(steps[j])
                    j += 1
                n_planes = 0
                while j < len(steps) and getattr(steps[j], "_fuse_role", None) in ("measure", "trace") and \
                        n_planes + steps[j]._n_planes() <= _MAX_PLANES:
                    n_planes += steps[j]._n_planes()
                    group.append(steps[j])
                    j += 1
                if len(group) > 1:
                    plan.append(("fused", group))
                    i = j
                    continue
            plan.append(("single", s))
            i += 1
        return plan

    def _run_pass(self):
        with self._dev_lock:
            self._run_pass_locked()

    def _run_pass_locked(self):
        key = tuple(id(s) for s in self.steps.values()) + (self.fuse,)
        if key != self._plan_key:
            self._plan_key, self._plan = key, self._build_plan()
        if self._k_wanted() > 1 and self._ahead_ok and not self._py_semantics() and self._multi_agreed() and \
                self._ahead_agreed(self._plan[0][1]):
            if self._run_multi(self._plan[0][1], [item for _, item in self._plan[1:]]):
                return
        for kind, item in self._plan:
            if kind == "fused":
                self._run_fused(item)
            elif item._device_native:
                self._to_device()
                item._device_run(self)
            else:
                # host plugin: sim.objects brings the state back into the Python objects on first touch
                self._readonly_scope = bool(item._reads_only)
                try:
                    item.run(self)
                finally:
                    self._readonly_scope = False

    def _multi_agreed(self):
        """_multi_eligible(), decided once per (plan, upload) and -- with ``comm`` -- agreed by all ranks: every rank
        must follow the same launch schedule, or the collectives of different schedules would wait for each other."""
        key = (self._plan_key, self._upload_gen)
        if key != self._multi_key:
            ok = 1 if self._multi_eligible() else 0
            if self.comm is not None:
                ok = -int(self.comm.allreduce_sum([-ok])[0]) == self.comm.world      # all ranks eligible
            self._multi_key, self._multi_ok = (self._plan_key, self._upload_gen), bool(ok)
        return self._multi_ok

    def _multi_eligible(self):
        """The whole pass is [UpdateTimeStep] followed by one or two groups [Newton + light step + counting measures]
        -- at most one ScatterIsotropicStep and one ScatterDeleteStep, either order -- with the device RNG: K passes can
        run as one launch (pcl_step_fused_multi / pcl_step_fused_delete_multi / pcl_step_mixed_multi)."""
        plan = self._plan
        if not (2 <= len(plan) <= 3) or plan[0][0] != "single" or type(plan[0][1]) is not UpdateTimeStep:
            return False
        seen = []
        for kind, group in plan[1:]:
            if kind != "fused":
                return False
            roles = [s._fuse_role for s in group]
            if roles[0] != "newton" or len(roles) < 2 or roles[1] not in ("scatter_iso", "scatter_delete") or \
                    any(r not in ("measure", "trace") for r in roles[2:]) or roles[1] in seen:
                return False
            seen.append(roles[1])
        if sum(m._n_planes() for _, group in plan[1:] for m in group[2:]) > self._hip.MAX_PLANES:
            return False
        if self._rng_mode() != self._hip.RNG_PHILOX:
            return False
        # a TracePathMeasureStep in the loop: its tracked subset is worked out on the device ahead of every launch
        # (pcl_store_trace_ahead) -- or, if it cannot be (it asks for more particles than that takes), the loop runs one
        # launch per light step with the step as a host plugin
        tracers = [m for _, group in plan[1:] for m in group[2:] if m._fuse_role == "trace"]
        if tracers:
            self._to_device()
            if any(m._ahead_set(self) is None for m in tracers):
                return False
        return True

    def _k_wanted(self):
        k = self.steps_per_launch
        return 32 if k is None else max(1, int(k))

    def _ahead_agreed(self, upd):
        """Before the first K-pass launch (and again when the functions or the steps change): ``exit`` and the time-step
        function must be plain functions that reach the run only through their argument (physicl_amd/ahead.py)."""
        key = (id(self.exit), id(upd.fn), self._plan_key)
        if key != self._ahead_key:
            self._ahead_key = key
            self._ahead_ok, self.launch_note = True, None
            steps = list(self.steps.values())
            for what, fn in (("exit", self.exit), ("the time-step function", upd.fn)):
                ok, why = _clock_only(fn, steps)
                if not ok:
                    self._ahead_ok = False
                    self.launch_note = "one launch per light step: %s %s" % (what, why)
                    break
        return self._ahead_ok

    def _ahead_off(self, what, name):
        self._ahead_ok = False
        self.launch_note = "one launch per light step: %s %s, which is not known ahead of a launch" % (what, name)

    def _count_known_ahead(self, upd):
        """Loops with a ScatterDeleteStep: the object count a later pass of the launch will see is not known ahead of it --
        only that a store that is not empty now stays "not empty" until the returned rows say otherwise.  ``exit`` and the
        time-step function may therefore ask ``sim.objects`` whether it is empty and nothing else; the use is read off their
        bytecode (ahead.count_use), once per pair of functions, before anything has advanced."""
        key = (id(self.exit), id(upd.fn))
        if key != self._count_key:
            self._count_key = key
            self._count_why = None
            for what, fn in (("exit(sim)", self.exit), ("the time-step function", upd.fn)):
                if _count_use(fn) == "other":
                    self._count_why = what
                    break
        if self._count_why is not None:
            self._ahead_off(self._count_why + " depends on", "how many objects are left")
            return False
        return True

    def _plan_passes(self, upd, k_max, count_matters):
        """Host part of up to ``k_max`` passes, ahead of the launch: the time update of each pass, then the exit test
        the outer loop would make before the next one -- both on an _AheadView of the simulation, each ONCE per pass as
        in the reference's loop (physicl/__init__.py:512-516).  Returns [(t, dt)] per pass and the code dt.  Planning
        stops early (and for good: ``_ahead_ok``) as soon as one of the two functions looks at something the view does
        not have; the passes planned so far still run as one launch, and the outer loop makes its next exit test on the
        real simulation.  ``count_matters`` (a ScatterDeleteStep is in the loop): nothing is planned unless both
        functions use the object count for emptiness only (_count_known_ahead)."""
        times, dt0 = [], None
        if count_matters and not self._count_known_ahead(upd):
            return times, dt0
        view = _AheadView(self, self._alive)
        fn, exit_fn, ts = upd.fn, self.exit, self.ts
        while len(times) < k_max:
            t_before, dt_before = _snap(self.t), self.dt   # ``t += dt`` is in place on an ndarray
            try:
                dt = fn(view)
            except _NotAhead as e:
                self._ahead_off("the time-step function reads", "sim." + str(e.args[0]))
                break
            self.dt = dt                              # UpdateTimeStep.run (physicl/__init__.py:337-343)
            self.t += dt
            ts.append(_snap(self.t))                  # (its own copy: a Measurement clock is advanced IN PLACE by the next pass)
            code = float(dt) if type(dt) in _PLAIN_FLOATS else self._dt_code()
            if dt0 is None:
                dt0 = code
            elif code != dt0:                         # the time step changed: that pass belongs to the next launch
                self.t, self.dt = t_before, dt_before
                ts.pop()
                break
            times.append((_snap(self.t), dt))         # ... and another one for the row replay, which re-installs it as sim.t
            if len(times) < k_max:
                try:
                    view.refresh()
                    stop = exit_fn(view)
                except _NotAhead as e:
                    self._ahead_off("exit(sim) reads", "sim." + str(e.args[0]))
                    break
                if stop:
                    break
        return times, dt0

    def _run_multi(self, upd, groups):
        """Up to ``steps_per_launch`` passes of the loop in one launch.  The host part of each pass runs first
        (_plan_passes); the device then advances the particles through all of them in one pass over the store (and one
        compaction if a ScatterDeleteStep is in the loop) and returns one counter row per light step per pass.
        ``exit`` usually waits for the store to empty, which the host cannot know ahead of the launch: the rows are
        replayed afterwards and, from the pass that left nothing alive on, the exit test is made again on the real
        simulation; the run is cut at the first pass where it is true (while photons are alive the planned verdicts hold:
        the functions ask the object list for emptiness only, _count_known_ahead)."""
        self._to_device()             # a host plugin or another thread may have taken the objects back since the last pass
        dev, hip = self._dev, self._hip
        P = len(groups)
        lights = [g[1] for g in groups]
        phases = ["iso" if s._fuse_role == "scatter_iso" else "delete" for s in lights]
        has_delete = "delete" in phases
        n_ts = len(self.ts)
        # PCL_MULTI_MAX = 64 rows per launch.  A delete-only loop takes all 64 when the pass count is automatic: the first
        # launch of such a run costs the same whatever it carries (0.37 ms at 1e7 photons with 16, 32 or 64 bodies -- the
        # photons die off inside it), so the 49-53 passes of test/test_light.py:52-59's run are ONE launch instead of two with
        # a replay and a planning round in between: 0.63 -> 0.54 ms (tools/delsim_split.py; round 3 measured the opposite,
        # when planning and replay cost three times as much)
        k_max = self._k_wanted()
        if self.steps_per_launch is None and phases == ["delete"]:
            k_max = 64                                # (automatic, delete-only loop: see above)
        times, dt0 = self._plan_passes(upd, max(1, min(k_max, 64 // P)), has_delete)
        k = len(times)
        if k == 0:
            return False                              # nothing could be planned ahead: this pass runs the plain way
        planes, span = [], []
        for g in groups:
            pl = [p for m in g[2:] for p in m._plane_rows()]
            span.append((len(planes), len(pl)))
            planes += pl
        sc = dl = None
        for s in lights:
            if s._fuse_role == "scatter_iso":
                sc = s._kernel_params(self)
            else:
                dl = s._kernel_consts()
        step0 = self._launch + 1
        self._launch += k * P
        # TracePathMeasureStep: where its tracked particles will be behind its group's light step in each of the k passes --
        # worked out from the store as it stands, BEFORE the launch that moves it (same constants, same launch indices)
        traced = []
        tracers = [(j, m) for j, g in enumerate(groups) for m in g[2:] if m._fuse_role == "trace"]
        for j, m in tracers:
            # (one tracer: its kernel is only enqueued, the rows -- pinned host memory of the context -- are read behind the launch;
            #  several share that buffer and are read one by one)
            got = dev.trace_ahead(m._ahead_set(self), dt0, k, phases, j, sc, dl, self.seed, step0, defer=len(tracers) == 1)
            traced.append((m, got if len(tracers) == 1 else (lambda rows=got: rows)))
        # raw rows from the library: one per light step per pass, columns [N, sign x 3, planes ..., hits | removed]
        if phases == ["iso"] and dev.is_uniform():
            sc.update(rng_mode=hip.RNG_PHILOX, seed=self.seed, step=step0)
            raw = dev.step_fused_multi(dt0, k, sc, planes, raw=True)
            self.schedule["fused_multi"] += 1
        elif phases == ["delete"]:
            raw = dev.step_fused_delete_multi(dt0, k, dl[0], dl[1], self.seed, step0, planes if groups[0][2:] else None, raw=True)
            self.schedule["fused_delete_multi"] += 1
        else:
            raw = dev.step_mixed_multi(dt0, k, phases, sc, dl, planes, self.seed, step0, raw=True)
            self.schedule["mixed_multi"] += 1
        traced = [(m, read()) for m, read in traced]   # (the rows were written while the launch ran: nothing to wait for)
        npl = len(planes)
        have = raw.shape[1] - 5                       # plane columns the library returned (0 when no measure step asked)
        flat = np.zeros((k * P, 5 + npl), dtype=np.int64)     # [N, event count, sign x 3, planes ...]: what is all-reduced
        flat[:, 0] = raw[:, 0]
        flat[:, 1] = raw[:, 4 + have]
        flat[:, 2:5] = raw[:, 1:4]
        flat[:, 5:5 + have] = raw[:, 4:4 + have]
        glob = self._global(flat.reshape(-1)).reshape(k * P, 5 + npl) if self.comm is not None else flat
        ts = self.ts
        # How many of the k passes does the loop keep?  Without a delete step all of them (the exit tests were planned).  With
        # one, the planned tests were made for a store that is not empty (the functions ask for emptiness only:
        # _count_known_ahead); a pass that empties it gets its test -- the one the outer loop would make there -- again, on
        # the simulation as it stood after that pass.  The later passes of the launch ran on an empty store: if the loop
        # stops, their times are dropped.
        done = [0]

        def flush(upto):
            """Passes [done, upto) become history: the state the last of them leaves behind, and their rows (each row carries
            its own pass's time)."""
            lo = done[0]
            if upto <= lo:
                return
            self.t, self.dt = times[upto - 1]
            for j, g in enumerate(groups):
                last = glob[(upto - 1) * P + j]
                if phases[j] == "iso":
                    self.hits = int(last[1])
                    self._scattered = True
                else:
                    self._alive, lights[j].removed = int(last[0]), int(last[1])
                at = 5 + span[j][0]
                for m in g[2:]:
                    if m._fuse_role == "trace":
                        continue
                    n_m = m._n_planes()
                    m._record_rows(self, [times[i][0] for i in range(lo, upto)], glob[lo * P + j:upto * P:P, 0],
                                   glob[lo * P + j:upto * P:P, 2:5], glob[lo * P + j:upto * P:P, at:at + n_m])
                    at += n_m
            for m, rows in traced:
                m._ahead_record(self, [times[i][0] for i in range(lo, upto)], rows[lo:upto])
            done[0] = upto

        keep = k
        if has_delete:
            jd = phases.index("delete")
            for i in np.flatnonzero(glob[jd:(k - 1) * P:P, 0] == 0).tolist():
                flush(i + 1)
                later = ts[n_ts + i + 1:]
                del ts[n_ts + i + 1:]
                if self.exit(self):
                    keep = i + 1
                    break
                ts.extend(later)
        flush(keep)
        return True

    def _run_fused(self, group):
        self._to_device()
        dev, hip = self._dev, self._hip
        scatter = next((s for s in group if s._fuse_role == "scatter_iso"), None)
        delete = next((s for s in group if s._fuse_role == "scatter_delete"), None)
        measures = [s for s in group if s._fuse_role == "measure"]
        tracers = [s for s in group if s._fuse_role == "trace"]
        planes = [p for m in measures for p in m._plane_rows()]

        def trace_ahead(phase, sc_, dl_, step):
            """The tracers of this group, before the launch: on the device where that works (a light step in the group, device
            RNG), the others run as host plugins behind it."""
            ahead = []
            for m in tracers:
                ids = m._ahead_set(self) if phase is not None and self._rng_mode() == hip.RNG_PHILOX else None
                ahead.append((m, None if ids is None else dev.trace_ahead(ids, self._dt_code(), 1, [phase], 0, sc_, dl_, self.seed, step)))
            return ahead

        def trace_file(ahead):
            for m, rows in ahead:
                if rows is not None:
                    m._ahead_record(self, [self.t], rows)
                else:
                    self._readonly_scope = True
                    try:
                        m.run(self)
                    finally:
                        self._readonly_scope = False

        if delete is not None:
            mode = self._rng_mode()
            if mode == hip.RNG_INPUT:
                self._host_randoms("delete")
            A_k, n_k = delete._kernel_consts()
            step = self._next_launch()
            ahead = trace_ahead("delete", None, (A_k, n_k), step)
            out = dev.step_fused_delete(self._dt_code(), A_k, n_k, mode, self.seed, step,
                                        planes if measures else None, lazy=True)
            self.schedule["fused_delete"] += 1
            g = self._global(np.concatenate([[out["N"], out["removed"]], out["sign"], out["planes"]]))
            self._alive, delete.removed = int(g[0]), int(g[1])
            k = 5
            for m in measures:
                npl = m._n_planes()
                m._record(self, int(g[0]), g[2:5], g[k:k + npl])
                k += npl
            trace_file(ahead)
            return
        sc = None
        ahead = []
        if scatter is not None:
            sc = scatter._kernel_params(self)
            sc.update(rng_mode=self._rng_mode(), seed=self.seed, step=self._next_launch())
            if sc["rng_mode"] == hip.RNG_INPUT:
                self._host_randoms("iso")
            self._scattered = True
            ahead = trace_ahead("iso", sc, None, sc["step"])
        else:
            ahead = trace_ahead(None, None, None, 0)
        # dr/dv stay implicit unless something after this pass looks at them (the store materialises on demand)
        out = dev.step_fused(self._dt_code(), sc, planes if (measures or scatter) else None, sync=True, lazy=True)
        self.schedule["fused"] += 1
        if out is not None:
            if scatter is not None:
                self.hits = int(self._global([out["hits"]])[0])
            glob = self._global(np.concatenate([[out["N"]], out["sign"], out["planes"]]))
            k = 4
            for m in measures:
                npl = m._n_planes()
                m._record(self, int(glob[0]), glob[1:4], glob[k:k + npl])
                k += npl
        trace_file(ahead)

    def run(self):
        # HIP's current device is per thread and a thread's first HIP call sets its runtime state up (tenths of a millisecond):
        # the simulation thread makes that call here, before the run's clock starts -- the reference creates its OpenCL context
        # and queue before the thread exists (physicl/__init__.py:427-429) and pays nothing comparable inside run()
        if self._dev is not None and os.environ.get("PCL_SIM_PREBIND", "1") != "0":
            try:
                self._dev.sync()
            except Exception:                   # noqa: BLE001 -- the first real call will report it
                pass
        self.start_time = time.time()
        self.t = 0
        self.dt = 0
        self.ts = []
        self.running = True
        try:
            while not self.exit(self):
                with self._state_lock:
                    self._run_pass()
            with self._state_lock:
                for step in self.steps.values():
                    step.terminate(self)
                self.run_time = time.time() - self.start_time
        except BaseException as e:          # a dead simulation thread must not look like a running one
            self.error = e
            raise
        finally:
            self.running = False

    def prepare(self):
        """Create / upload the particles and bind the device now instead of in the first pass of ``run`` (no counterpart in
        the reference, whose objects never leave the host): a script that times ``run_time`` then times stepping, not the
        allocation of the store."""
        self._need_device("prepare")
        self._to_device()
        self._dev.sync()

    def get_state(self):
        if self.state_need_lock:
            with self._state_lock:
                return self.state_fn(self)
        return self.state_fn(self)

    # ------------------------------------------------------------------ device introspection
    @staticmethod
    def get_device_info():
        """{device name: properties} for every visible HIP device (the reference dumps OpenCL
        platform/device info here, physicl/__init__.py:470-499)."""
        from . import _hip
        out = {}
        for i in range(_hip.device_count()):
            with _hip.Device(i) as d:
                info = d.info()
                out["%d: %s" % (i, info["name"])] = info
        return out

    @staticmethod
    def set_dev(id):
        """The reference's stub (physicl/__init__.py:526-529); pass ``device=`` to Simulation instead."""

    def download(self, field):
        """(n, 3) or (n,) array of a state field ('r','v','dr','dv','E','id') straight from the device --
        the way to look at 1e8 photons without creating Python objects."""
        self._to_device()
        d = self._dev
        if field == "E":
            return d.download(self._hip.E)
        if field == "id":
            return d.download_ids()
        return np.stack([d.download(f) for f in self._hip.FIELD_GROUPS[field]], 1)

    def close(self, download=True):
        """Free the device store and context (also happens at garbage collection).  ``download=False`` drops the
        device state instead of bringing it back into the Python objects first."""
        if self._dev is not None:
            if download and self._residency == DEVICE and self._batch is None:
                self._to_host(True)
            self._dev.close()
            self._dev = None


# ----------------------------------------------------------------------------------------------
# kernel-glue classes of the reference (physicl/__init__.py:543-664), kept for source compatibility
# ----------------------------------------------------------------------------------------------
class CLInput:
    types = ["obj", "obj_def", "obj_action", "const", "other"]

    def __init__(self, **kw):
        self.name, self.type = kw["name"], kw["type"]
        self.ctype = kw.get("ctype", "double")
        self.obj_attr, self.obj_def, self.obj_track = kw.get("obj_attr"), kw.get("obj_def"), kw.get("obj_track")
        self.code = kw.get("code")
        self.const_value = kw.get("const_value")


class CLOutput:
    def __init__(self, **kw):
        self.name = kw["name"]
        self.ctype = kw.get("ctype", "double")


class CLProgram:
    """User-defined kernels with the reference's glue API (physicl/__init__.py:567-664): describe the
    inputs with ``CLInput`` (per-object attributes, per-object expressions, constants, tracked objects),
    the outputs with ``CLOutput``, give the kernel BODY in the OpenCL-C dialect of the reference's kernels;
    ``build_kernel()`` compiles it (hipRTC instead of an OpenCL driver) and ``run()`` gathers the inputs
    from ``sim.objects``, launches one work-item per gathered object and returns ``{output name: ndarray}``.

    This is the reference's slow path by design (a Python gather per call); the steps shipped with this
    build do not use it.  Same contract as the reference: every array input is marshalled as float64
    whatever its ``ctype`` (physicl/__init__.py:613); ``int`` outputs are 32-bit; the global size is the
    length of the first ``"obj"`` input; ``obj_track`` lists (e.g. ``pht``) are left on the program
    object for the caller.
    """

    def __init__(self, sim, name, kernel_code):
        self.sim, self.prog_name, self.kernel_code = sim, name, kernel_code
        self.prep_metadata, self.output_metadata, self.variables, self.prog = [], [], {}, None

    def _signature(self):
        params = []
        for item in self.prep_metadata:
            if item.type in ("obj", "obj_def"):
                params.append((item.ctype, item.name, True))
            elif item.type == "const":
                params.append((item.ctype, item.name, False))
        params.extend((o.ctype, o.name, True) for o in self.output_metadata)
        return params

    def build_kernel(self):
        dev = self.sim._need_device("CLProgram")
        self.prog = dev.user_kernel(self.prog_name, self._signature(), self.kernel_code)

    def _gather(self):
        """One pass over sim.objects.  The per-object statements run in metadata order inside a single
        ``for obj in ...`` loop (an ``obj_action`` may ``continue`` to skip an object), exactly the execution
        model scripts written against the reference rely on."""
        lists = [it.name for it in self.prep_metadata if it.type in ("obj", "obj_def", "obj_track")]
        body = []
        for it in self.prep_metadata:
            if it.type == "obj":
                body.append("%s.append(obj.%s)" % (it.name, it.obj_attr))
            elif it.type == "obj_def":
                body.append("%s.append(%s)" % (it.name, it.obj_def))
            elif it.type == "obj_track":
                body.append("%s.append(%s)" % (it.name, it.obj_track))
            elif it.type == "obj_action":
                body.append(it.code)
        src = "for obj in _objects:\n\t" + "\n\t".join(body) if body else ""
        import types
        from . import light, newton
        ns = {"np": np, "self": self, "_objects": list(self.sim.objects),
              "physicl": types.SimpleNamespace(light=light, newton=newton, **{k: globals()[k] for k in (
                  "Measurement", "Object", "Step", "Simulation")})}
        ns["phys"] = ns["physicl"]
        for nm in lists:
            ns[nm] = []
        exec(src, ns)                          # noqa: S102 -- user-supplied gather code, as in the reference
        return {nm: ns[nm] for nm in lists}

    def run(self):
        if self.prog is None:
            self.build_kernel()
        dev = self.sim._dev
        with self.sim._dev_lock:
            got = self._gather()
            for it in self.prep_metadata:
                if it.type == "obj_track":
                    setattr(self, it.name, got[it.name])
            n, args, temps = None, [], []
            for it in self.prep_metadata:
                if it.type in ("obj", "obj_def"):
                    arr = np.array(got[it.name], dtype=np.double)
                    setattr(self, it.name + "_np", arr)
                    if n is None and it.type == "obj":
                        n = arr.shape[0]
                    d = dev.array(arr)
                    temps.append(d)
                    args.append(d)
                elif it.type == "const":
                    args.append(float(np.double(it.const_value)))
            if n is None:
                raise ValueError("CLProgram needs at least one input of type 'obj' to define the global size")
            outs = []
            for o in self.output_metadata:
                d = dev.empty(max(n, 1), {"int": np.int32, "double": np.float64, "float": np.float32}[o.ctype])
                temps.append(d)
                outs.append(d)
                args.append(d)
            try:
                self.prog(n, *args)
                return {o.name: d.get()[:n] for o, d in zip(self.output_metadata, outs)}
            finally:
                for d in temps:
                    d.free()
