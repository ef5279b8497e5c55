"""physicl.light equivalent: photons, scatter / delete steps and the counting measure steps, all on
the device store.  Public names follow the reference module (physicl/light.py); the OpenCL kernels
it builds at run time are replaced by the hand-written HIP kernels of libphysicl_hip.so."""
import collections.abc
import copy

import numpy as np
import numpy.linalg as np_lin

from .core import DeviceStep, MeasureStep, Object, PhotonBatch, Step
from .units import Measurement

# SI-defined constants (physicl/light.py:14-16); Measurements, so they follow the code scale in force at import
c = Measurement(np.double(299792458), "m**1 s**-1")
h = Measurement(np.double(6.62607015e-34), "J**1 s**1")
kB = Measurement(np.double(1.380649e-23), "J**1 K**-1")


class PhotonObject(Object):
    """Photon: needs an energy ``E`` and ``|v| == |c|`` (physicl/light.py:18-35)."""

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
        if np_lin.norm(self.v) != np_lin.norm(c):
            raise Exception("Not a valid speed.")
        if "E" not in kwargs:
            raise Exception("Needs a valid energy.")


def E_from_wavelength(wavelength):
    return (h * c) / wavelength


def wavelength_from_E(E):
    return (h * c) / E


# ---------------------------------------------------------------------------------------------- Planck sampling
def planck_distribution(E, T):
    """Normalised Planck photon-energy density, J**-1 (physicl/light.py:53-60).  Set-up time only."""
    E_ = E.__unscaled__() if isinstance(E, Measurement) else E
    T_ = T.__unscaled__() if isinstance(T, Measurement) else T
    k_ = kB.__unscaled__()
    # operation order of the reference (bit-exact): 15/(pi^4 kB T) * (E/(kB T))^3 * 1/e^(E/(kB T))
    norm = 15 / (np.pi ** 4 * k_ * T_)
    x = E_ / (k_ * T_)
    return Measurement(norm * (x ** 3) * (1 / (np.e ** (E_ / (k_ * T_)))), "J**-1")


def planck_probability(E_min, E_max, T, integrator=None):
    import scipy.integrate
    integrator = integrator or (lambda fn, a, b: scipy.integrate.quad(fn, a, b))
    return integrator(lambda x: planck_distribution(x, T), E_min, E_max)


_planck_cache = {"key": None, "cdf": None}


def planck_phot_distribution(E_min, E_max, T, bins=1000):
    """One photon energy drawn from the binned Planck CDF (physicl/light.py:73-104)."""
    key = [float(np.asarray(x.__unscaled__() if isinstance(x, Measurement) else x)) for x in (E_min, E_max, T, bins)]
    lo, hi, T_, nb = key
    grid = np.linspace(lo, hi, int(nb))
    if _planck_cache["key"] != key:
        area = [planck_probability(grid[k], grid[k + 1], T_)[0] for k in range(len(grid) - 1)]
        # left-to-right total and running sum, as the reference adds them (light.py:88-94; np.sum would pair the terms up):
        # the table is the reference's bit for bit (tests/golden g7_setup), so a draw picks the same bin
        _planck_cache["key"], _planck_cache["cdf"] = key, np.cumsum(np.array(area) / sum(area))
    cdf = _planck_cache["cdf"]
    u = np.random.rand()
    for k in range(1, len(cdf)):
        if cdf[k] >= u >= cdf[k - 1]:
            return Measurement(grid[k], "J**1")


def generate_photons_from_E(E):
    return [PhotonObject(E=x, v=c * [1, 0, 0]) for x in E]


def generate_photons(n, fn=lambda: np.random.power(3), min=0, max=0, bins=-1, dist=None):
    """``n`` PhotonObjects moving along +x with ``E = min + (max - min) * fn()`` (physicl/light.py:112-128).
    ``bins``/``dist`` are accepted and ignored like in the older scripts (examples/runtime1.py:67).
    For large n use ``generate_photons_bulk``."""
    return [PhotonObject(E=min + (max - min) * fn(), v=Measurement([c, 0, 0], "m**1 s**-1")) for _ in range(int(n))]


def generate_photons_bulk(n, min=0, max=0, seed=0, T=None, bins=1000, fn_vec=None):
    """``n`` photons created directly in device memory when the simulation first needs them (returns a
    PhotonBatch for ``sim.add_objs``).  Default: the distribution of ``generate_photons`` with its default
    sampler.  With a temperature ``T``: energies from the binned Planck distribution between ``min`` and
    ``max`` -- ``generate_photons_from_E([planck_phot_distribution(min, max, T, bins) ...])``
    (physicl/light.py:73-110) for all photons at once; the bin masses use the closed-form integral of the
    Planck density instead of ``bins`` calls to scipy.quad.
    ``fn_vec``: any other sampler, vectorised -- ``fn_vec(size) -> size numbers`` -- the bulk form of ``generate_photons``'s
    ``fn`` (physicl/light.py:112-128: ``E = min + (max - min) * fn()`` per photon): evaluated on the host in chunks of 4M
    photons, in photon order, and uploaded; ``fn_vec=lambda size: np.random.power(3, size)`` after ``np.random.seed(s)``
    gives photon i the energy ``generate_photons`` gives it after the same seed (numpy fills an array from the stream
    its scalar calls walk).  No Python object per photon either way."""
    if fn_vec is not None:
        if T is not None:
            raise ValueError("generate_photons_bulk: give a temperature T or a sampler fn_vec, not both")
        return PhotonBatch(n, min, max, seed, fn_vec=fn_vec)
    if T is None:
        return PhotonBatch(n, min, max, seed)
    lo, hi, T_ = (float(np.asarray(v.__unscaled__() if isinstance(v, Measurement) else v)) for v in (min, max, T))
    grid = np.linspace(lo, hi, int(bins))
    xk = grid / (float(np.asarray(kB.__unscaled__())) * T_)
    mass = np.diff(-np.exp(-xk) * (xk ** 3 + 3 * xk ** 2 + 6 * xk + 6))
    cdf = np.cumsum(mass / mass.sum())
    cdf[-1] = 1.0
    scale = float(np.asarray(Measurement(1, "J**1").scale))          # table energies in code units
    return PhotonBatch(n, lo * scale, hi * scale, seed, table=(cdf, grid[:-1] * scale))


# ---------------------------------------------------------------------------------------------- helpers
def _kernel_const(x):
    """The reference pastes ``str(value)`` into the kernel call and parses it with ``np.double``
    (physicl/light.py:236, 287; physicl/__init__.py:648): the code-unit number."""
    return float(np.double(str(x)))


def _c_h_literals():
    """Values of the literals str(c), str(h).upper() inside the kernel text (physicl/light.py:301, 309)."""
    return float(str(c)), float(str(h).upper())


# ---------------------------------------------------------------------------------------------- delete
class ScatterDeleteStep(DeviceStep):
    """Removes each photon with probability ``A*n*|dr|`` per step (physicl/light.py:225-260): flag kernel
    + stable compaction of the whole state, fused in pcl_step_scatter_delete.  The reference hands its
    kernel ``A := n`` and ``n := A`` (light.py:236); the product is the same."""

    _fuse_role = "scatter_delete"

    def __init__(self, n, A):
        self.n, self.A = n, A
        self.built = False
        self.removed = 0

    def _kernel_consts(self):
        return _kernel_const(self.n), _kernel_const(self.A)      # kernel A := user n, n := user A (light.py:236)

    def _device_run(self, sim):
        hip, dev = sim._hip, sim._dev
        mode = sim._rng_mode()
        if mode == hip.RNG_INPUT:
            sim._host_randoms("delete")
        A_k, n_k = self._kernel_consts()
        alive, removed = dev.step_scatter_delete(A_k, n_k, mode, sim.seed, sim._next_launch())
        g = sim._global([alive, removed])
        sim._alive, self.removed = int(g[0]), int(g[1])


class ScatterDeleteStepReference(ScatterDeleteStep):
    """Second statement of the same step in the reference (physicl/light.py:131-223): same kernel maths with
    the argument order (dx, dy, dz, rand, n, A, result).  Unlike ScatterDeleteStep it has a CPU path
    (``__run_py``, light.py:216-223), selected by ``Simulation(cl_on=False)``: that path removes photons from the list
    it is iterating over, so the object after every removed photon is skipped -- neither tested nor drawing a random
    number (456 instead of the expected 586 removals of 2000 photons per step at pcoll 0.3).  Reproduced as it is."""

    def _device_run(self, sim):
        if not sim._py_semantics():
            return ScatterDeleteStep._device_run(self, sim)
        from . import pyorder
        dev = sim._dev
        A_k, n_k = self._kernel_consts()
        pcoll = dev.scatter_pcoll(A_k, n_k, 0, 0.0, 0.0)                     # n * A * |dr|   light.py:220
        photon = None if sim._all_photons else dev.download_kind() != 0
        alive, removed = dev.step_delete_flags(pyorder.flags_delete_reference_py(pcoll, photon))
        sim._alive, self.removed = alive, removed


# ---------------------------------------------------------------------------------------------- isotropic scatter
class ScatterIsotropicStep(DeviceStep):
    """Isotropic re-direction with probability ``A * n * |dr|`` (physicl/light.py:262-359).

    Options as in the reference: ``wavelength_dep_scattering`` multiplies by ``pow((h*c)/E, -4)``;
    ``variable_n`` replaces ``n`` by the OpenCL-C expression ``variable_n_fn`` over ``r0[gid]``,
    ``r1[gid]``, ``r2[gid]`` (compiled into the kernel with hipRTC).

    Reference quirk kept: the kernel receives ``A := n`` and ``n := A`` (light.py:287), so with
    ``variable_n=True`` the user's ``A`` is unused and ``n`` (default 1) scales the probability.
    """
    _fuse_role = "scatter_iso"

    def __init__(self, **kwargs):
        self.n = kwargs.get("n", 1)
        self.A = kwargs.get("A", 1)
        self.wavelength_dep_scattering = kwargs.get("wavelength_dep_scattering", False)
        self.variable_n = kwargs.get("variable_n", False)
        self.variable_n_fn = kwargs.get("variable_n_fn", None)
        self.prog = None
        self.built = False

    def _kernel_params(self, sim):
        hip = sim._hip
        flags = (hip.SCATTER_WAVELENGTH if self.wavelength_dep_scattering else 0) | \
                (hip.SCATTER_VARIABLE_N if self.variable_n else 0)
        c_lit, h_lit = _c_h_literals()
        expr = str(self.variable_n_fn) if self.variable_n else None
        return dict(A=_kernel_const(self.n), n=_kernel_const(self.A), flags=flags, c=c_lit, h=h_lit, n_expr=expr)

    def _run_py_semantics(self, sim):
        """``Simulation(cl_on=False)``: ScatterIsotropicStep.__run_py (physicl/light.py:335-350) -- per photon one
        draw for the decision and, only on a hit, phi then theta; a hit leaves ``dv = v_old``; ``variable_n`` is
        ignored ("this does not support variable n scattering", light.py:334).  The host walks the np.random stream
        against the device's collision probabilities (physicl_amd/pyorder.py); kernel and write-back run on the device."""
        from . import pyorder
        hip, dev = sim._hip, sim._dev
        flags = hip.SCATTER_WAVELENGTH if self.wavelength_dep_scattering else 0
        A_k, n_k = _kernel_const(self.n), _kernel_const(self.A)
        c_val, h_val = float(np.asarray(c)), float(np.asarray(h))            # the path multiplies the Measurements themselves
        pcoll = dev.scatter_pcoll(A_k, n_k, flags, c_val, h_val)             # n * A * |dr| [* ((h*c)/E)**-4]   light.py:339-341
        photon = None if sim._all_photons else dev.download_kind() != 0
        rtheta, rphi, rand, hits = pyorder.draw_isotropic_py(pcoll, photon)
        for w, arr in enumerate((rtheta, rphi, rand)):
            dev.upload_rand(w, arr)
        got = dev.step_scatter_isotropic(A_k, n_k, flags | hip.SCATTER_PY_DV, c_val, h_val, None, hip.RNG_INPUT, 0, 0)
        if got != hits:
            raise RuntimeError("cl_on=False scatter: the host walked %d hits, the device applied %d" % (hits, got))
        sim._scattered = True
        sim.hits = hits

    def _device_run(self, sim):
        if sim._py_semantics():
            return self._run_py_semantics(sim)
        hip, dev = sim._hip, sim._dev
        p = self._kernel_params(sim)
        mode = sim._rng_mode()
        if mode == hip.RNG_INPUT:
            sim._host_randoms("iso")
        hits = dev.step_scatter_isotropic(p["A"], p["n"], p["flags"], p["c"], p["h"], p["n_expr"], mode, sim.seed,
                                          sim._next_launch())
        sim._scattered = True
        sim.hits = int(sim._global([hits])[0])


class ScatterSphericalStep(ScatterIsotropicStep):
    """Spelling used by the shipped examples: ``ScatterSphericalStep(n, A, wavelength_dep_scattering=...)``
    (examples/runtime1.py:77, examples/variable_n_scattering.ipynb:56)."""

    def __init__(self, n=1, A=1, **kwargs):
        super().__init__(n=n, A=A, **kwargs)


# ---------------------------------------------------------------------------------------------- measure steps
def _plane_axis(loc):
    loc = np.asarray(loc, dtype=np.float64).reshape(3)
    return loc


class _CountingMeasure(DeviceStep, MeasureStep):
    """Measure steps whose rows are counters: one fused reduction on the device."""
    _fuse_role = "measure"

    def _n_planes(self):
        return 0

    def _plane_rows(self):
        return []

    def _device_run(self, sim):
        cnt = sim._dev.step_counters(self._plane_rows())
        g = sim._global(cnt)
        self._record(sim, int(g[0]), g[1:4], g[4:])

    def _record_rows(self, sim, ts, n, sign, planes):
        """The rows of several passes at once (a K-pass launch): what ``_record`` appends pass by pass.  With a plain-number
        clock the rows are cut from ONE float array (``np.array([t, N, ...])`` of a float and integers is a float64 array:
        same values, same dtype); a clock with units takes the row-by-row way."""
        if not all(type(t) in (float, np.float64) for t in ts):
            t_keep = sim.t
            for i, t in enumerate(ts):
                sim.t = t
                self._record(sim, int(n[i]), sign[i], planes[i])
            sim.t = t_keep
            return
        cols = self._row_columns(n, sign, planes)
        block = np.empty((len(ts), 1 + len(cols)), dtype=np.float64)
        block[:, 0] = ts
        for c, col in enumerate(cols):
            block[:, 1 + c] = col
        self.data.extend(list(block))

    def _row_columns(self, n, sign, planes):
        raise NotImplementedError


class ScatterMeasureStep(_CountingMeasure):
    """Row per step: ``[t, N, crossings of plane 0, ...]`` (physicl/light.py:361-404).  A plane is a
    3-vector with NaN in the coordinates that do not define it.  With ``measure_E`` each plane's count is followed
    by the list of the crossing photons' energies (object order), gathered on the device."""

    def __init__(self, out_fn, measure_n=True, measure_locs=[], measure_E=False):
        MeasureStep.__init__(self, out_fn)
        self.measure_locs, self.measure_n, self.measure_E = measure_locs, measure_n, measure_E
        if measure_E:
            # rows carry variable-length energy lists: a separate gather per plane after the counters, outside the
            # fused kernels (this instance takes no part in step fusion / steps_per_launch)
            self._fuse_role = None

    def _device_run(self, sim):
        if not self.measure_E:
            return _CountingMeasure._device_run(self, sim)
        dev = sim._dev
        cnt = dev.step_counters(self._plane_rows())
        glob = sim._global(cnt)                                              # counts over all shards
        hip = sim._hip
        row = [sim.t]
        if self.measure_n:
            row.append(int(glob[hip.CNT_N]))
        for p, loc in enumerate(self._plane_rows()):                        # physicl/light.py:378-402
            row.append(int(glob[hip.CNT_PLANE0 + p]))
            Es = dev.plane_energies(loc, n_hint=int(cnt[hip.CNT_PLANE0 + p]))
            if sim.comm is not None:
                Es = sim.comm.allgather_concat(Es)                           # rank order == particle order
            row.append(Es.tolist())                                          # crossing photons' E, object order
        out = np.empty(len(row), dtype=object)                               # ragged row, as the reference's np.array(out)
        out[:] = row
        self.data.append(out)

    def _n_planes(self):
        return len(self.measure_locs)

    def _plane_rows(self):
        return [np.asarray(loc, dtype=np.float64).reshape(3) for loc in self.measure_locs]

    def _record(self, sim, n, sign, planes):
        row = [sim.t]
        if self.measure_n:
            row.append(n)
        row.extend(int(x) for x in planes)
        self.data.append(np.array(row))

    def _row_columns(self, n, sign, planes):
        return ([n] if self.measure_n else []) + [planes[:, p] for p in range(planes.shape[1])]


class ScatterSignMeasureStep(_CountingMeasure):
    """Row per step: ``[t, N, #v_x>0, #v_y>0, #v_z>0]`` (physicl/light.py:406-431)."""

    def __init__(self, out_fn, measure_n=True):
        MeasureStep.__init__(self, out_fn)
        self.measure_n = measure_n

    def _record(self, sim, n, sign, planes):
        row = [sim.t]
        if self.measure_n:
            row.append(n)
        row.extend(int(x) for x in sign)
        self.data.append(np.array(row))

    def _row_columns(self, n, sign, planes):
        return ([n] if self.measure_n else []) + [sign[:, k] for k in range(3)]


def _DEFAULT_ID_INFO(x):
    """The reference's default ``lambda x: str(type(x))`` (light.py:438), recognised by identity.  The label goes into the trace
    table's first column: this package's own classes read as the reference's (``<class 'physicl.light.PhotonObject'>``, what a
    trace file written by the reference holds -- tests/golden/g5_trace.npz), anybody else's class as it is."""
    return str(type(x)).replace("<class 'physicl_amd.", "<class 'physicl.", 1)


_PHOTON_LABEL = "<class 'physicl.light.PhotonObject'>"


class _Col:
    """The positions of tracked particle ``j`` over the first ``n`` logged passes: column j of the blocks the device returned
    ((passes, tracked, 4) arrays), made into vectors when somebody looks at them."""
    __slots__ = ("blocks", "j", "n")

    def __init__(self, blocks, j, n):
        self.blocks, self.j, self.n = blocks, j, n

    def __len__(self):
        return self.n

    def __iter__(self):
        left = self.n
        for b in self.blocks:
            if left <= 0:
                break
            k = min(left, len(b))
            yield from b[:k, self.j, :3]
            left -= k


class _Lazy(collections.abc.MutableSequence):
    """A list whose elements are made when somebody looks at them: the concatenation of ``parts`` (lists, or 2-D arrays whose
    rows are the elements).  The trace of 1000 photons over 500 passes is half a million position vectors; kept as the blocks
    the device returned, ``terminate`` costs a millisecond instead of a tenth of a second -- as long as the run itself
    (physicl/__init__.py:519-524 stops the run's clock after the steps' terminate).  Behaves like the reference's plain
    lists (``pos_dict[i]["pos"]``, the rows of ``data``) for everything a script does with them."""

    def __init__(self, parts=()):
        self._parts, self._list = list(parts), None

    def _all(self):
        if self._list is None:
            self._list = [x for part in self._parts for x in part]
            self._parts = None
        return self._list

    def __len__(self):
        return len(self._list) if self._list is not None else sum(len(part) for part in self._parts)

    def __getitem__(self, i):
        return self._all()[i]

    def __setitem__(self, i, x):
        self._all()[i] = x

    def __delitem__(self, i):
        del self._all()[i]

    def insert(self, i, x):
        self._all().insert(i, x)

    def append(self, x):
        self.extend([x])

    def extend(self, seq):
        if self._list is not None:
            self._list.extend(seq)
        elif self._parts and type(self._parts[-1]) is list and type(seq) is list:
            self._parts[-1].extend(seq)
        else:
            self._parts.append(seq)

    def __iter__(self):
        return iter(self._all())

    def __eq__(self, other):
        return list(self) == list(other)

    def __repr__(self):
        return repr(self._all())


class TracePathMeasureStep(MeasureStep):
    """Records objects' positions at every step (physicl/light.py:433-483).

    Two ways of running:

    * **tracked subset, on the device** -- when the step sits behind a light step in a fused group ([Newton][ScatterIsotropic |
      ScatterDelete][measures / this step]) and the run uses the device RNG: the positions of the tracked particles over
      the next launch's passes are worked out ahead of the launch by one thread per tracked particle
      (``pcl_store_trace_ahead``; photons do not interact and their random streams are keyed by their ids, so a particle's
      history is a function of its own state) -- nothing is downloaded per step, no Python object is built, the K-passes-
      per-launch schedule stays, and it works for a ``PhotonBatch`` of 1e8 photons and for sharded runs.  Which particles:
      ``trace_ids=[...]`` (ids = positions in the object list at upload, or photon numbers of a PhotonBatch), ``track=K``
      (the first K), default: every object of an explicit-object run (up to 65536), the first 1000 photons of a
      PhotonBatch.  The reference traces every object (O(N*T) host memory): with explicit objects that is the default here too.
    * **host plugin** otherwise (host-drawn randoms, ``cl_on=False``, a step list that is not fused): the first time an
      object is seen it is looked at as a Python object (``id_info_fn(obj)``, exactly as in the reference); after that, while
      the particles live on the device, a step costs three array downloads (ids, r, dv)."""
    _reads_only = True
    _fuse_role = "trace"
    MAX_TRACKED = 65536          # == PCL_TRACE_MAX (include/physicl_hip.h)

    def __init__(self, out_fn, trace_type=Object, id_info_fn=_DEFAULT_ID_INFO, trace_dv=False, track=None, trace_ids=None):
        super().__init__(out_fn)
        self.trace_type, self.id_info_fn, self.trace_dv = trace_type, id_info_fn, trace_dv
        self.track, self.trace_ids = track, trace_ids
        self.id_counter = 0
        self.id_dict, self.pos_dict = {}, {}
        self._tid_map, self._map_gen, self._log = None, None, []
        self._ahead_ids, self._ahead_gen, self._ahead_log, self._ahead_tids, self._ahead_objs = None, None, [], None, None

    # fused-group protocol of the counting measures (core.Simulation._build_plan): no planes, no counter row
    def _n_planes(self):
        return 0

    def _plane_rows(self):
        return []

    # ------------------------------------------------------------------ tracked subset, ahead of the launch
    def _ahead_set(self, sim):
        """The tracked ids (ascending int64) if this run can be traced on the device, else None."""
        if sim._py_semantics() or sim._dev is None or sim._rng_mode() != sim._hip.RNG_PHILOX:
            return None
        if self._ahead_ids is not None and self._ahead_gen == sim._upload_gen:
            return self._ahead_ids
        if self._ahead_ids is not None:
            self._flush_ahead(sim)                        # a new upload: file what the old population left
        n_all = sim._batch.n if sim._batch is not None else len(sim._objects._items) if isinstance(sim._uploaded, list) else None
        if n_all is None:
            return None                                   # (a materialised batch that went back up: host plugin)
        if self.trace_ids is not None:
            ids = np.unique(np.asarray(self.trace_ids, dtype=np.int64))
        else:
            k = self.track if self.track is not None else (1000 if sim._batch is not None else n_all)
            ids = np.arange(min(int(k), n_all), dtype=np.int64)
        ids = ids[(ids >= 0) & (ids < n_all)]
        if len(ids) > self.MAX_TRACKED:
            if self.track is None and self.trace_ids is None:
                return None                               # every object of a big explicit-object run: the host plugin, as before
            raise ValueError("TracePathMeasureStep tracks at most %d particles on the device, %d asked for" % (self.MAX_TRACKED, len(ids)))
        self._ahead_ids, self._ahead_gen = ids, sim._upload_gen
        self._ahead_tids = None
        # explicit objects: the objects behind the ids NOW (device id == place in the list that was just uploaded) -- by the time the
        # rows are filed the list may have lost its removed photons (a host visit) or been uploaded again
        explicit = sim._batch is None and isinstance(sim._uploaded, list)
        self._ahead_objs = [sim._objects._items[i] for i in ids.tolist()] if explicit else None
        return ids

    def _ahead_record(self, sim, ts, rows):
        """``rows`` = (len(ts), n_tracked, 4) of pcl_store_trace_ahead for the passes whose times are ``ts``."""
        if not self._ahead_log:
            self._ahead_t0 = _snap_t(ts[0])
        self._ahead_log.append(rows[:len(ts)])        # (a view of the launch's own array: nobody else writes it)

    def _assign_tids(self, sim, t0, present):
        """First sight of the tracked particles (light.py:450-455): a trace id each, in object order, unless the object
        already carries one (the host plugin saw it earlier, or an earlier upload of the same objects was traced).  A
        particle that is gone before the step first runs is never seen (``present``): no id, no row -- as in the reference."""
        explicit = self._ahead_objs is not None
        tids = np.full(len(self._ahead_ids), -1, dtype=np.int64)
        for j, i in enumerate(self._ahead_ids.tolist()):
            obj = self._ahead_objs[j] if explicit else None
            tid = obj.__dict__.get("__trace_path_id") if explicit else (self._uid_tid or {}).get(i)
            if tid is None and present[j]:
                tid = self.id_counter
                self.id_counter += 1
                if explicit:
                    obj.__dict__["__trace_path_id"] = tid
                    self.id_dict[tid] = self.id_info_fn(obj)
                else:
                    self.id_dict[tid] = _PHOTON_LABEL if self.id_info_fn is _DEFAULT_ID_INFO else self.id_info_fn(_batch_photon(sim, i))
                self.pos_dict[tid] = {"start": t0, "pos": _Lazy()}
                if self.trace_dv:
                    self.pos_dict[tid]["freq"] = 0
            if tid is not None:
                tids[j] = tid
        return tids

    def _flush_ahead(self, sim=None):
        """File the device rows under the trace ids (a photon's list ends where it was removed).  The blocks are not taken
        apart: a particle's positions are a view of its column (_Col) -- terminate() belongs to the run's clock."""
        if not self._ahead_log:
            return
        blocks, self._ahead_log = self._ahead_log, []
        comm = getattr(sim, "comm", None)
        if comm is not None and comm.world > 1:
            blocks = [_merge_shards(comm, np.concatenate(blocks, axis=0))]
        n_there = np.zeros(blocks[0].shape[1], dtype=np.int64)               # removal is for good: a prefix of the passes
        freq = np.zeros(blocks[0].shape[1], dtype=np.int64)
        first = None
        for blk in blocks:
            there = ~np.isnan(blk[:, :, 0])
            if first is None:
                first = there[0]
            n_there += there.sum(axis=0)
            if self.trace_dv:
                freq += ((blk[:, :, 3] != 0) & there).sum(axis=0)
        if self._ahead_tids is None:
            self._ahead_tids = self._assign_tids(sim, self._ahead_t0, first)
        n_there, freq = n_there.tolist(), freq.tolist()
        for j, tid in enumerate(self._ahead_tids.tolist()):
            if tid < 0 or not n_there[j]:
                continue
            entry = self.pos_dict[tid]
            entry["pos"].extend(_Col(blocks, j, n_there[j]))
            if self.trace_dv:
                entry["freq"] += freq[j]

    # ------------------------------------------------------------------ host plugin
    def _device_rows(self, sim):
        """(trace ids, positions, moved flags) of the resident particles straight from the device, or None when some
        particle has not been seen as an object yet / the state is not on the device / the run is sharded."""
        dev = getattr(sim, "_dev", None)
        if dev is None or sim._batch is not None or sim.comm is not None or sim._residency == "host" or not isinstance(sim._uploaded, list) or not sim._uploaded:
            return None
        if self._tid_map is None or self._map_gen != sim._upload_gen:
            self._tid_map = np.array([o.__dict__.get("__trace_path_id", -1) for o in sim._uploaded], dtype=np.int64)
            self._map_gen = sim._upload_gen
        with sim._dev_lock:
            n = dev.count
            idx = dev.download_ids(n) - sim._upload_lo
            if n and (idx.min() < 0 or idx.max() >= len(self._tid_map)):
                return None
            tids = self._tid_map[idx]
            if n and tids.min() < 0:
                return None
            hip = sim._hip
            r = np.stack([dev.download(f, n) for f in (hip.R0, hip.R1, hip.R2)], 1) if n else np.zeros((0, 3))
            moved = None
            if self.trace_dv:
                moved = np.zeros(n, dtype=bool)
                for f in (hip.DV0, hip.DV1, hip.DV2):
                    moved |= dev.download(f, n) != 0
        return tids, r, moved

    def _flush(self, sim):
        self._flush_ahead(sim)
        for tids, r, moved in self._log:
            for k, tid in enumerate(tids.tolist()):
                self.pos_dict[tid]["pos"].append(r[k])
            if moved is not None:
                for tid in tids[moved].tolist():
                    self.pos_dict[tid]["freq"] += 1
        self._log = []

    def run(self, sim):
        if self._ahead_ids is not None:
            # the tracked subset has been traced on the device so far: the objects the host plugin is about to walk carry
            # those trace ids on (explicit objects; a PhotonBatch's photons are looked up by their uid)
            self._flush(sim)
            self._adopt_ahead(sim)
        rows = self._device_rows(sim)
        if rows is not None:
            self._log.append(rows)
            return
        self._flush(sim)
        self._tid_map = None
        for obj in sim.objects:
            tid = obj.__dict__.get("__trace_path_id")
            if tid is None and self._uid_tid:
                tid = self._uid_tid.get(obj.__dict__.get("uid"))
                if tid is not None:
                    obj.__dict__["__trace_path_id"] = tid
            if tid is None:
                tid = obj.__dict__["__trace_path_id"] = self.id_counter
                self.id_dict[tid] = self.id_info_fn(obj)
                self.pos_dict[tid] = {"start": copy.deepcopy(sim.t), "pos": []}
                if self.trace_dv:
                    self.pos_dict[tid]["freq"] = 0
                self.id_counter += 1
            self.pos_dict[tid]["pos"].append(copy.deepcopy(obj.r))
            if self.trace_dv and np.any(np.asarray(obj.dv) != 0):
                self.pos_dict[tid]["freq"] += 1

    _uid_tid = None

    def _adopt_ahead(self, sim):
        """The run leaves the device-traced schedule (a host plugin joined, the objects were taken back, ...): from here on the
        particles are walked as Python objects, which must find the trace ids the device rows were filed under -- explicit
        objects carry them (_assign_tids), a PhotonBatch's photons are looked up by their uid."""
        ids, tids, explicit = self._ahead_ids, self._ahead_tids, self._ahead_objs is not None
        self._ahead_ids = self._ahead_gen = self._ahead_tids = self._ahead_objs = None
        if tids is not None and not explicit:
            self._uid_tid = dict(self._uid_tid or {}, **{int(i): int(t) for i, t in zip(ids, tids) if t >= 0})

    def terminate(self, sim):
        """data[0] = ["t", t0, t1, ...]; data[1+i] = [id info, (freq,) NaN-padded positions of object i]."""
        self._flush(sim)
        cols = len(sim.ts)
        table = [["t"] + copy.deepcopy(sim.ts)]
        for i in range(len(self.id_dict)):
            entry = self.pos_dict[i]
            head = [self.id_dict[i]]
            if self.trace_dv:
                head.append(entry["freq"])
            before = sim.ts.index(entry["start"])
            after = cols - len(entry["pos"])                                 # (as the reference counts it, light.py:477)
            table.append(_Lazy([head, [np.nan, np.nan, np.nan] * before, entry["pos"], [np.nan, np.nan, np.nan] * after]))
        self.data = table


def _snap_t(t):
    return t if isinstance(t, (int, float, np.generic)) else copy.deepcopy(t)


def _batch_photon(sim, i):
    """Photon ``i`` of a PhotonBatch as generate_photons would have made it (physicl/light.py:126-128), for id_info_fn."""
    # (the energy is looked up while the store still holds every photon at its own index; afterwards it is not known here)
    E = sim._dev.download(sim._hip.E, 1, int(i)) if (sim.comm is None and sim._dev.is_uniform() and i < sim._dev.count) else np.nan
    o = PhotonObject.__new__(PhotonObject)
    Object.__init__(o, E=np.double(np.asarray(E).reshape(-1)[0]), v=Measurement._from_code([float(np.asarray(c)), 0, 0], units="m**1 s**-1"),
                    uid=int(i))
    return o


def _merge_shards(comm, rows):
    """Every rank traced the same ids and holds NaN rows for the particles of other shards: one int64 sum all-reduce of the
    bit patterns (zeros where a rank has nothing) + of the "I have it" flags puts the table together on every rank."""
    mine = ~np.isnan(rows[:, :, 0])
    bits = np.where(mine[:, :, None], rows, 0.0).view(np.int64)
    tot = comm.allreduce_sum(np.concatenate([bits.reshape(-1), mine.astype(np.int64).reshape(-1)]))
    got = tot[:bits.size].reshape(rows.shape).view(np.float64).copy()
    owners = tot[bits.size:].reshape(mine.shape)
    got[owners != 1] = np.nan
    return got
