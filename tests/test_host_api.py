"""CPU: the plugin API of the host layer (no GPU needed: cl_on=False, host plugins only)."""
import threading

import numpy as np
import pytest

import physicl_amd as phys
import physicl_amd.light as light
import physicl_amd.newton as newton


def photons(n):
    return [light.PhotonObject(v=np.array([light.c, 0, 0], dtype=np.double), E=np.double(1), uid=i) for i in range(n)]


class Drift(phys.Step):
    """A user plugin written against the reference API: a per-object Python loop."""

    def run(self, sim):
        for o in sim.objects:
            o.r = o.r + np.asarray(o.v) * sim.dt


class Reaper(phys.Step):
    def run(self, sim):
        for o in list(sim.objects):
            if o.uid % 2 == 0 and sim.t >= 0.002:
                sim.remove_obj(o)


def test_alias_packages_expose_the_reference_names():
    import physicl
    import physicl.light
    import physicl.newton
    import phys
    import phys.light
    for mod in (physicl, phys):
        for name in ("Measurement", "MeasurementError", "Step", "UpdateTimeStep", "MeasureStep", "Object",
                     "Simulation", "CLInput", "CLOutput", "CLProgram"):
            assert hasattr(mod, name), name
    for name in ("c", "h", "kB", "PhotonObject", "E_from_wavelength", "wavelength_from_E", "planck_distribution",
                 "planck_probability", "planck_phot_distribution", "generate_photons", "generate_photons_from_E",
                 "ScatterDeleteStepReference", "ScatterDeleteStep", "ScatterIsotropicStep", "ScatterMeasureStep",
                 "ScatterSignMeasureStep", "TracePathMeasureStep", "ScatterSphericalStep"):
        assert hasattr(physicl.light, name) and hasattr(phys.light, name), name
    assert physicl.newton.NewtonianKinematicsStep is newton.NewtonianKinematicsStep
    assert physicl.Simulation is phys.Simulation is __import__("physicl_amd").Simulation


def test_simulation_runs_host_plugins_in_insertion_order_on_its_own_thread():
    order = []

    class Tag(phys.Step):
        def __init__(self, tag):
            self.tag = tag

        def run(self, sim):
            order.append((self.tag, threading.current_thread() is sim))

        def terminate(self, sim):
            order.append(("end" + self.tag, True))

    sim = phys.Simulation(cl_on=False, exit=lambda s: s.t >= 0.003)
    sim.add_step(2, phys.UpdateTimeStep(lambda s: np.double(0.001)))     # keys are labels, not an order
    sim.add_step(1, Tag("a"))
    sim.add_step(3, Tag("b"))
    with pytest.raises(IndexError):
        sim.add_step(1, Tag("dup"))
    sim.start()
    sim.join()
    assert order == [("a", True), ("b", True)] * 3 + [("enda", True), ("endb", True)]
    assert sim.ts == pytest.approx([0.001, 0.002, 0.003]) and not sim.running and sim.run_time >= 0
    sim.remove_step(3)
    assert list(sim.steps) == [2, 1]


def test_default_exit_state_and_get_state():
    sim = phys.Simulation(cl_on=False)
    assert sim.exit(sim) is True                     # no objects -> finished (physicl/__init__.py:414)
    sim.add_objs(photons(3))
    st = sim.get_state()
    assert st["objects"] == 3 and set(st) == {"objects", "t", "dt", "run_time"}
    sim.state_need_lock = True
    assert sim.get_state()["objects"] == 3
    assert sim.cl_ctx is None and sim.cl_q is None


def test_legacy_constructor_spellings():
    a = phys.Simulation({"cl_on": False, "exit": lambda c: True})
    b = phys.Simulation(params={"cl_on": False, "bounds": np.array([1, 2, 3])})
    assert a.cl_on is False and b.cl_on is False and list(b.bounds) == [1, 2, 3]


def test_host_plugins_see_and_modify_real_objects():
    sim = phys.Simulation(cl_on=False, exit=lambda s: s.t >= 0.004)
    sim.add_objs(photons(6))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, Drift())
    sim.add_step(2, Reaper())
    sim.start()
    sim.join()
    assert sim.error is None
    assert [o.uid for o in sim.objects] == [1, 3, 5] and len(sim.objects) == 3
    assert float(sim.objects[0].r[0]) == pytest.approx(4 * 0.001 * 299792458.0)
    assert sim.objects[0] in sim.objects and sim.objects.index(sim.objects[1]) == 1


def _no_gpu_here():
    from physicl_amd import _hip
    return _hip.device_count() == 0


@pytest.mark.skipif(not _no_gpu_here(), reason="needs a machine WITHOUT a GPU")
def test_device_steps_fail_loudly_without_a_gpu():
    """cl_on=False selects the reference's CPU-path semantics, not a CPU implementation: the steps still need the HIP
    device (created on first use) and raise when there is none -- nothing falls back to the host."""
    sim = phys.Simulation(cl_on=False, exit=lambda s: s.t >= 0.001)
    assert sim.cl_ctx is None and sim.cl_q is None and sim._dev is None          # as the reference: no context yet
    sim.add_objs(photons(2))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, newton.NewtonianKinematicsStep())
    with pytest.raises(RuntimeError, match="no ROCm-capable device|PCL_ERR"):
        sim.run()
    assert sim.running is False and isinstance(sim.error, RuntimeError)
    for step in (light.ScatterIsotropicStep(A=1e-3, n=1e-3), light.ScatterDeleteStep(1e-3, 1e-3),
                 light.ScatterDeleteStepReference(1e-3, 1e-3), light.ScatterSignMeasureStep(None)):
        with pytest.raises(RuntimeError, match="no ROCm-capable device|PCL_ERR"):
            step.run(sim)
    with pytest.raises(RuntimeError, match="no ROCm-capable device|PCL_ERR"):
        phys.Simulation(cl_on=True)                                              # the default creates the context at once


def test_fusion_plan_groups_only_adjacent_native_steps():
    sim = phys.Simulation(cl_on=False)                # (no device needed for planning)
    sim.cl_on = True                                  # cl_on=False would switch fusion off: see the end of this test
    upd, nk = phys.UpdateTimeStep(lambda s: 1e-3), newton.NewtonianKinematicsStep()
    sc = light.ScatterIsotropicStep(A=1e-3, n=1e-3)
    sign = light.ScatterSignMeasureStep(None)
    meas = light.ScatterMeasureStep(None, True, [[1.0, np.nan, np.nan]])
    dele = light.ScatterDeleteStep(1e-3, 1e-3)
    for i, s in enumerate((upd, nk, sc, sign, meas, Drift(), nk2 := newton.NewtonianKinematicsStep(), dele)):
        sim.add_step(i, s)
    plan = sim._build_plan()
    assert [k for k, _ in plan] == ["single", "fused", "single", "fused"]
    assert plan[1][1] == [nk, sc, sign, meas] and plan[3][1] == [nk2, dele]   # Newton + Delete: one pipeline
    sim.fuse = False
    assert all(k == "single" for k, _ in sim._build_plan())
    sim.fuse, sim.cl_on = True, False                 # the reference's CPU-path semantics: nothing is fused
    assert all(k == "single" for k, _ in sim._build_plan())


def test_measure_step_csv_and_kernel_glue_stubs(tmp_path):
    m = phys.MeasureStep(str(tmp_path / "out.csv"))
    m.data = [np.array([0.001, 5, 2]), np.array([0.002, 4, 1])]
    m.terminate(None)
    assert (tmp_path / "out.csv").read_text().splitlines() == ["0.001, 5.0, 2.0", "0.002, 4.0, 1.0"]
    i = phys.CLInput(name="d0", type="obj", obj_attr="dr[0]")
    o = phys.CLOutput(name="res", ctype="int")
    assert (i.name, i.ctype, o.ctype) == ("d0", "double", "int")
    sim = phys.Simulation(cl_on=False)
    prog = phys.CLProgram(sim, "k", "int gid = get_global_id(0); res[gid] = 1;")
    prog.prep_metadata, prog.output_metadata = [i, phys.CLInput(name="A", type="const", const_value="2.5")], [o]
    assert prog._signature() == [("double", "d0", True), ("double", "A", False), ("int", "res", True)]
    if _no_gpu_here():
        with pytest.raises(RuntimeError, match="no ROCm-capable device|PCL_ERR"):
            prog.build_kernel()
    m_e = light.ScatterMeasureStep(None, True, [[1.0, np.nan, np.nan]], measure_E=True)
    assert m_e._fuse_role is None and light.ScatterMeasureStep(None, True, [])._fuse_role == "measure"   # ragged rows: not fused


def test_kernel_constants_follow_the_reference_swap_and_code_scale():
    sim = phys.Simulation(cl_on=False)
    sim._hip = __import__("physicl_amd._hip", fromlist=["x"])
    st = light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.002), wavelength_dep_scattering=True,
                                    variable_n=True, variable_n_fn="1e-9 * exp(r0[gid] - 5)")
    p = st._kernel_params(sim)
    assert (p["A"], p["n"]) == (0.002, 0.001)                      # kernel A := user n (light.py:287)
    assert p["flags"] == 3 and p["n_expr"] == "1e-9 * exp(r0[gid] - 5)"
    assert (p["c"], p["h"]) == (299792458.0, 6.62607015e-34)
    n = phys.Measurement(2.0e25, "m**-3")
    assert light._kernel_const(n) == 2.0e25
    legacy = light.ScatterSphericalStep(0.5, 0.25, wavelength_dep_scattering=True)
    assert (legacy.n, legacy.A, legacy.wavelength_dep_scattering) == (0.5, 0.25, True)


def test_planck_helpers_are_usable_at_setup_time():
    d = light.planck_distribution(phys.Measurement(3e-19, "J**1"), 5778)
    assert d.units == {"M": -1, "L": -2, "T": 2} and 0 < float(d) < 1e19
    np.random.seed(0)
    E = light.planck_phot_distribution(light.E_from_wavelength(2500e-9), light.E_from_wavelength(200e-9), 5778, bins=50)
    assert E is None or 7e-20 < float(E) < 1e-18


def test_alias_package_does_not_import_light_eagerly():
    """Scripts set the code scale BEFORE importing physicl.light (code_unit_scale_test.ipynb:55): the alias
    package must not create c and h behind their back."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import physicl as phys; "
            "assert 'physicl_amd.light' not in sys.modules and 'physicl.light' not in sys.modules; "
            "phys.Measurement.set_code_scale('m', 0.001); import physicl.light as light; "
            "assert float(light.c) == 299792.458, float(light.c); print('ok')") % __import__("os").path.dirname(
                __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr[-500:]


def test_bulk_photons_from_a_vectorised_sampler_are_the_per_object_photons():
    """generate_photons_bulk(fn_vec=...) (physicl/light.py:112-128's ``fn``, vectorised): photon i gets the energy
    generate_photons gives it after the same seed; the sampler is walked for every photon of the batch whatever the shard."""
    import physicl_amd as phys
    from physicl_amd import light
    from physicl_amd.core import PhotonBatch
    lo_e, hi_e = float(light.E_from_wavelength(700e-9)), float(light.E_from_wavelength(200e-9))
    np.random.seed(11)
    objs = light.generate_photons(300, min=lo_e, max=hi_e)
    want = np.array([float(np.asarray(o.E)) for o in objs])
    b = light.generate_photons_bulk(300, min=lo_e, max=hi_e, fn_vec=lambda size: np.random.power(3, size))
    assert isinstance(b, PhotonBatch) and b.fn_vec is not None
    PhotonBatch.FN_CHUNK = 128                     # three chunks
    try:
        np.random.seed(11)
        got = np.empty(300)
        for off, E in b.host_energies(0, 300):
            got[off:off + len(E)] = E
        assert np.array_equal(got, want)
        np.random.seed(11)                          # a shard [100, 250): the same numbers for its photons
        part = np.empty(150)
        for off, E in b.host_energies(100, 250):
            part[off:off + len(E)] = E
        assert np.array_equal(part, want[100:250])
        with pytest.raises(ValueError):
            list(light.generate_photons_bulk(10, fn_vec=lambda size: np.zeros(size + 1)).host_energies(0, 10))
    finally:
        PhotonBatch.FN_CHUNK = 1 << 22
    with pytest.raises(ValueError):
        light.generate_photons_bulk(10, T=300.0, fn_vec=lambda size: np.zeros(size))


def test_limits_of_the_host_layer_are_the_headers():
    """Numbers the host layer repeats from include/physicl_hip.h: the measure planes a fused group may carry, the particles a
    TracePathMeasureStep may track on the device, the rows of a K-pass launch."""
    import os
    import re
    from physicl_amd import _hip, core
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "physicl_hip.h")).read()
    define = lambda name: int(re.search(r"#define\s+%s\s+(\d+)" % name, text).group(1))
    assert core._MAX_PLANES == _hip.MAX_PLANES == define("PCL_MAX_PLANES")
    assert light.TracePathMeasureStep.MAX_TRACKED == define("PCL_TRACE_MAX")
    assert (_hip.PHASE_ISOTROPIC, _hip.PHASE_DELETE) == (define("PCL_PHASE_ISOTROPIC"), define("PCL_PHASE_DELETE"))


def test_public_surface_is_the_reference_s(tmp_path):
    """tests/golden/g14_api.json: every class and function the reference's three modules define (read off the imported
    reference by make_golden.py: names, parameter names and kinds, literal defaults, base class names).  Here: every name
    exists, every method exists, its parameters start with the reference's (extras only behind them, with defaults), literal
    defaults are equal, and the reference's base classes are among the class's ancestors."""
    import inspect
    import json
    import os
    import physicl
    import physicl.light
    import physicl.newton
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "g14_api.json")))
    mods = {"physicl": physicl, "physicl.light": physicl.light, "physicl.newton": physicl.newton}

    def check(full, fn, want):
        got = list(inspect.signature(fn).parameters.values())
        got = [p for p in got if p.kind is not inspect.Parameter.VAR_POSITIONAL or any(w[1] == "VAR_POSITIONAL" for w in want)]
        assert len(got) >= len(want), full
        for p, (name, kind, lit) in zip(got, want):
            assert str(p.kind) == kind, (full, p.name)
            if kind not in ("VAR_KEYWORD", "VAR_POSITIONAL"):
                assert p.name == name, (full, p.name, name)
            if lit is None:
                assert p.default is inspect._empty, (full, name)
            elif lit != "<object>":
                assert repr(p.default) == lit, (full, name, p.default, lit)
            else:
                assert p.default is not inspect._empty, (full, name)
        for p in got[len(want):]:                         # this build's own extras: optional, behind the reference's
            assert p.default is not inspect._empty or p.kind in (inspect.Parameter.VAR_KEYWORD, inspect.Parameter.VAR_POSITIONAL), (full, p.name)

    assert len(ref["api"]) >= 25
    for full, e in ref["api"].items():
        modname, name = full.rsplit(".", 1)
        obj = getattr(mods[modname], name, None)
        assert obj is not None, full
        if e["kind"] == "function":
            check(full, obj, e["params"])
            continue
        assert inspect.isclass(obj), full
        ancestors = {c.__name__ for c in obj.__mro__}
        assert set(e["bases"]) <= ancestors, (full, e["bases"], sorted(ancestors))
        for mn, want in e["methods"].items():
            m = getattr(obj, mn, None)
            assert m is not None, (full, mn)
            check(full + "." + mn, m, want)
    for k, v in ref["light_constants"].items():
        assert float(np.asarray(getattr(physicl.light, k))) == v
