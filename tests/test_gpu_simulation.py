"""GPU: whole simulations through the reference's plugin API (``import physicl as phys``), compared
with the golden vectors recorded from the reference and with the reference's own two integration
tests (test/test_light.py:27-66)."""
import numpy as np
import pytest

import physicl as phys
import physicl.light
import physicl.newton

pytestmark = pytest.mark.gpu

C_LIT = 299792458.0
V_ABS_TOL = 4 * np.spacing(C_LIT)


def rand_ray():                                     # test/test_light.py:12-17
    return {"s": np.array([0] * 3, dtype=np.double), "v": np.array([phys.light.c, 0, 0], dtype=np.double),
            "E": np.double(1)}


def make_sim(n=10000, **kw):                        # test/test_light.py:19-24
    s = phys.Simulation(bounds=np.array([1000, 1000, 1000]), cl_on=True, exit=lambda cond: cond.t >= 0.100, **kw)
    for i in range(n):
        s.add_obj(phys.light.PhotonObject(uid=i, **rand_ray()))
    return s


def run(sim):
    sim.start()
    sim.join()
    assert sim.error is None
    return sim


@pytest.mark.parametrize("rng", ["numpy", "philox"])
def test_scatter_spherical(rng):
    """test/test_light.py:27-43, thresholds unchanged."""
    np.random.seed(3)
    x = make_sim(rng=rng, seed=3)
    x.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    x.add_step(1, phys.newton.NewtonianKinematicsStep())
    x.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
    step = phys.light.ScatterSignMeasureStep(None, True)
    x.add_step(3, step)
    run(x)
    error = (np.double(step.data[0][1] * 0.5) - (sum([y[2] for y in step.data]) / len(step.data))) / \
        np.double(step.data[0][1] * 0.5)
    assert np.isclose(error, 0, 0, 0.10)
    assert len(step.data) == 100 and step.data[0][1] == 10000


@pytest.mark.parametrize("rng", ["numpy", "philox"])
def test_scatter_delete(rng):
    """test/test_light.py:45-66, thresholds unchanged."""
    np.random.seed(4)
    x = make_sim(rng=rng, seed=4)
    x.exit = lambda x: len(x.objects) == 0
    N_i = len(x.objects)
    x.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    x.add_step(1, phys.newton.NewtonianKinematicsStep())
    n, A = 0.001, 0.001
    x.add_step(2, phys.light.ScatterDeleteStep(np.double(n), np.double(A)))
    step = phys.light.ScatterMeasureStep(None, True, [[1 / (n * A), np.nan, np.nan]])
    x.add_step(3, step)
    run(x)
    N_x = sum(step.data[2])
    error = (np.e ** -1 - (N_x / N_i)) / (np.e ** -1)
    assert np.isclose(error, 0, 0, 0.10)
    assert len(x.objects) == 0 and step.data[-1][1] == 0


@pytest.mark.parametrize("fuse", [True, False])
def test_seeded_isotropic_run_reproduces_the_reference(golden, fuse):
    """np.random.seed + the same step list as the reference's OpenCL path: the measure rows (exact
    integers) and the final velocities match the golden run, fused into one kernel or not."""
    z = golden("g2_iso_base")
    N, K, dt = len(z["k0_rand"]), int(z["K"]), float(z["dt"])
    sim = phys.Simulation(cl_on=True, exit=lambda s: s.t >= (K - 0.5) * dt, fuse=fuse)
    sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i)
                  for i in range(N)])
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(dt)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(z["A_user"]), n=np.double(z["n_user"])))
    sign = phys.light.ScatterSignMeasureStep(None, True)
    meas = phys.light.ScatterMeasureStep(None, True, [np.array(p) for p in z["planes"]])
    sim.add_step(3, sign)
    sim.add_step(4, meas)
    np.random.seed(int(z["seed"]))
    run(sim)
    assert np.array_equal(np.array(sign.data, dtype=np.float64), z["sign_rows"])
    assert np.array_equal(np.array(meas.data, dtype=np.float64), z["measure_rows"])
    assert sim.hits == (~np.isnan(z["k%d_res0" % (K - 1)])).sum()
    v = np.array([np.asarray(o.v) for o in sim.objects])
    dv = np.array([np.asarray(o.dv) for o in sim.objects])
    r = np.array([np.asarray(o.r) for o in sim.objects])
    dr = np.array([np.asarray(o.dr) for o in sim.objects])
    last = "k%d_post_" % (K - 1)
    assert np.max(np.abs(v - z[last + "v"])) <= V_ABS_TOL and np.max(np.abs(dv - z[last + "dv"])) <= 2 * V_ABS_TOL
    assert np.max(np.abs(dr - z[last + "dr"])) <= V_ABS_TOL * dt + 1e-12
    assert np.max(np.abs(r - z[last + "r"])) <= K * V_ABS_TOL * dt + 4 * np.spacing(np.abs(r).max())
    assert [o.uid for o in sim.objects] == list(range(N))


def test_seeded_delete_run_reproduces_the_reference(golden):
    z = golden("g4_delete")
    N = int(z["N"])
    sim = phys.Simulation(cl_on=True)                                   # default exit: no objects left
    sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i)
                  for i in range(N)])
    survivors = []

    class Census(phys.Step):                                            # a user plugin that walks sim.objects
        def run(self, s):
            survivors.append([o.uid for o in s.objects])

    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(z["dt"])))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterDeleteStep(np.double(z["n_user"]), np.double(z["A_user"])))
    meas = phys.light.ScatterMeasureStep(None, True, [np.array(p) for p in z["planes"]])
    sign = phys.light.ScatterSignMeasureStep(None, True)
    sim.add_step(3, meas)
    sim.add_step(4, sign)
    sim.add_step(5, Census())
    np.random.seed(int(z["seed"]))
    run(sim)
    K = int(z["K"])
    assert len(survivors) == K
    for k in range(K):
        assert survivors[k] == list(z["k%d_survivor_uid" % k])
    assert np.array_equal(np.array(meas.data, dtype=np.float64), z["measure_rows"])
    assert np.array_equal(np.array(sign.data, dtype=np.float64), z["sign_rows"])
    assert len(sim.objects) == 0 and sim.ts == pytest.approx([float(z["dt"]) * (k + 1) for k in range(K)])


def test_host_plugin_between_device_steps_round_trips_the_state():
    """A user Step that edits objects in Python sits between device steps: the state goes
    device -> objects -> device every pass and nothing is lost or reordered."""
    class Kick(phys.Step):
        def run(self, s):
            for o in s.objects:
                if o.uid == 7:
                    o.r = o.r + np.array([0.0, 1.0, 0.0])

    def build(with_plugin):
        sim = phys.Simulation(cl_on=True, rng="philox", seed=9, exit=lambda s: s.t >= 0.0045)
        sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i)
                      for i in range(500)])
        sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
        sim.add_step(1, phys.newton.NewtonianKinematicsStep())
        if with_plugin:
            sim.add_step(2, Kick())
        sim.add_step(3, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
        return run(sim)

    a, b = build(False), build(True)
    ra = np.array([np.asarray(o.r) for o in a.objects])
    rb = np.array([np.asarray(o.r) for o in b.objects])
    assert np.array_equal(ra[:, 0], rb[:, 0]) and np.array_equal(ra[:, 2], rb[:, 2])
    d = rb[:, 1] - ra[:, 1]
    assert d[7] == pytest.approx(5.0) and np.count_nonzero(np.delete(d, 7)) == 0
    assert np.array_equal(np.array([np.asarray(o.v) for o in a.objects]), np.array([np.asarray(o.v) for o in b.objects]))


def test_mixed_objects_and_trace_path():
    sim = phys.Simulation(cl_on=True, rng="philox", seed=2, exit=lambda s: s.t >= 0.0035)
    objs = []
    for i in range(40):
        if i % 4 == 0:
            objs.append(phys.Object(v=phys.Measurement([1.0, 2.0, 3.0], "m**1 s**-1"), uid=i))
        else:
            objs.append(phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i))
    sim.add_objs(objs)
    tp = phys.light.TracePathMeasureStep(None, trace_dv=True)
    sim.add_step(3, phys.UpdateTimeStep(lambda s: 0.001))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterIsotropicStep(A=1.0, n=1.0))           # pcoll >> 1: every photon scatters
    sim.add_step(0, tp)
    run(sim)
    assert len(tp.data) == 41 and tp.data[0][0] == "t" and len(tp.data[0]) == 5
    plain = sim.objects[0]
    assert np.allclose(np.asarray(plain.r), 4 * 0.001 * np.array([1.0, 2.0, 3.0]))
    assert np.array_equal(np.asarray(plain.v), [1.0, 2.0, 3.0]) and not np.any(np.asarray(plain.dv))
    photon_row, plain_row = tp.data[2], tp.data[1]
    assert photon_row[1] == 4 and plain_row[1] == 0                         # freq: dv != 0 on every step / never
    assert len(photon_row) == 2 + 4 and np.asarray(photon_row[2]).shape == (3,)
    speeds = [np.linalg.norm(np.asarray(o.v)) for o in sim.objects if type(o) is phys.light.PhotonObject]
    assert np.allclose(speeds, C_LIT, rtol=1e-15)


def test_photon_batch_runs_without_python_objects():
    n = 2_000_000
    sim = phys.Simulation(cl_on=True, seed=12, exit=lambda s: s.t >= 0.0095)
    sim.add_objs(phys.light.generate_photons_bulk(n, min=phys.light.E_from_wavelength(700e-9),
                                                  max=phys.light.E_from_wavelength(200e-9), seed=12))
    assert len(sim.objects) == n and sim.rng == "philox"
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
    sign = phys.light.ScatterSignMeasureStep(None, True)
    sim.add_step(3, sign)
    run(sim)
    assert len(sign.data) == 10 and all(row[1] == n for row in sign.data)
    p = 1e-6 * C_LIT * 1e-3
    assert abs(sim.hits - n * p) < 6 * np.sqrt(n * p)
    frac_xp = sign.data[-1][2] / n                                          # -> 1/2 as photons randomise
    assert 0.5 < frac_xp < (1 - p) ** 10 + 0.5 and abs(sign.data[-1][3] / n - 0.5 * (1 - (1 - p) ** 10)) < 0.01
    r = sim.download("r")
    assert r.shape == (n, 3) and np.isfinite(r).all()
    E = sim.download("E")
    assert E.min() >= float(phys.light.E_from_wavelength(700e-9)) and E.max() <= float(phys.light.E_from_wavelength(200e-9))
    with pytest.raises(NotImplementedError):
        sim.add_obj(phys.Object())


def test_variable_n_example_through_the_plugin_api():
    """examples/variable_n_scattering.ipynb:52-60 with the legacy spellings, small N."""
    import phys as old
    import phys.light
    import phys.newton
    cl_n = "0.000000001 * exp(r0[gid] - 5)"
    sim = old.Simulation(cl_on=True, exit=lambda cond: cond.t >= 0.0245)
    sim.add_step(2, old.UpdateTimeStep(lambda c: 0.005))
    sim.add_step(1, old.newton.NewtonianKinematicsStep())
    sim.add_step(3, old.light.ScatterSphericalStep(0.000000000000001, 0.0000000000000000001,
                                                   wavelength_dep_scattering=True, variable_n=True, variable_n_fn=cl_n))
    tp = old.light.TracePathMeasureStep(None)
    sim.add_step(0, tp)
    np.random.seed(1)
    sim.add_objs(old.light.generate_photons(200, bins=10, min=old.light.E_from_wavelength(200e-9),
                                            max=old.light.E_from_wavelength(700e-9)))
    run(sim)
    assert len(tp.data) == 201 and len(tp.data[1]) == 1 + 5
    # the exp() overflow regime: every photon at x > 0 scatters every step (SURVEY.md 8(d) caveat)
    assert sim.hits > 0
    with pytest.raises(ValueError):
        bad = old.Simulation(cl_on=True, exit=lambda c: c.t >= 0.001)
        bad.add_objs(old.light.generate_photons(2, min=1e-19, max=2e-19))
        bad.add_step(0, old.UpdateTimeStep(lambda c: 0.001))
        bad.add_step(1, old.light.ScatterSphericalStep(1, 1, variable_n=True, variable_n_fn="r0[gid+1]"))
        bad.run()


def test_device_info():
    info = phys.Simulation.get_device_info()
    assert len(info) >= 1 and all("gfx950" in k for k in info)


def test_foreign_threads_can_poll_and_touch_objects_while_the_simulation_runs():
    """The reference's usage pattern: the simulation runs on its own thread while the main thread polls
    get_state() (every example) -- and may even index sim.objects.  len()/get_state never touch the device;
    object access waits for the pass in flight, syncs the state back and the run continues from it."""
    import time
    sim = phys.Simulation(cl_on=True, rng="philox", seed=5, exit=lambda s: s.t >= 0.2995)
    sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i)
                  for i in range(2000)])
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
    sign = phys.light.ScatterSignMeasureStep(None, True)
    sim.add_step(3, sign)
    sim.start()
    polls, touched = 0, 0
    while sim.running or polls == 0:
        st = sim.get_state()
        assert st["objects"] == 2000
        polls += 1
        if polls % 3 == 0 and sim.running:
            o = sim.objects[17]                      # forces a device -> host sync between two passes
            assert o.uid == 17 and np.isfinite(np.asarray(o.r)).all()
            touched += 1
        time.sleep(0.002)
    sim.join()
    assert sim.error is None and len(sign.data) == 300 and polls > 1
    # the interleaved syncs did not disturb the physics: same result as an undisturbed run
    ref = phys.Simulation(cl_on=True, rng="philox", seed=5, exit=lambda s: s.t >= 0.2995)
    ref.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i)
                  for i in range(2000)])
    ref.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    ref.add_step(1, phys.newton.NewtonianKinematicsStep())
    ref.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
    sign2 = phys.light.ScatterSignMeasureStep(None, True)
    ref.add_step(3, sign2)
    run(ref)
    assert np.array_equal(np.array(sign.data), np.array(sign2.data))
    assert np.array_equal(np.array([np.asarray(o.r) for o in sim.objects]), np.array([np.asarray(o.r) for o in ref.objects]))


def test_two_simulations_back_to_back_share_the_device():
    """runtime1-style: several Simulations (contexts) in one process, one after the other and alive together."""
    sims = []
    for seed in (1, 2):
        s = phys.Simulation(cl_on=True, rng="philox", seed=seed, exit=lambda c: c.t >= 0.0095)
        s.add_objs(phys.light.generate_photons_bulk(50_000, min=1e-19, max=2e-19, seed=seed))
        s.add_step(0, phys.UpdateTimeStep(lambda c: np.double(0.001)))
        s.add_step(1, phys.newton.NewtonianKinematicsStep())
        s.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001),
                                                      variable_n=True, variable_n_fn="0.001 * exp(r1[gid] / 1e9)"))
        sims.append(s)
    for s in sims:
        s.start()
    for s in sims:
        s.join()
        assert s.error is None and len(s.ts) == 10 and s.hits > 0
    assert sims[0].hits != sims[1].hits


def test_user_step_built_on_clprogram(golden):
    """A user-defined Step written against the reference's kernel-glue API (CLInput / CLOutput / CLProgram,
    physicl/__init__.py:543-664): gather per-object inputs, run an OpenCL-C body, act on the result.
    Here: an absorber like ScatterDeleteStep, with its own kernel text; checked against the oracle."""
    from oracle import physicl_oracle as orc

    class MyAbsorber(phys.Step):
        def __init__(self, sigma):
            self.sigma, self.prog = sigma, None

        def run(self, sim):
            if self.prog is None:
                skip = phys.CLInput(name="only_photons", type="obj_action",
                                    code="if type(obj) != physicl.light.PhotonObject:\n \t\t continue")
                d = [phys.CLInput(name="d%d" % k, type="obj", obj_attr="dr[%d]" % k) for k in range(3)]
                u = phys.CLInput(name="u", type="obj_def", obj_def="np.random.random()")
                sg = phys.CLInput(name="sigma", type="const", const_value=str(self.sigma))
                who = phys.CLInput(name="who", type="obj_track", obj_track="obj")
                self.prog = phys.CLProgram(sim, "absorb", """
                    int gid = get_global_id(0);
                    double path = sqrt(d0[gid] * d0[gid] + d1[gid] * d1[gid] + d2[gid] * d2[gid]);
                    gone[gid] = (sigma * path >= u[gid]) ? 1 : 0;
                    depth[gid] = sigma * path;
                """)
                self.prog.prep_metadata = [skip] + d + [u, who, sg]
                self.prog.output_metadata = [phys.CLOutput(name="gone", ctype="int"), phys.CLOutput(name="depth")]
                self.prog.build_kernel()
            out = self.prog.run()
            self.last = out
            for idx, x in enumerate(out["gone"]):
                if x == 1:
                    sim.remove_obj(self.prog.who[idx])

    sim = phys.Simulation(cl_on=True, exit=lambda s: s.t >= 0.0025)
    objs = [phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i)
            for i in range(3000)]
    objs.insert(10, phys.Object(v=phys.Measurement([5.0, 0, 0], "m**1 s**-1"), uid=-1))     # skipped by the obj_action
    sim.add_objs(objs)
    absorber = MyAbsorber(np.double(1e-6))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, absorber)
    np.random.seed(99)
    run(sim)
    # replay with the oracle: same random stream (one draw per photon per step), same kernel maths
    rs = np.random.RandomState(99)
    alive = np.arange(3000)
    for step in range(3):
        n = len(alive)
        d0 = np.full(n, 299792458.0 * 0.001)
        flags = orc.delete_flags(d0, np.zeros(n), np.zeros(n), rs.random_sample(n), 1e-6, 1.0)
        alive = alive[orc.survivors(flags)]
    assert [o.uid for o in sim.objects] == [*alive[alive < 10], -1, *alive[alive >= 10]]    # order kept, Object untouched
    assert absorber.last["gone"].dtype == np.int32 and absorber.last["depth"].dtype == np.float64
    assert np.all(absorber.last["depth"] == 1e-6 * (299792458.0 * 0.001))
    with pytest.raises(phys._hip_error()):
        bad = phys.CLProgram(sim, "broken", "int gid = get_global_id(0); res[gid] = undefined_symbol;")
        bad.prep_metadata, bad.output_metadata = [phys.CLInput(name="d0", type="obj", obj_attr="dr[0]")], [phys.CLOutput(name="res")]
        bad.build_kernel()


# ============================================================================ steps_per_launch (K passes per launch)
def _batch_sim(n, K, t_end, dt_fn=lambda s: np.double(0.001), **kw):
    sim = phys.Simulation(cl_on=True, seed=77, exit=lambda s: s.t >= t_end, steps_per_launch=K, **kw)
    sim.add_objs(phys.light.generate_photons_bulk(n, min=phys.light.E_from_wavelength(700e-9),
                                                  max=phys.light.E_from_wavelength(200e-9), seed=5))
    sim.add_step(0, phys.UpdateTimeStep(dt_fn))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.002), wavelength_dep_scattering=False))
    sign = phys.light.ScatterSignMeasureStep(None, True)
    count = phys.light.ScatterMeasureStep(None, True, [])
    sim.add_step(3, sign)
    sim.add_step(4, count)
    return sim, sign, count


@pytest.mark.parametrize("K", [2, 5, 1000])       # 1000: clamped to the library maximum of 64
def test_steps_per_launch_is_bit_identical_to_one_pass_per_launch(K):
    """13 passes in launches of K: same ts, same measure rows (each with its own pass's t), same hits, same state."""
    n, t_end = 300_001, 0.0125
    ref, ref_sign, ref_count = _batch_sim(n, 1, t_end)
    run(ref)
    sim, sign, count = _batch_sim(n, K, t_end)
    run(sim)
    assert len(ref.ts) == 13 and [float(t) for t in sim.ts] == [float(t) for t in ref.ts] and float(sim.t) == float(ref.t)
    assert len(sign.data) == 13 and len(count.data) == 13
    for a, b in zip(sign.data + count.data, ref_sign.data + ref_count.data):
        assert [float(x) for x in a] == [float(x) for x in b]
    assert sim.hits == ref.hits
    for f in ("r", "v", "dr", "dv", "E", "id"):
        assert np.array_equal(sim.download(f), ref.download(f)), f


def test_steps_per_launch_with_changing_dt_and_ineligible_plans():
    """A time step that changes mid-run splits the launch at the change; a pass with a host plugin, host randoms or
    plane measures is simply run one launch per pass."""
    n = 50_001
    dt_fn = lambda s: np.double(0.001 if len(s.ts) < 4 else 0.0005)
    ref, ref_sign, _ = _batch_sim(n, 1, 0.0073, dt_fn)
    run(ref)
    sim, sign, _ = _batch_sim(n, 8, 0.0073, dt_fn)
    run(sim)
    assert [float(t) for t in sim.ts] == [float(t) for t in ref.ts] and len(ref.ts) == 11
    assert [[float(x) for x in r] for r in sign.data] == [[float(x) for x in r] for r in ref_sign.data]
    assert np.array_equal(sim.download("r"), ref.download("r")) and np.array_equal(sim.download("dv"), ref.download("dv"))
    # plane measure in the pass: not eligible, still correct
    out = []
    for K in (1, 8):
        s, sg, _ = _batch_sim(n, K, 0.0045)
        pl = phys.light.ScatterMeasureStep(None, True, [[100.0, np.nan, np.nan]])
        s.add_step(5, pl)
        run(s)
        out.append(([[float(x) for x in r] for r in pl.data], s.download("v")))
    assert out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1])


@pytest.mark.parametrize("K", [3, 64])
def test_steps_per_launch_delete_until_empty_matches_one_pass_per_launch(K):
    """test/test_light.py:47-66's delete run through the K-passes-per-launch path: same alive sequence, same measure
    rows, same ts (the run is cut at the pass that empties the store), same survivors' state."""
    def build(k):
        n = 40_000
        sim = phys.Simulation(cl_on=True, seed=21, steps_per_launch=k)           # default exit: len(objects) == 0
        sim.add_objs(phys.light.generate_photons_bulk(n, min=1.0, max=1.0, seed=2))
        sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
        sim.add_step(1, phys.newton.NewtonianKinematicsStep())
        sim.add_step(2, phys.light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
        m = phys.light.ScatterMeasureStep(None, True, [[C_LIT * 0.0035, np.nan, np.nan]])
        sg = phys.light.ScatterSignMeasureStep(None, True)
        sim.add_step(3, m)
        sim.add_step(4, sg)
        return sim, m, sg
    ref, rm, rs = build(1)
    run(ref)
    sim, m, sg = build(K)
    run(sim)
    assert len(ref.ts) > 15 and [float(t) for t in sim.ts] == [float(t) for t in ref.ts]
    assert [[float(x) for x in r] for r in m.data] == [[float(x) for x in r] for r in rm.data]
    assert [[float(x) for x in r] for r in sg.data] == [[float(x) for x in r] for r in rs.data]
    assert len(sim.objects) == 0 and float(sim.t) == float(ref.t)
    assert rm.data[3][2] > 0                                                    # the plane at 3.5 steps is crossed in pass 4

    # an exit on time leaves survivors: their state must agree too
    out = []
    for k in (1, K):
        s, _, _ = build(k)
        s.exit = lambda x: x.t >= 0.0045
        run(s)
        out.append((len(s.objects), s.download("r"), s.download("id"), s.download("dr")))
    assert out[0][0] == out[1][0] > 0
    for a, b in zip(out[0][1:], out[1][1:]):
        assert np.array_equal(a, b)


def test_scatter_measure_step_with_energies():
    """ScatterMeasureStep(measure_E=True), the row layout of physicl/light.py:374-404: [t, N, n_plane0, [E...], ...]."""
    n = 500
    sim = phys.Simulation(cl_on=True, exit=lambda s: s.t >= 0.0055, rng="philox", seed=1)
    Es = np.linspace(1.0, 2.0, n)
    sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(Es[i]), uid=i)
                  for i in range(n)])
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    m = phys.light.ScatterMeasureStep(None, True, [[2.5e-3 * C_LIT, np.nan, np.nan], [np.nan, 1.0, np.nan]], measure_E=True)
    sim.add_step(2, m)
    run(sim)
    assert len(m.data) == 6
    for k, row in enumerate(m.data):
        assert len(row) == 6 and row[1] == n and float(row[0]) == float(sim.ts[k])
        if k == 2:                                   # the third move takes every photon across x = 2.5 c dt
            assert row[2] == n and np.array_equal(np.array(row[3]), Es)
        else:
            assert row[2] == 0 and row[3] == []
        assert row[4] == 0 and row[5] == []


@pytest.mark.parametrize("kind", ["scatter", "delete"])
def test_trace_path_from_device_arrays_equals_the_per_object_walk(kind, monkeypatch):
    """TracePathMeasureStep as a host plugin (fuse=False: behind separate steps; in a fused group its tracked subset is worked
    out on the device, tests/test_gpu_trace.py) reads ids / r / dv arrays from the device once every object has been seen; the
    table is the one the per-object walk of the reference (light.py:447-483) builds -- also when a delete step thins the list."""
    def build():
        sim = phys.Simulation(cl_on=True, rng="philox", seed=4, fuse=False, exit=lambda s: s.t >= 0.0075 or len(s.objects) == 0)
        sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0 + i), uid=i)
                      for i in range(300)])
        tp = phys.light.TracePathMeasureStep(None, id_info_fn=lambda o: str(o.E), trace_dv=True)
        sim.add_step(0, phys.UpdateTimeStep(lambda s: 0.001))
        sim.add_step(1, phys.newton.NewtonianKinematicsStep())
        if kind == "scatter":
            sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
        else:
            sim.add_step(2, phys.light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
        sim.add_step(3, tp)
        return sim, tp
    fast, tf = build()
    calls = []
    orig = phys.light.TracePathMeasureStep._device_rows
    monkeypatch.setattr(phys.light.TracePathMeasureStep, "_device_rows",
                        lambda self, sim: calls.append(orig(self, sim)) or calls[-1])
    run(fast)
    assert sum(c is not None for c in calls) >= len(fast.ts) - 1 and calls[0] is None      # only the first sight is a walk
    monkeypatch.setattr(phys.light.TracePathMeasureStep, "_device_rows", lambda self, sim: None)
    slow, tsl = build()
    run(slow)
    # (with the delete step ahead of the trace step, photons removed in the very first pass are never seen)
    assert len(tf.data) == len(tsl.data) and (len(tf.data) == 301 if kind == "scatter" else 150 < len(tf.data) < 301)
    assert [float(t) for t in tf.data[0][1:]] == [float(t) for t in tsl.data[0][1:]]
    for a, b in zip(tf.data[1:], tsl.data[1:]):
        assert a[0] == b[0] and a[1] == b[1] and len(a) == len(b)
        for x, y in zip(a[2:], b[2:]):
            assert np.array_equal(np.asarray(x, dtype=float), np.asarray(y, dtype=float), equal_nan=True)
    if kind == "delete":
        assert any(np.isscalar(x) and np.isnan(x) for row in tf.data[1:] for x in row[2:])      # NaN padding after removal


def test_steps_per_launch_survives_another_thread_looking_at_the_objects():
    """with steps_per_launch > 1 a pass must re-upload after anything took the objects back to the host.
    The main thread keeps indexing ``sim.objects`` while the simulation thread runs K passes per launch: rows, times
    and final state are those of the undisturbed run (looking never changes a run: ids, hence random streams, stay)."""
    import time

    def build(K):
        sim = phys.Simulation(cl_on=True, rng="philox", seed=3, steps_per_launch=K, exit=lambda s: len(s.ts) >= 24)
        sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i)
                      for i in range(3000)])
        sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
        sim.add_step(1, phys.newton.NewtonianKinematicsStep())
        sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
        sign = phys.light.ScatterSignMeasureStep(None, True)
        sim.add_step(3, sign)
        return sim, sign
    ref, ref_sign = build(1)
    run(ref)
    sim, sign = build(4)
    sim.start()
    looks = 0
    while sim.is_alive():
        o = sim.objects[looks % 3000]                # brings the state to the host between launches
        assert o.uid == looks % 3000
        looks += 1
        time.sleep(0.001)
    sim.join()
    assert sim.error is None and looks > 0
    assert [[float(x) for x in r] for r in sign.data] == [[float(x) for x in r] for r in ref_sign.data] and len(sign.data) == 24
    for f in ("r", "v", "dv", "id"):
        assert np.array_equal(sim.download(f), ref.download(f)), f


@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_bulk_photons_from_a_user_sampler_never_become_python_objects(devices):
    """generate_photons_bulk(fn_vec=...): the energies of physicl/light.py:112-128 with the user's own (vectorised) sampler, on
    the device without a Python object per photon -- photon i has the energy generate_photons gives it after the same seed,
    on one context and sharded over two."""
    n = 100_003
    lo_e, hi_e = phys.light.E_from_wavelength(700e-9), phys.light.E_from_wavelength(200e-9)
    np.random.seed(5)
    want = np.array([float(np.asarray(o.E)) for o in phys.light.generate_photons(2000, fn=lambda: np.random.beta(2.0, 5.0), min=lo_e, max=hi_e)])
    kw = dict(cl_on=True, seed=3, exit=lambda s: s.t >= 0.0035)
    if devices:
        kw["devices"] = devices
    sim = phys.Simulation(**kw)
    np.random.seed(5)
    sim.add_objs(phys.light.generate_photons_bulk(n, min=lo_e, max=hi_e, fn_vec=lambda size: np.random.beta(2.0, 5.0, size)))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
    sign = phys.light.ScatterSignMeasureStep(None, True)
    sim.add_step(3, sign)
    run(sim)
    assert sim._batch is not None and len(sign.data) == 4 and sign.data[-1][1] == n
    E = sim.download("E")
    assert E.shape == (n,) and np.array_equal(E[:2000], want)
    assert float(lo_e) <= E.min() and E.max() <= float(hi_e) and abs(E.mean() - (float(lo_e) + (float(hi_e) - float(lo_e)) * 2 / 7)) < 0.01 * float(hi_e)
    sim.close(download=False)


@pytest.mark.parametrize("spl", [None, 1])
@pytest.mark.parametrize("tag", ["iso", "del"])
def test_measure_files_are_the_reference_s_byte_for_byte(golden, tmp_path, tag, spl):
    """The CSVs MeasureStep.terminate writes (physicl/__init__.py:360-378) after the reference's seeded runs
    (tests/golden/make_golden.py g8_csv): the same text from the K-passes-per-launch schedule and from one launch per step."""
    z = golden("g8_csv")
    N, dt, passes = int(z[tag + "_N"]), float(z[tag + "_dt"]), int(z[tag + "_passes"])
    kw = dict(cl_on=True, steps_per_launch=spl)
    if tag == "iso":
        kw["exit"] = lambda s: s.t >= (passes - 0.5) * dt
    sim = phys.Simulation(**kw)
    sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i) for i in range(N)])
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(dt)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)) if tag == "iso"
                 else phys.light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
    files = [str(tmp_path / (tag + nm)) for nm in ("_sign.csv", "_meas.csv", "_meas_no_n.csv")]
    sim.add_step(3, phys.light.ScatterSignMeasureStep(files[0], True))
    sim.add_step(4, phys.light.ScatterMeasureStep(files[1], True, [np.array(p) for p in z["planes"]]))
    sim.add_step(5, phys.light.ScatterMeasureStep(files[2], False, [np.array(z["planes"][0])]))
    np.random.seed(int(z[tag + "_seed"]))
    run(sim)
    assert len(sim.ts) == passes
    for f, key in zip(files, ("_sign_csv", "_meas_csv", "_meas_no_n_csv")):
        assert open(f).read() == str(z[tag + key]), key


@pytest.mark.parametrize("spl", [None, 1])
def test_run_loop_is_the_reference_s_own(golden, spl):
    """Simulation.start() / join() of the reference itself (tests/golden/make_golden.py g9_run: physicl/__init__.py:501-524):
    steps registered out of index order run in registration order (the measure step first, at t = 0), a user Step removes an
    object in pass 2 and adds one in pass 4 through the simulation's own methods and looks at every object every pass; clock,
    rows, who is in the list and where, get_state() afterwards, the np.random stream left where the reference leaves it."""
    z = golden("g9_run")
    N, dt = int(z["N"]), float(z["dt"])
    sim = phys.Simulation(cl_on=True, exit=lambda s: s.t >= 0.0075, steps_per_launch=spl)
    sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i) for i in range(N)])
    rows = []

    class Probe(phys.Step):
        def __init__(self):
            self.passes, self.terminated_at = 0, None

        def run(self, sim):
            if self.passes == 2:
                sim.remove_obj(sim.objects[0])
            if self.passes == 4:
                sim.add_obj(phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=1000))
            rows.append((float(sim.t), float(sim.dt), [o.uid for o in sim.objects], [float(np.asarray(o.r)[0]) for o in sim.objects]))
            self.passes += 1

        def terminate(self, sim):
            self.terminated_at = float(sim.t)

    sign, probe = phys.light.ScatterSignMeasureStep(None, True), Probe()
    sim.add_step(5, sign)
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(dt)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterDeleteStep(np.double(0.0005), np.double(0.001)))
    sim.add_step(7, probe)
    # the reference means to refuse a second step on an index and trips over an undefined name doing so (NameError,
    # __init__.py:441); here the refusal is an IndexError with the reference's message
    assert str(z["duplicate_index_error"]) == "NameError"
    with pytest.raises(IndexError, match="existing index"):
        sim.add_step(7, probe)
    np.random.seed(int(z["seed"]))
    run(sim)
    assert np.array_equal(np.array(sign.data, dtype=np.float64), z["sign_rows"])
    assert np.array_equal(np.array(sim.ts, dtype=np.float64), z["ts"])
    assert np.array_equal([p[0] for p in rows], z["probe_t"]) and np.array_equal([p[1] for p in rows], z["probe_dt"])
    assert np.array_equal([len(p[2]) for p in rows], z["probe_n"])
    assert np.array_equal([u for p in rows for u in p[2]], z["probe_uids"])
    assert np.array_equal([x for p in rows for x in p[3]], z["probe_r0"])
    assert probe.terminated_at == float(z["terminated_at"])
    st = sim.get_state()
    assert sorted(st.keys()) == list(z["state_keys"])
    assert (st["objects"], float(st["t"]), float(st["dt"])) == (int(z["state_objects"]), float(z["state_t"]), float(z["state_dt"]))
    assert sim.running == bool(z["running_after"])
    assert np.random.random() == float(z["next_random"])
    sim.remove_step(7)
    assert list(sim.steps.keys()) == list(z["steps_after_remove"])


def test_user_kernel_through_clprogram_is_the_reference_s_own(golden):
    """The kernel-glue classes with a user's kernel, as the reference itself ran it (tests/golden/make_golden.py
    g10_clprogram: CLInput of every type, an int and a double CLOutput, physicl/__init__.py:543-664): the outputs of every
    pass, who is left, and the np.random stream behind the per-object draws."""
    z = golden("g10_clprogram")
    N, sigma = int(z["N"]), np.double(z["sigma"])

    class Absorber(phys.Step):
        def __init__(self):
            self.prog, self.outs = None, []

        def run(self, sim):
            if self.prog is None:
                skip = phys.CLInput(name="only_photons", type="obj_action",
                                    code="if type(obj) != physicl.light.PhotonObject:\n \t\t continue")
                d = [phys.CLInput(name="d%d" % k, type="obj", obj_attr="dr[%d]" % k) for k in range(3)]
                u = phys.CLInput(name="u", type="obj_def", obj_def="np.random.random()")
                e2 = phys.CLInput(name="e2", type="obj_def", obj_def="obj.E * 2")
                sg = phys.CLInput(name="sigma", type="const", const_value=str(sigma))
                who = phys.CLInput(name="who", type="obj_track", obj_track="obj")
                self.prog = phys.CLProgram(sim, "absorb", """
                    int gid = get_global_id(0);
                    double path = sqrt(d0[gid] * d0[gid] + d1[gid] * d1[gid] + d2[gid] * d2[gid]);
                    gone[gid] = (sigma * path >= u[gid]) ? 1 : 0;
                    depth[gid] = sigma * path + 0.25 * e2[gid];
                """)
                self.prog.prep_metadata = [skip] + d + [u, e2, who, sg]
                self.prog.output_metadata = [phys.CLOutput(name="gone", ctype="int"), phys.CLOutput(name="depth")]
                self.prog.build_kernel()
            res = self.prog.run()
            self.outs.append({k: np.array(v) for k, v in res.items()})
            for idx, x in enumerate(res["gone"]):
                if x == 1:
                    sim.remove_obj(self.prog.who[idx])

    sim = phys.Simulation(cl_on=True, exit=lambda s: s.t >= 0.0025)
    E = z["init_E"]
    objs = [phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(e), uid=i)
            for i, e in enumerate(np.delete(E, 10))]
    objs.insert(10, phys.Object(v=phys.Measurement([5.0, 0, 0], "m**1 s**-1"), uid=-1))
    sim.add_objs(objs)
    ab = Absorber()
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, ab)
    np.random.seed(int(z["seed"]))
    run(sim)
    assert len(ab.outs) == int(z["passes"])
    for k, o in enumerate(ab.outs):
        assert sorted(o.keys()) == list(z["k%d_keys" % k])
        assert o["gone"].dtype == z["k%d_gone" % k].dtype and np.array_equal(o["gone"], z["k%d_gone" % k])
        assert o["depth"].dtype == np.float64 and np.array_equal(o["depth"], z["k%d_depth" % k])      # *, +, sqrt: exact
    assert np.array_equal([o.uid for o in sim.objects], z["survivor_uid"])
    assert np.random.random() == float(z["next_random"])


@pytest.mark.parametrize("spl", [None, 1])
def test_mixed_loop_is_the_reference_s_own(golden, spl):
    """BASELINE configs[4]'s loop as the reference ran it (tests/golden/make_golden.py g11_mixed): one np.random stream feeds
    the isotropic step (three draws per photon) and the delete step (one) of every pass; rows of both measures, who is left,
    where, and the stream afterwards."""
    z = golden("g11_mixed")
    N, dt, K = int(z["N"]), float(z["dt"]), int(z["K"])
    sim = phys.Simulation(cl_on=True, steps_per_launch=spl, exit=lambda s: s.t >= (K - 0.5) * dt)
    sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i) for i in range(N)])
    sign = phys.light.ScatterSignMeasureStep(None, True)
    meas = phys.light.ScatterMeasureStep(None, True, [np.array(p) for p in z["planes"]])
    for i, st in enumerate([phys.UpdateTimeStep(lambda s: np.double(dt)), phys.newton.NewtonianKinematicsStep(),
                            phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)), sign,
                            phys.newton.NewtonianKinematicsStep(), phys.light.ScatterDeleteStep(np.double(0.0004), np.double(0.001)), meas]):
        sim.add_step(i, st)
    np.random.seed(int(z["seed"]))
    run(sim)
    # (numpy's stream is drawn on the host in the reference's order and uploaded: such a run takes one launch per light step
    # whatever steps_per_launch says -- the K-passes-per-launch kernels draw on the device)
    assert not sim.schedule.get("mixed_multi")
    assert np.array_equal(np.array(sign.data, dtype=np.float64), z["sign_rows"])
    assert np.array_equal(np.array(meas.data, dtype=np.float64), z["measure_rows"])
    assert np.array_equal([o.uid for o in sim.objects], z["final_uid"])
    v = np.array([np.asarray(o.v) for o in sim.objects])
    r = np.array([np.asarray(o.r) for o in sim.objects])
    assert np.max(np.abs(v - z["final_v"])) <= V_ABS_TOL
    assert np.max(np.abs(r - z["final_r"])) <= 2 * K * V_ABS_TOL * dt + 4 * np.spacing(np.abs(r).max())
    assert np.random.random() == float(z["next_random"])


ISO_API = {                                         # the step arguments make_golden.py's g2_iso gave the reference, tag by tag
    "lambda": dict(n=0.000000000000001, A=0.0000000000000000001, wavelength_dep_scattering=True),
    "varn": dict(n=0.000000000000001, A=0.0000000000000000001, wavelength_dep_scattering=True, variable_n=True,
                 variable_n_fn="0.000000001 * exp(r0[gid] - 5)"),
    "varn_radial": dict(n=0.5, A=123.0, variable_n=True,
                        variable_n_fn="2.5 * exp(-1 * (sqrt(pow(r0[gid], 2) + pow(r1[gid], 2) + pow(r2[gid], 2)) - 6.0)/(3.5))"),
    "varn_overflow": dict(n=0.000000000000001, A=0.0000000000000000001, wavelength_dep_scattering=True, variable_n=True,
                          variable_n_fn="0.000000001 * exp(r0[gid] - 5)"),
}


@pytest.mark.parametrize("fuse", [True, False])
@pytest.mark.parametrize("tag", sorted(ISO_API))
def test_seeded_variable_n_runs_reproduce_the_reference(golden, tag, fuse):
    """The wavelength term and variable_n_fn through the plugin API's ScatterIsotropicStep (A and n reach the kernel swapped,
    light.py:287; with variable_n the user's A is not used at all, light.py:299) after np.random.seed: the reference's measure
    rows and its final state.  A photon whose pcoll lies within 1e-14 of its draw may decide the other way (exp / pow of
    another maths library): none does in these fixtures, the rows are compared exactly."""
    z = golden("g2_iso_" + tag)
    N, K, dt = len(z["k0_rand"]), int(z["K"]), float(z["dt"])
    sim = phys.Simulation(cl_on=True, exit=lambda s: s.t >= (K - 0.5) * dt, fuse=fuse)
    objs = []
    for i in range(N):
        p = phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(z["init_E"][i]), uid=i)
        p.r = phys.Measurement(z["init_r"][i], "m**1")
        objs.append(p)
    sim.add_objs(objs)
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(dt)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterIsotropicStep(**ISO_API[tag]))
    sign = phys.light.ScatterSignMeasureStep(None, True)
    meas = phys.light.ScatterMeasureStep(None, True, [np.array(p) for p in z["planes"]])
    sim.add_step(3, sign)
    sim.add_step(4, meas)
    np.random.seed(int(z["seed"]))
    run(sim)
    assert np.array_equal(np.array(sign.data, dtype=np.float64), z["sign_rows"])
    assert np.array_equal(np.array(meas.data, dtype=np.float64).reshape(z["measure_rows"].shape), z["measure_rows"])
    last = "k%d_post_" % (K - 1)
    v = np.array([np.asarray(o.v) for o in sim.objects])
    r = np.array([np.asarray(o.r) for o in sim.objects])
    assert np.max(np.abs(v - z[last + "v"])) <= V_ABS_TOL
    assert np.max(np.abs(r - z[last + "r"])) <= K * V_ABS_TOL * dt + 4 * np.spacing(np.abs(z[last + "r"]).max())
    assert sim.hits == (~np.isnan(z["k%d_res0" % (K - 1)])).sum()


@pytest.mark.parametrize("spl", [None, 1])
@pytest.mark.parametrize("tag", ["iso", "del"])
def test_plain_objects_among_the_photons_as_the_reference_treats_them(golden, tag, spl):
    """g12_kinds: every seventh object is a plain Object.  The light steps skip it without drawing a random number
    (light.py:233, 283), Newton moves it, both measures count it: the reference's rows, survivors, final state and random stream."""
    z = golden("g12_kinds")
    N, dt = int(z[tag + "_N"]), float(z[tag + "_dt"])
    sim = phys.Simulation(cl_on=True, steps_per_launch=spl, exit=lambda s: s.t >= 5.5 * dt)
    objs = []
    for i in range(N):
        if z[tag + "_is_obj"][i]:
            objs.append(phys.Object(v=phys.Measurement(np.array([5.0, -3.0, 2.0]) * (1 + i % 3), "m**1 s**-1"), uid=i))
        else:
            objs.append(phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i))
    sim.add_objs(objs)
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(dt)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)) if tag == "iso"
                 else phys.light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
    sign = phys.light.ScatterSignMeasureStep(None, True)
    meas = phys.light.ScatterMeasureStep(None, True, [np.array(p) for p in z["planes"]])
    sim.add_step(3, sign)
    sim.add_step(4, meas)
    np.random.seed(int(z[tag + "_seed"]))
    run(sim)
    assert np.array_equal(np.array(sign.data, dtype=np.float64), z[tag + "_sign_rows"])
    assert np.array_equal(np.array(meas.data, dtype=np.float64), z[tag + "_measure_rows"])
    assert np.array_equal([o.uid for o in sim.objects], z[tag + "_final_uid"])
    v = np.array([np.asarray(o.v) for o in sim.objects])
    r = np.array([np.asarray(o.r) for o in sim.objects])
    assert np.max(np.abs(v - z[tag + "_final_v"])) <= V_ABS_TOL
    assert np.max(np.abs(r - z[tag + "_final_r"])) <= 6 * V_ABS_TOL * dt + 4 * np.spacing(np.abs(r).max())
    plain = np.array([type(o) is phys.Object for o in sim.objects])
    assert plain.sum() == z[tag + "_is_obj"].sum() and np.array_equal(r[plain], z[tag + "_final_r"][plain])    # Euler: exact
    assert np.random.random() == float(z[tag + "_next_random"])


def _edge_sim(z, iso=False, **kw):
    N, dt = int(z["N"]), float(z["dt"])
    sim = phys.Simulation(cl_on=True, exit=lambda s: s.t >= 6.5 * dt, **kw)
    sim.add_objs([phys.light.PhotonObject(v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1.0), uid=i) for i in range(N)])
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(dt)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.0003), n=np.double(0.001)) if iso
                 else phys.light.ScatterDeleteStep(np.double(0.0003), np.double(0.001)))
    meas = phys.light.ScatterMeasureStep(None, True, [np.array(p) for p in z["planes"]])
    sim.add_step(3, meas)
    return sim, meas


def test_planes_exactly_where_photons_stop_count_as_in_the_reference(golden):
    """g13_plane_edges: planes at x = k * fl(c dt).  ``r - dr <= loc <= r`` includes both ends (light.py:385-399): a photon that
    stops ON a plane is counted when it arrives and, where fl(r - dr) gives the plane back, again when it leaves.  The
    reference's rows from the seeded run; and, on the device's own random stream, the K-passes-per-launch kernels' rows ==
    the one-launch-per-step rows on the same planes."""
    z = golden("g13_plane_edges")
    sim, meas = _edge_sim(z)
    np.random.seed(int(z["seed"]))
    run(sim)
    rows = np.array(meas.data, dtype=np.float64)
    assert np.array_equal(rows, z["measure_rows"])
    assert np.array_equal([o.uid for o in sim.objects], z["final_uid"])
    assert (rows[:2, 2] == rows[:2, 1]).all() and rows[2, 2] == 0          # plane 1: on arrival (pass 1) and on leaving (pass 2)
    for iso, sched in ((False, "fused_delete_multi"), (True, "fused_multi")):
        out = []
        for spl in (None, 1):
            s, m = _edge_sim(z, iso=iso, rng="philox", seed=5, steps_per_launch=spl)
            run(s)
            out.append((np.array(m.data, dtype=np.float64), dict(s.schedule)))
        assert out[0][1].get(sched, 0) >= 1 and not out[1][1].get(sched)
        assert np.array_equal(out[0][0], out[1][0])
        if not iso:
            assert (out[0][0][:2, 2] == out[0][0][:2, 1]).all() and out[0][0][2, 2] == 0
        else:                                                              # most photons still fly straight: both passes see them on plane 1
            assert out[0][0][0, 2] == out[0][0][0, 1] and 0 < out[0][0][1, 2] < out[0][0][1, 1]
