"""GPU: ``Simulation(devices=[...])`` -- several library contexts inside ONE process, particles sharded by index,
counters summed on the host (physicl_amd/multidev.py).  The reference is one process with one simulation thread
(physicl/__init__.py:400-432, 501-524); this is how a script written against it uses a node's GPUs without a launcher.

On a one-GPU test box the contexts share device 0 (``devices=[0, 0]``, ``[0, 0, 0]``; with more GPUs visible they go to
distinct devices, ``_device_lists``): the sharding, the fan-out and
the host-side reduction are exactly those of N GPUs.  Everything observable must equal the single-context run: ``ts``,
every measure row, ``hits``, what host plugins saw, ``len(sim.objects)``, the position of the global ``np.random``
stream, and the final state of every object bit for bit -- for the randomly drawn simulations of
tests/test_gpu_random_simulations.py (explicit objects and bulk batches, host plugins that read, edit and remove
objects, host-drawn and device randoms, every launch formulation)."""
import numpy as np
import pytest

import physicl as phys
import physicl.light
import physicl.newton

from test_gpu_random_simulations import assert_same, build_and_run, draw_config

pytestmark = pytest.mark.gpu


def _device_lists():
    """The contexts of these tests go to DISTINCT devices as soon as the box has them ([0, 1], [0, 1, 2]); on a one-GPU box
    they share device 0.  Counting devices does not initialise the GPU (pcl_device_count -> hipGetDeviceCount)."""
    from physicl_amd import _hip
    n = max(1, _hip.device_count())
    return [i % n for i in range(2)], [i % n for i in range(3)]


DEV2, DEV3 = _device_lists()


@pytest.mark.parametrize("seed", range(24))
def test_sharded_in_process_equals_one_context(seed):
    cfg = draw_config(np.random.RandomState(500 + seed))
    one = build_and_run(cfg, "default", True)
    two = build_and_run(cfg, "default", True, devices=DEV2)
    assert_same(one, two, ("devices=%s" % DEV2, cfg))
    assert one["schedule"] == two["schedule"]                  # the same launch formulations were chosen
    if seed % 3 == 0:
        three = build_and_run(cfg, 1, True, devices=DEV3)
        assert_same(one, three, ("devices=%s, one launch per light step" % DEV3, cfg))


def test_bulk_run_on_two_contexts_rows_counts_and_state():
    """configs[2]-style bulk run and a delete-until-empty run, 200k photons over two contexts: rows, counts while
    running (get_state from the main thread) and downloaded state equal the one-context run."""
    def run(**kw):
        out = {}
        sim = phys.Simulation(seed=5, exit=lambda s: len(s.ts) >= 40, **kw)
        sim.add_objs(phys.light.generate_photons_bulk(200_001, min=phys.light.E_from_wavelength(700e-9),
                                                      max=phys.light.E_from_wavelength(200e-9), seed=5))
        sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(1e-9)))
        sim.add_step(1, phys.newton.NewtonianKinematicsStep())
        sim.add_step(2, phys.light.ScatterIsotropicStep(n=1e-15, A=1e-19, wavelength_dep_scattering=True, variable_n=True,
                                                        variable_n_fn="0.000000001 * exp(r0[gid] - 5)"))
        m = phys.light.ScatterSignMeasureStep(None, True)
        sim.add_step(3, m)
        sim.start()
        sim.join()
        assert sim.error is None, sim.error
        out["iso"] = ([list(map(float, r)) for r in m.data], sim.hits, len(sim.objects), dict(sim.schedule),
                      {f: sim.download(f) for f in ("r", "v", "dr", "dv", "E", "id")})
        sim.close(download=False)
        sim = phys.Simulation(seed=5, **kw)                                   # default exit: no objects left
        sim.add_objs(phys.light.generate_photons_bulk(200_001, min=1.0, max=1.0, seed=5))
        sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
        sim.add_step(1, phys.newton.NewtonianKinematicsStep())
        sim.add_step(2, phys.light.ScatterDeleteStep(np.double(0.001), np.double(0.001)))
        m = phys.light.ScatterMeasureStep(None, True, [[1.0e6, np.nan, np.nan]])
        sim.add_step(3, m)
        sim.start()
        sim.join()
        assert sim.error is None, sim.error
        out["del"] = ([list(map(float, r)) for r in m.data], len(sim.ts), len(sim.objects), dict(sim.schedule))
        sim.close(download=False)
        return out
    a, b = run(), run(devices=DEV2)
    assert a["iso"][:4] == b["iso"][:4] and a["iso"][3] == {"fused_multi": 2}
    for f in a["iso"][4]:
        assert np.array_equal(a["iso"][4][f], b["iso"][4][f]), f
    assert a["del"] == b["del"] and a["del"][2] == 0 and a["del"][3].get("fused_delete_multi", 0) >= 1


def test_devices_and_comm_exclude_each_other():
    from physicl_amd.dist import CounterComm
    with pytest.raises(ValueError):
        phys.Simulation(devices=DEV2, comm=CounterComm(0, 1, "gloo"))


def test_device_group_of_the_c_abi_equals_multidevice_and_one_device():
    """``pcl_group_*`` (the shim owns the contexts and their worker threads) against ``MultiDevice`` (Python fan-out) and
    one ``Device``: K-step scatter rows, K-body delete rows, a mixed launch, windows of ids and positions."""
    from physicl_amd import _hip as hip
    from physicl_amd.multidev import MultiDevice
    N, seed = 120_003, 9
    C, H = 299792458.0, 6.62607015e-34
    sc = dict(A=1e-3, n=1e-3, flags=0, c=C, h=H, seed=seed, step=3)
    res = {}
    for name, make in (("one", lambda: hip.Device(0)), ("multidev", lambda: MultiDevice(DEV3)), ("group", lambda: hip.DeviceGroup(DEV3))):
        with make() as d:
            d.store_alloc(N)
            d.fill_photons(N, 50, C, 2.8e-19, 9.9e-19, seed)
            if name == "group":
                a = d.step_fused_multi(1e-3, 5, sc)
                b = d.step_mixed_multi(1e-3, 3, ("iso", "delete"), dict(sc, step=0), (1e-3, 0.4e-3), seed, 20)
                c = d.step_fused_delete_multi(1e-3, 4, 1e-3, 0.5e-3, seed, 40, [[1e6, np.nan, np.nan]])
                o = d.step_fused_delete(1e-3, 1e-3, 0.5e-3, seed, 60, [[1e6, np.nan, np.nan]])
                assert d.shard(N, 1) == (N // 3, 2 * N // 3)
            else:
                a = d.step_fused_multi(1e-3, 5, sc, (), True, True)
                b = d.step_mixed_multi(1e-3, 3, ("iso", "delete"), dict(sc, step=0), (1e-3, 0.4e-3), (), seed, 20, True)
                c = d.step_fused_delete_multi(1e-3, 4, 1e-3, 0.5e-3, seed, 40, [[1e6, np.nan, np.nan]], True)
                od = d.step_fused_delete(1e-3, 1e-3, 0.5e-3, hip.RNG_PHILOX, seed, 60, [[1e6, np.nan, np.nan]], lazy=True)
                o = np.array([od["N"]] + [int(x) for x in od["sign"]] + [int(od["planes"][0]), od["removed"]])
            n = d.count
            res[name] = (np.asarray(a), np.asarray(b), np.asarray(c), np.asarray(o), n, d.download_ids(), d.download(hip.R0),
                         d.download(hip.V1, 1000, n // 2))
    for name in ("multidev", "group"):
        for x, y in zip(res["one"], res[name]):
            assert np.array_equal(x, y), name
    assert 0 < res["one"][4] < N


def test_cl_off_semantics_on_two_contexts_equal_one_context():
    """``Simulation(cl_on=False, devices=DEV2)``: the reference's CPU-path semantics (data-dependent np.random order of
    ScatterIsotropicStep.__run_py, physicl/light.py:335-350; the skip-after-removal iteration of
    ScatterDeleteStepReference.__run_py, light.py:216-223) on a store sharded over two contexts: same decisions, same
    state, same position of the np.random stream as on one context."""
    def run(**kw):
        np.random.seed(99)
        rs = np.random.RandomState(3)
        sim = phys.Simulation(cl_on=False, exit=lambda s: len(s.ts) >= 6, **kw)
        objs = []
        for i in range(3001):
            d = np.zeros(3)
            d[rs.randint(3)] = rs.choice([-1.0, 1.0])
            if i % 7 == 3:
                objs.append(phys.Object(v=phys.Measurement(rs.normal(size=3), "m**1 s**-1"), uid=i))
            else:
                objs.append(phys.light.PhotonObject(v=d * 299792458.0, E=np.double(rs.uniform(2.8e-19, 9.9e-19)), uid=i))
        sim.add_objs(objs)
        sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
        sim.add_step(1, phys.newton.NewtonianKinematicsStep())
        sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
        sg = phys.light.ScatterSignMeasureStep(None, True)
        sim.add_step(3, sg)
        sim.add_step(4, phys.light.ScatterDeleteStepReference(np.double(0.001), np.double(0.0004)))
        sim.start()
        sim.join()
        assert sim.error is None, sim.error
        left = list(sim.objects)
        out = ([[float(x) for x in r] for r in sg.data], [o.uid for o in left],
               np.array([np.asarray(o.r, dtype=float) for o in left]), np.array([np.asarray(o.v, dtype=float) for o in left]),
               float(np.random.random_sample()))
        sim.close()
        return out
    a, b = run(), run(devices=DEV2)
    assert a[0] == b[0] and a[1] == b[1] and a[4] == b[4] and 0 < len(a[1]) < 3001
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])


def test_group_downloads_in_the_stores_own_precision_and_windows_that_start_in_a_later_shard():
    """``DeviceGroup.download`` allocated float64 whatever the store held; a window that starts beyond the first
    shard's capacity used to hand that shard an offset it refuses (Python MultiDevice); shards are sized for any count up to
    the capacity."""
    from physicl_amd import _hip as hip
    from physicl_amd.multidev import MultiDevice
    N = 90_001
    with hip.DeviceGroup(DEV3) as g:
        g.store_alloc(N, "f32")
        g.fill_photons(N, 0, 299792458.0, 1.0, 2.0, 3)
        e = g.download(hip.E)
        assert e.dtype == np.float32 and len(e) == N and np.all((e >= 1.0) & (e <= 2.0))
        assert g.download(hip.E, dtype=np.float32).dtype == np.float32
    with MultiDevice(DEV3) as md:
        md.store_alloc(N)
        md.fill_photons(N, 10, 299792458.0, 1.0, 2.0, 3)
        tail = md.download_ids(100, N - 100)                       # the window lies in the last shard only
        assert np.array_equal(tail, np.arange(N - 100 + 10, N + 10))
        md.fill_photons(N - 7, 10, 299792458.0, 1.0, 2.0, 3)      # fewer photons into the same store: every shard has room
        assert md.count == N - 7 and md.capacity >= N
