"""GPU parity of pcl_step_fused_multi: K fused steps in one pass over the store.

Bars: bit-identical -- state (r, v, dr, dv, E) and every per-step counter -- to K calls of the single-step
lazy fused path (itself pinned to the reference / the oracle in test_gpu_parity.py), for fp64 and fp32,
every scatter variant and ragged sizes; and, against the CPU oracle run K steps on its own, equal hit /
sign counters per step with positions within K * dt * 4 ulp(c) (the oracle's libm sin/cos differ from
OCML's by <= 4 ulp of c in a scattered velocity, which then feeds the later Euler steps).
"""
import os

import numpy as np
import pytest

from oracle import physicl_oracle as orc

pytestmark = pytest.mark.gpu

C_LIT = 299792458.0
H_LIT = 6.62607015e-34
V_ABS_TOL = 4 * np.spacing(C_LIT)
EXPR_EX = "0.000000001 * exp(r0[gid] - 5)"
EXPR_RAD = "2.5 * exp(-1 * (sqrt(pow(r0[gid], 2) + pow(r1[gid], 2) + pow(r2[gid], 2)) - 6.0)/(3.5))"
CASES = {
    # tag: (use_E, expr, A, n, dt)
    "base": (False, None, 1e-3, 1e-3, 1e-3),
    "lambda": (True, None, 1e-15, 1e-19, 5e-3),
    "varn": (True, EXPR_EX, 1e-15, 1e-19, 1e-9),
    "varn_radial": (False, EXPR_RAD, 1e-9, 1.0, 1e-9),
}


@pytest.fixture(scope="module")
def hip():
    from physicl_amd import _hip
    return _hip


@pytest.fixture()
def make_store(hip):
    devs = []

    def make(capacity, dtype="f64"):
        d = hip.Device(0)
        d.store_alloc(capacity, dtype)
        devs.append(d)
        return d
    yield make
    for d in devs:
        d.close()


def scatter_dict(hip, tag, seed, step):
    use_e, expr, A, n, dt = CASES[tag]
    flags = (hip.SCATTER_WAVELENGTH if use_e else 0) | (hip.SCATTER_VARIABLE_N if expr else 0)
    return dict(A=A, n=n, flags=flags, c=C_LIT, h=H_LIT, n_expr=expr, rng_mode=hip.RNG_PHILOX, seed=seed, step=step), dt


def initial(N, dtype, seed):
    rs = np.random.RandomState(seed)
    npdt = np.float64 if dtype == "f64" else np.float32
    return {"r": rs.uniform(-8, 8, (N, 3)).astype(npdt), "v": np.tile([C_LIT, 0.0, 0.0], (N, 1)).astype(npdt),
            "E": rs.uniform(2.8e-19, 9.9e-19, N).astype(npdt), "id_base": 7_000_000_001}


def state_equal(a, b):
    assert np.array_equal(a["E"], b["E"])
    for f in ("r", "v", "dr", "dv"):
        for k in range(3):
            assert np.array_equal(a[f][k], b[f][k]), (f, k)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("tag", sorted(CASES))
@pytest.mark.parametrize("N,K", [(1, 1), (2, 3), (777, 5), (2049, 2), (200_001, 7), (65_537, 64)])
def test_multi_is_bit_identical_to_single_steps(make_store, hip, tag, N, K, dtype):
    init = initial(N, dtype, N + K)
    seed, step0 = 0xC0FFEE1234, 11
    # measure planes (light.py:385-399) ride along in half of the cases; the single steps then take the generic kernel
    planes = [[0.5, np.nan, np.nan], [np.nan, np.nan, -2.0], [np.nan, 1e-3, np.nan]] if N in (777, 200_001) else []
    a = make_store(N, dtype)
    a.upload_state(init)
    single = []
    for k in range(K):
        sc, dt = scatter_dict(hip, tag, seed, step0 + k)
        o = a.step_fused(dt, sc, planes, lazy=True)
        single.append((o["N"], o["hits"], list(o["sign"]), list(o["planes"])))
    b = make_store(N, dtype)
    b.upload_state(init)
    sc, dt = scatter_dict(hip, tag, seed, step0)
    rows = b.step_fused_multi(dt, K, sc, planes)
    assert [(o["N"], o["hits"], list(o["sign"]), list(o["planes"])) for o in rows] == single
    if planes and tag == "base":
        assert sum(sum(o["planes"]) for o in rows) > 0
    assert b.last_scatter_hits() == single[-1][1]
    state_equal(a.download_state(), b.download_state())         # includes the implicit dr / dv of the last step


def test_multi_chains_with_single_steps_and_consumers(make_store, hip):
    """multi -> single lazy step -> multi -> delete: the implicit dr/dv hand-over works in every direction."""
    N, tag = 30_011, "varn"
    init = initial(N, "f64", 5)
    seed = 99
    out = []
    for use_multi in (False, True):
        d = make_store(N)
        d.upload_state(init)
        log, step = [], 0
        for chunk in (4, 1, 3):
            sc, dt = scatter_dict(hip, tag, seed, step)
            if use_multi and chunk > 1:
                log += [(o["hits"], list(o["sign"])) for o in d.step_fused_multi(dt, chunk, sc)]
            else:
                for k in range(chunk):
                    sc, dt = scatter_dict(hip, tag, seed, step + k)
                    o = d.step_fused(dt, sc, [], lazy=True)
                    log.append((o["hits"], list(o["sign"])))
            step += chunk
        mid = d.download_state()
        alive, removed = d.step_scatter_delete(1e-3, 1e-3, hip.RNG_PHILOX, seed, 1000)   # reads the real dr
        out.append((log, mid, alive, removed, d.download_state()))
    assert out[0][0] == out[1][0] and out[0][2:4] == out[1][2:4]
    state_equal(out[0][1], out[1][1])
    state_equal(out[0][4], out[1][4])
    assert np.array_equal(out[0][4]["id"], out[1][4]["id"])


@pytest.mark.parametrize("tag", ["base", "varn"])
def test_multi_vs_oracle_chain(make_store, hip, tag):
    N, K = 1000, 6
    init = initial(N, "f64", 42)
    use_e, expr, A, n, dt = CASES[tag]
    st = {"r": [np.ascontiguousarray(init["r"][:, k]) for k in range(3)],
          "v": [np.ascontiguousarray(init["v"][:, k]) for k in range(3)],
          "dr": [np.zeros(N)] * 3, "dv": [np.zeros(N)] * 3, "E": init["E"].copy(),
          "id": np.arange(N, dtype=np.int64) + init["id_base"]}
    seed, step0 = 2024, 3
    ref = []
    for k in range(K):
        orc.step_newton(st, dt)
        hit = orc.step_scatter_isotropic(st, orc.philox_draws(seed, step0 + k, st["id"]), A, n, C_LIT, h=H_LIT,
                                         use_E=use_e, n_expr=expr)
        ref.append((int(hit.sum()), [int((st["v"][j] > 0).sum()) for j in range(3)]))
    d = make_store(N)
    d.upload_state(init)
    sc, _ = scatter_dict(hip, tag, seed, step0)
    rows = d.step_fused_multi(dt, K, sc)
    assert [(o["hits"], list(o["sign"])) for o in rows] == ref
    s = d.download_state()
    assert np.max(np.abs(np.stack(s["v"], 1) - np.stack(st["v"], 1))) <= V_ABS_TOL
    assert np.max(np.abs(np.stack(s["r"], 1) - np.stack(st["r"], 1))) <= K * dt * V_ABS_TOL + 1e-15


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("tag", sorted(CASES))
@pytest.mark.parametrize("N,K,with_planes", [(1, 1, False), (255, 2, True), (257, 7, False), (511, 33, False), (513, 3, True), (2049, 9, False),
                                             (100_003, 5, True), (33_333, 64, False)])
def test_the_k_step_pass_directly_vs_the_oracle_chain(make_store, hip, tag, N, K, with_planes, dtype):
    """The K-step kernel (256 photons per wave, velocities in LDS) against the ORACLE's own chain of K steps -- not against
    another device kernel -- over the shapes its geometry has edges at: fewer photons than a lane group, one short of / one over
    a wave's 256 and 512, a ragged last tile, K = 1, odd K (a Philox decision block split between launches), K above a launch's
    usual 32 and the maximum 64; every scatter variant; fp64 and fp32 (the oracle's float32 restatement); with and without
    measure planes.  Hit and sign counters per step exact (decisions are compares of IEEE products against the same Philox
    uniforms; an exp / pow rounding tie would show as a one-photon difference and none has), plane crossings exact on the
    device's own positions' bound, velocities within 4 ulp of c, positions within the bound that follows from it."""
    use_e, expr, A, n, dt = CASES[tag]
    np_t = np.float64 if dtype == "f64" else np.float32
    init = initial(N, dtype, 1000 + N + K)
    st = {"r": [np.ascontiguousarray(init["r"][:, k]) for k in range(3)], "v": [np.ascontiguousarray(init["v"][:, k]) for k in range(3)],
          "dr": [np.zeros(N, np_t)] * 3, "dv": [np.zeros(N, np_t)] * 3, "E": init["E"].copy(),
          "id": np.arange(N, dtype=np.int64) + init["id_base"]}
    seed, step0 = 77, 5                                   # (odd first step: the first launch starts inside a decision block)
    planes = [[0.5, np.nan, np.nan], [np.nan, np.nan, -2.0]] if with_planes else []
    ref = []
    for k in range(K):
        orc.step_newton(st, dt, np_t)
        hit = orc.step_scatter_isotropic(st, orc.philox_draws(seed, step0 + k, st["id"], np_t), A, n, C_LIT, h=H_LIT, use_E=use_e, n_expr=expr,
                                         dtype=np_t)
        ref.append((int(hit.sum()), [int((st["v"][j] > 0).sum()) for j in range(3)]))
    d = make_store(N, dtype)
    d.upload_state(init)
    sc, _ = scatter_dict(hip, tag, seed, step0)
    rows = d.step_fused_multi(dt, K, sc, planes)
    got = [(o["hits"], list(o["sign"])) for o in rows]
    if dtype == "f64":
        assert got == ref
    else:       # fp32: exp / pow of OCML and of numpy's float32 may round a tie differently: at most a photon here and there
        assert all(abs(g[0] - r[0]) <= max(2, N // 20000) and max(abs(a - b) for a, b in zip(g[1], r[1])) <= max(2, N // 20000) for g, r in zip(got, ref))
    # (PCL_MULTI_NQ2=0 selects the 128-photon instantiation, which exists for the hipRTC specialisations in fp64)
    rtc64 = dtype == "f64" and expr is not None
    want = 128 if os.environ.get("PCL_MULTI_NQ2") == "0" and rtc64 else (192 if os.environ.get("PCL_MULTI_NQ3") == "1" and rtc64 else 256)
    if dtype == "f64" and expr is None and os.environ.get("PCL_MULTI_NQ2") != "0":
        # constant n, ahead-of-time kernels: 192 photons per wave (k_multi3) when forced, or by itself where the loop's hit
        # probability A n c dt -- known before the first launch without the wavelength term -- lies in [0.25, 0.333)
        nq3 = os.environ.get("PCL_MULTI_NQ3")
        want = 192 if nq3 == "1" or (nq3 is None and not use_e and 0.25 <= A * n * C_LIT * dt < 0.333) else 256
    assert d.last_multi_work()[2] == want
    s = d.download_state()
    tol_v = 4 * float(np.spacing(np_t(C_LIT)))
    if got == ref:                                         # same decisions: the states are comparable photon by photon
        assert np.max(np.abs(np.stack(s["v"], 1).astype(np.float64) - np.stack(st["v"], 1).astype(np.float64))) <= tol_v
        r_ref = np.stack(st["r"], 1).astype(np.float64)
        slack = K * float(np.spacing(np_t(max(1.0, np.max(np.abs(r_ref))))))
        assert np.max(np.abs(np.stack(s["r"], 1).astype(np.float64) - r_ref)) <= K * dt * tol_v + slack + 1e-15
    if planes:
        assert all(len(o["planes"]) == 2 for o in rows)


def test_multi_state_errors(make_store, hip):
    d = make_store(100)
    d.upload_state({"v": np.ones((100, 3)), "E": np.ones(100)})
    sc, dt = scatter_dict(hip, "base", 1, 0)
    with pytest.raises(hip.HipError):
        d.step_fused_multi(dt, 0, sc)
    with pytest.raises(hip.HipError):
        d.step_fused_multi(dt, 65, sc)
    d.set_count(0, 0)                                              # an empty store: K rows of zeros, nothing launched
    rows = d.step_fused_multi(dt, 3, sc, [[0.0, np.nan, np.nan]])
    assert [(o["N"], o["hits"], list(o["sign"]), list(o["planes"])) for o in rows] == [(0, 0, [0, 0, 0], [0])] * 3
    d.set_count(100, 0)
    d.upload_kind(np.zeros(100, dtype=np.uint8))                   # a store with plain Objects: not eligible
    with pytest.raises(hip.HipError):
        d.step_fused_multi(dt, 2, sc)


# ============================================================================ K delete loop bodies per pass
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("N,K", [(1, 1), (64, 3), (2049, 2), (100_003, 9), (30_011, 12), (70_001, 13), (500_003, 40)])
def test_delete_multi_is_bit_identical_to_single_fused_delete_steps(make_store, hip, N, K, dtype):
    """pcl_step_fused_delete_multi == K x pcl_step_fused_delete (lazy, Philox): per-step rows (alive, sign counts,
    plane crossings, removed) and the survivors' whole state, ids and kinds included; mixed kinds, a previous
    compaction (explicit ids) and an odd first launch number on the way."""
    npdt = np.float64 if dtype == "f64" else np.float32
    rs = np.random.RandomState(N + K)
    vdir = rs.normal(size=(N, 3))
    vdir /= np.linalg.norm(vdir, axis=1)[:, None]
    init = {"r": rs.uniform(-3, 3, (N, 3)).astype(npdt), "v": (vdir * C_LIT).astype(npdt),
            "dv": rs.normal(size=(N, 3)).astype(npdt), "E": rs.uniform(1, 2, N).astype(npdt), "id_base": 123,
            "kind": (rs.random_sample(N) < 0.9).astype(np.uint8)}
    A, n, dt, seed, step0 = 1e-3, 0.4e-3, 1e-3, 777, 5          # pcoll ~ 0.12 per step
    planes = [[0.5, np.nan, np.nan], [np.nan, -1.0, np.nan]]
    out = []
    for multi in (False, True):
        d = make_store(N, dtype)
        d.upload_state(init)
        first = d.step_fused_delete(dt, A, n, hip.RNG_PHILOX, seed, step0 - 1, planes, lazy=True)   # ids become explicit
        if multi:
            rows = d.step_fused_delete_multi(dt, K, A, n, seed, step0, planes)
        else:
            rows = [d.step_fused_delete(dt, A, n, hip.RNG_PHILOX, seed, step0 + k, planes, lazy=True) for k in range(K)]
        log = [(o["N"], o["removed"], list(o["sign"]), list(o["planes"])) for o in rows]
        st = d.download_state() if d.count else None
        kind = d.download_kind(d.count) if d.count else None
        out.append((first["N"], log, d.count, st, kind))
    a, b = out
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
    if a[3] is not None:
        state_equal(a[3], b[3])
        assert np.array_equal(a[3]["id"], b[3]["id"]) and np.array_equal(a[4], b[4])
        assert np.all(np.diff(a[3]["id"]) > 0)


def test_delete_multi_until_empty_and_without_counters(make_store, hip):
    N = 20_000
    d = make_store(N)
    d.upload_state({"v": np.tile([C_LIT, 0.0, 0.0], (N, 1)), "E": np.ones(N)})
    rows = d.step_fused_delete_multi(1e-3, 64, 1e-3, 1e-3, seed=3, step=0, planes=None)     # survival 0.70 per step
    alive = [o["N"] for o in rows]
    assert alive[-1] == 0 and d.count == 0 and all(x >= y for x, y in zip(alive, alive[1:]))
    assert sum(o["removed"] for o in rows) == N and all(o["sign"].sum() == 0 for o in rows)
    assert abs(alive[0] - 0.7002 * N) < 6 * np.sqrt(N * 0.21)
    rows = d.step_fused_delete_multi(1e-3, 3, 1e-3, 1e-3, seed=3, step=64)                   # empty store: all zeros
    assert all(o["N"] == 0 and o["removed"] == 0 for o in rows)


def test_full_size_multi_equals_single_steps_at_1e8(make_store, hip):
    """BASELINE configs[2] size: 1e8 photons filled on the device, the bench's expression; 6 steps in one pass give the
    counters of 6 single launches, and the far ends of the store hold the same positions / velocities / implicit dv."""
    N, K, seed = 100_000_000, 6, 1234
    sc = lambda k: dict(A=1e-15, n=1e-19, flags=hip.SCATTER_WAVELENGTH | hip.SCATTER_VARIABLE_N, c=C_LIT, h=H_LIT,
                        n_expr=EXPR_EX, rng_mode=hip.RNG_PHILOX, seed=seed, step=k)
    out = []
    for multi in (True, False):
        d = make_store(N)
        d.fill_photons(N, 0, C_LIT, H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9, seed)
        if multi:
            rows = d.step_fused_multi(5e-3, K, sc(3))
        else:
            rows = [d.step_fused(5e-3, sc(3 + k), [], lazy=True) for k in range(K)]
        log = [(o["N"], o["hits"], list(o["sign"])) for o in rows]
        ends = [d.download(f, 4096, off) for f in (hip.R0, hip.R1, hip.V2, hip.DV0, hip.DR1) for off in (0, N // 2 + 77, N - 4096)]
        out.append((log, ends))
        d.close()
    assert out[0][0] == out[1][0]
    assert all(np.array_equal(a, b) for a, b in zip(out[0][1], out[1][1]))
    assert out[0][0][0][1] == N and 0.3 * N < out[0][0][-1][1] < 0.9 * N          # step 1: every photon at x > 0 scatters


def test_full_size_delete_multi_equals_single_steps_at_1e8(make_store, hip):
    """BASELINE configs[1](ii) at 1e8 photons: 5 delete loop bodies in one pass + one compaction == 5 fused delete
    launches (alive / removed / sign / plane rows, survivor ids and positions at both ends of the store)."""
    N, K, seed = 100_000_000, 5, 99
    plane = [[1.0 / (1e-3 * 1e-3), np.nan, np.nan]]
    out = []
    for multi in (True, False):
        d = make_store(N)
        d.fill_photons(N, 0, C_LIT, 1.0, 1.0, seed)
        if multi:
            rows = d.step_fused_delete_multi(1e-3, K, 1e-3, 1e-3, seed, 0, plane)
        else:
            rows = [d.step_fused_delete(1e-3, 1e-3, 1e-3, hip.RNG_PHILOX, seed, k, plane, lazy=True) for k in range(K)]
        log = [(o["N"], o["removed"], list(o["sign"]), list(o["planes"])) for o in rows]
        n = d.count
        ends = [d.download_ids(4096, off) for off in (0, n // 2, n - 4096)] + \
               [d.download(f, 4096, off) for f in (hip.R0, hip.DR0, hip.E) for off in (0, n - 4096)]
        out.append((log, n, ends))
        d.close()
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert all(np.array_equal(a, b) for a, b in zip(out[0][2], out[1][2]))
    assert abs(out[0][1] / N - 0.7002 ** K) < 1e-3


def test_the_formulation_follows_the_hit_fraction_and_nothing_shows(hip):
    """Four launches of 24 steps on the bench's workload (hit fraction 1.0 in the first step, below 25 % after ~70 steps):
    the rows and the state equal 96 single steps.  Since round 5 a wave owns 256 photons at every hit fraction (their
    velocities live in LDS: pcl_multi_body_lds); PCL_MULTI_NQ2=0 -- one of conftest.py's knob cases -- takes the
    128-photon instantiation; the kernel's own tally says which one ran."""
    N, seed = 200_000, 5
    expr = "0.000000001 * exp(r0[gid] - 5)"
    sc = lambda k: dict(A=1e-15, n=1e-19, flags=hip.SCATTER_WAVELENGTH | hip.SCATTER_VARIABLE_N, c=299792458.0, h=6.62607015e-34,
                        n_expr=expr, rng_mode=hip.RNG_PHILOX, seed=seed, step=k)
    out, forms = {}, []
    for how in ("multi", "single"):
        with hip.Device(0) as d:
            d.store_alloc(N)
            d.fill_photons(N, 0, 299792458.0, 2.8e-19, 9.9e-19, seed)
            rows = []
            if how == "multi":
                for k in range(0, 96, 24):
                    rows += [(o["hits"], tuple(int(x) for x in o["sign"])) for o in d.step_fused_multi(5e-3, 24, sc(k))]
                    forms.append(d.last_multi_work()[2])
                    ghz = d.last_multi_clock()                       # GHz the chip held under the launch, measured in the kernel
                    assert np.isfinite(ghz) and ghz > 0              # (which band it falls in is the box's business: the instrumentation test below)
            else:
                rows = [(o["hits"], tuple(int(x) for x in o["sign"])) for o in (d.step_fused(5e-3, sc(k), [], lazy=True) for k in range(96))]
            out[how] = (rows, d.download_state())
    assert out["multi"][0] == out["single"][0]
    assert out["multi"][0][71][0] < 0.25 * N < out["multi"][0][23][0]     # the last launch started below the threshold, the second above
    if os.environ.get("PCL_MULTI_NQ2") == "0":
        assert forms == [128] * 4
    elif os.environ.get("PCL_MULTI_NQ3") == "1":
        assert forms == [192] * 4
    elif os.environ.get("PCL_MULTI_NQ3") == "0":
        assert forms == [256] * 4
    else:                                       # by itself: 192 photons per wave for the launch that starts at h = 0.28 .. 0.355
        assert forms[0] == 256 and set(forms) <= {192, 256}
        starts = [N] + [out["multi"][0][k - 1][0] for k in (24, 48, 72)]       # hits of the step before each launch (the first: unknown)
        assert forms[1:] == [192 if 0.28 * N <= h < 0.355 * N else 256 for h in starts[1:]]
    for f in ("r", "v", "dr", "dv"):
        for k in range(3):
            assert np.array_equal(out["multi"][1][f][k], out["single"][1][f][k]), (f, k)


def test_instrumentation_the_clock_the_kernel_measures_is_a_plausible_engine_clock(hip):
    """INSTRUMENTATION, not parity: pcl_store_last_multi_clock (shader cycles over 100 MHz ticks, pcl_clock_begin / _end) reads
    between 1.0 and 2.6 GHz on an MI355X under its default power cap.  A box that clocks otherwise fails HERE and nowhere else."""
    N = 2_000_000
    with hip.Device(0) as d:
        d.store_alloc(N)
        d.fill_photons(N, 0, 299792458.0, 2.8e-19, 9.9e-19, 5)
        d.step_fused_multi(5e-3, 24, dict(A=1e-15, n=1e-19, flags=hip.SCATTER_WAVELENGTH | hip.SCATTER_VARIABLE_N, c=299792458.0,
                                          h=6.62607015e-34, n_expr="0.000000001 * exp(r0[gid] - 5)", rng_mode=hip.RNG_PHILOX, seed=5, step=0))
        assert 1.0 < d.last_multi_clock() < 2.6
