"""GPU parity: the HIP kernels, called through the C ABI, against (1) the golden vectors produced
by running the reference and (2) the CPU oracle on seeded inputs.  Run with ``-m gpu`` on MI355X.

Bars: bit-exact for Euler r/dr, delete flags, compaction indices, counters, ids and anything built
from IEEE + - * / sqrt; <= V_ABS_TOL (4 ulp of |v| = c) per component for scattered velocities
(device sin/cos come from OCML, the reference leaves them to its OpenCL device); a hit-mask
mismatch is accepted only where |pcoll - rand| <= 1e-14 * pcoll (pow/exp rounding ties).
"""
import numpy as np
import pytest

from oracle import physicl_oracle as orc

pytestmark = pytest.mark.gpu

C_LIT = 299792458.0
H_LIT = 6.62607015e-34
V_ABS_TOL = 4 * np.spacing(C_LIT)      # 2.4e-7 m/s
DV_ABS_TOL = 2 * V_ABS_TOL

EXPR_EX = "0.000000001 * exp(r0[gid] - 5)"
EXPR_RAD = "2.5 * exp(-1 * (sqrt(pow(r0[gid], 2) + pow(r1[gid], 2) + pow(r2[gid], 2)) - 6.0)/(3.5))"
ISO = {
    "base": dict(use_E=False, expr=None),
    "lambda": dict(use_E=True, expr=None),
    "varn": dict(use_E=True, expr=EXPR_EX),
    "varn_radial": dict(use_E=False, expr=EXPR_RAD),
    "varn_overflow": dict(use_E=True, expr=EXPR_EX),
}


@pytest.fixture(scope="module")
def hip():
    from physicl_amd import _hip
    return _hip


@pytest.fixture(scope="module")
def dev(hip):
    d = hip.Device(0)
    yield d
    d.close()


def cols(a):
    return [np.ascontiguousarray(a[:, i]) for i in range(3)]


def flags_of(hip, cfg):
    return (hip.SCATTER_WAVELENGTH if cfg["use_E"] else 0) | (hip.SCATTER_VARIABLE_N if cfg["expr"] else 0)


def assert_hits(hit_dev, hit_ref, pcoll, rand):
    mism = hit_dev != hit_ref
    if mism.any():
        assert np.all(np.abs(pcoll[mism] - rand[mism]) <= 1e-14 * np.abs(pcoll[mism])), \
            "%d hit-mask mismatches away from a tie" % mism.sum()


# ============================================================================ Level 1 kernels
def test_device_is_mi355x(dev):
    info = dev.info()
    assert "gfx950" in info["name"] and info["wavefront"] == 64 and info["compute_units"] >= 200


def test_l1_delete_kernels_bit_exact_vs_reference(golden, dev):
    z = golden("g4_delete")
    A_k, n_k = float(z["n_user"]), float(z["A_user"])        # kernel constants after the swap
    for k in range(int(z["K"])):
        g = lambda nm: z["k%d_%s" % (k, nm)]
        N = len(g("rand"))
        d = [dev.array(g(nm)) for nm in ("d0", "d1", "d2", "rand")]
        res = dev.empty(N, np.int32)
        dev.k_scatter_delete_test(d[0], d[1], d[2], d[3], A_k, n_k, res, N)          # (.., A, n, res)
        assert np.array_equal(res.get(), g("flags"))
        res.fill_bytes(0xFF)
        dev.k_light_scatter_step_del(d[0], d[1], d[2], d[3], n_k, A_k, res, N)       # (.., n, A, result)
        assert np.array_equal(res.get(), g("flags"))
        for a in d + [res]:
            a.free()


@pytest.mark.parametrize("N", [1, 2, 63, 64, 65, 127, 2047, 2048, 2049, 4097, 100003, 1 << 20, 3_000_001])
@pytest.mark.parametrize("p_remove", [0.0, 0.3, 1.0])
def test_l1_compact_indices_bit_exact(dev, N, p_remove):
    rs = np.random.RandomState(N % 9973 + int(p_remove * 10))
    flags = (rs.random_sample(N) < p_remove).astype(np.int32)
    want = orc.survivors(flags)
    dflags, didx = dev.array(flags, np.int32), dev.empty(N, np.int64)
    didx.fill_bytes(0xEE)
    keep = dev.k_compact_indices(dflags, N, didx)
    got = didx.get()
    assert keep == len(want)
    assert np.array_equal(got[:keep], want)
    if keep < N:
        assert np.all(got[keep:] == np.int64(-1229782938247303442))  # 0xEE.. : nothing written past the end
    dflags.free(), didx.free()


def test_l1_compact_indices_empty(dev, hip):
    d = dev.empty(1, np.int32)
    o = dev.empty(1, np.int64)
    assert dev.k_compact_indices(d, 0, o) == 0
    d.free(), o.free()


@pytest.mark.parametrize("tag", sorted(ISO))
def test_l1_sphere_kernel_vs_reference(golden, dev, hip, tag):
    z = golden("g2_iso_" + tag)
    cfg = ISO[tag]
    for k in range(int(z["K"])):
        g = lambda nm: z["k%d_%s" % (k, nm)]
        N = len(g("rand"))
        a = {nm: dev.array(g(nm)) for nm in ("d0", "d1", "d2", "rtheta", "rphi", "rand")}
        E = dev.array(g("E")) if cfg["use_E"] else None
        r = [dev.array(g("r%d" % i)) for i in range(3)] if cfg["expr"] else None
        res = [dev.array(np.full(N, -7.0)) for _ in range(3)]
        dev.k_light_scatter_step_sphere(a["d0"], a["d1"], a["d2"], a["rtheta"], a["rphi"], a["rand"], float(g("A")),
                                        float(g("n")), E, r, res[0], res[1], res[2], N, flags_of(hip, cfg), C_LIT,
                                        H_LIT, cfg["expr"])
        out = [x.get() for x in res]
        ref_hit = ~np.isnan(g("res0"))
        hit = ~np.isnan(out[0])
        pc = orc.scatter_pcoll(g("d0"), g("d1"), g("d2"), float(g("A")), float(g("n")), h=H_LIT, c=C_LIT,
                               E=g("E") if cfg["use_E"] else None, n_expr=cfg["expr"],
                               r=[g("r0"), g("r1"), g("r2")] if cfg["expr"] else None)
        assert_hits(hit, ref_hit, pc, g("rand"))
        both = hit & ref_hit
        for mine, nm in zip(out, ("res0", "res1", "res2")):
            assert np.max(np.abs(mine[both] - g(nm)[both]), initial=0.0) <= V_ABS_TOL
        assert np.all(out[1][~hit] == -7.0) and np.all(out[2][~hit] == -7.0)   # untouched on a miss
        for x in list(a.values()) + res + ([E] if E else []) + (r or []):
            x.free()


def test_bad_expression_is_rejected_before_launch(dev, hip):
    a = dev.array(np.ones(4))
    with pytest.raises(hip.ExpressionError):
        dev.k_light_scatter_step_sphere(a, a, a, a, a, a, 1.0, 1.0, None, [a, a, a], a, a, a, 4,
                                        hip.SCATTER_VARIABLE_N, C_LIT, H_LIT, "r0[gid + 1]")
    with pytest.raises(hip.ExpressionError):       # passes the validator, fails in hipRTC (pow arity)
        dev.k_light_scatter_step_sphere(a, a, a, a, a, a, 1.0, 1.0, None, [a, a, a], a, a, a, 4,
                                        hip.SCATTER_VARIABLE_N, C_LIT, H_LIT, "pow(r0[gid])")
    a.free()


# ============================================================================ Level 2 store
@pytest.fixture()
def store(dev):
    def make(capacity):
        dev.store_alloc(capacity)
        return dev
    yield make
    dev.store_free()


@pytest.mark.parametrize("case", [0, 1, 2])
def test_newton_bit_exact_vs_reference(golden, store, hip, case):
    z = golden("g1_newton")
    N = len(z["r_init"])
    d = store(N)
    d.upload_state({"r": z["r_init"], "v": z["v_init"], "E": np.ones(N)})
    dt = float(z["c%d_dt" % case])
    for k in range(1, 11):
        d.step_newton(dt)
        if k in (1, 10):
            s = d.download_state()
            assert np.array_equal(np.stack(s["r"], 1), z["c%d_r_after%d" % (case, k)])
            assert np.array_equal(np.stack(s["dr"], 1), z["c%d_dr_after%d" % (case, k)])
            assert np.array_equal(np.stack(s["v"], 1), z["v_init"])


@pytest.mark.parametrize("N", [1, 2, 3, 511, 512, 513, 100001])
def test_newton_ragged_sizes_bit_exact(store, N):
    rs = np.random.RandomState(N)
    r, v = rs.normal(size=(N, 3)) * 1e5, rs.normal(size=(N, 3)) * 1e8
    d = store(N + 7)
    d.upload_state({"r": r, "v": v, "E": np.ones(N)})
    rr, vv = cols(r), cols(v)
    for _ in range(3):
        d.step_newton(1.25e-4)
        rr, dr = orc.newton_euler(rr, vv, 1.25e-4)
    s = d.download_state()
    assert np.array_equal(np.stack(s["r"], 1), np.stack(rr, 1))
    assert np.array_equal(np.stack(s["dr"], 1), np.stack(dr, 1))


def test_newton_config1_100_steps(golden, store):
    z = golden("g1_newton")
    N = 10000
    d = store(N)
    d.upload_state({"v": np.tile([C_LIT, 0, 0], (N, 1)), "E": np.ones(N)})
    for _ in range(100):
        d.step_newton(0.001)
    s = d.download_state()
    assert np.all(np.stack(s["r"], 1) == z["cfg1_r_after100"])
    assert np.all(np.stack(s["dr"], 1) == z["cfg1_dr_after100"])


@pytest.mark.parametrize("tag", sorted(ISO))
def test_fused_scatter_rng_input_vs_reference(golden, store, hip, tag):
    """Per step: device state := reference state before the scatter, randoms := the reference's draws;
    the fused kernel must reproduce the reference's post-write-back v and dv."""
    z = golden("g2_iso_" + tag)
    cfg = ISO[tag]
    N = len(z["k0_rand"])
    d = store(N)
    v_prev = np.tile([C_LIT, 0.0, 0.0], (N, 1))
    for k in range(int(z["K"])):
        g = lambda nm: z["k%d_%s" % (k, nm)]
        d.upload_state({"r": g("post_r"), "v": v_prev, "dr": g("post_dr"), "dv": np.full((N, 3), 123.0),
                        "E": z["init_E"]})
        for w, nm in enumerate(("rtheta", "rphi", "rand")):
            d.upload_rand(w, g(nm))
        hits = d.step_scatter_isotropic(float(g("A")), float(g("n")), flags_of(hip, cfg), C_LIT, H_LIT, cfg["expr"],
                                        rng_mode=hip.RNG_INPUT)
        s = d.download_state()
        ref_hit = ~np.isnan(g("res0"))
        v_dev, dv_dev = np.stack(s["v"], 1), np.stack(s["dv"], 1)
        hit_dev = np.any(v_dev != v_prev, axis=1) | np.any(dv_dev != 0.0, axis=1)
        pc = orc.scatter_pcoll(g("d0"), g("d1"), g("d2"), float(g("A")), float(g("n")), h=H_LIT, c=C_LIT,
                               E=g("E") if cfg["use_E"] else None, n_expr=cfg["expr"],
                               r=cols(g("post_r")) if cfg["expr"] else None)
        assert_hits(hit_dev, ref_hit, pc, g("rand"))
        assert abs(hits - ref_hit.sum()) <= (hit_dev != ref_hit).sum()
        same = hit_dev == ref_hit
        assert np.max(np.abs(v_dev[same] - g("post_v")[same])) <= V_ABS_TOL
        assert np.max(np.abs(dv_dev[same] - g("post_dv")[same])) <= DV_ABS_TOL
        miss = ~hit_dev
        assert np.array_equal(v_dev[miss], v_prev[miss]) and np.all(dv_dev[miss] == 0.0)
        assert np.array_equal(np.stack(s["r"], 1), g("post_r"))           # untouched
        v_prev = g("post_v")


@pytest.mark.parametrize("tag", ["base", "varn"])
def test_fused_chain_newton_scatter_vs_reference(golden, store, hip, tag):
    """Whole-step chain on the device from the initial state with the reference's random stream:
    positions stay within the error the <=4-ulp velocity tolerance allows."""
    z = golden("g2_iso_" + tag)
    cfg = ISO[tag]
    N = len(z["k0_rand"])
    d = store(N)
    d.upload_state({"r": z["init_r"], "v": np.tile([C_LIT, 0.0, 0.0], (N, 1)), "E": z["init_E"]})
    dt = float(z["dt"])
    for k in range(int(z["K"])):
        g = lambda nm: z["k%d_%s" % (k, nm)]
        d.step_newton(dt)
        for w, nm in enumerate(("rtheta", "rphi", "rand")):
            d.upload_rand(w, g(nm))
        d.step_scatter_isotropic(float(g("A")), float(g("n")), flags_of(hip, cfg), C_LIT, H_LIT, cfg["expr"],
                                 rng_mode=hip.RNG_INPUT)
        s = d.download_state()
        assert np.max(np.abs(np.stack(s["v"], 1) - g("post_v"))) <= V_ABS_TOL
        tol_r = (k + 1) * V_ABS_TOL * dt + 4 * np.spacing(np.abs(g("post_r")).max())
        assert np.max(np.abs(np.stack(s["r"], 1) - g("post_r"))) <= tol_r
        cnt = d.step_counters(z["planes"])
        assert cnt[hip.CNT_N] == N
        # sign counters are exact unless a component sits within tolerance of zero
        near0 = (np.abs(g("post_v")) <= V_ABS_TOL).sum(axis=0)
        for ax in range(3):
            assert abs(int(cnt[hip.CNT_XP + ax]) - int(z["sign_rows"][k][2 + ax])) <= near0[ax]


@pytest.mark.parametrize("tag", ["base", "lambda", "varn"])
@pytest.mark.parametrize("N", [1, 1000, 262147])
def test_fused_scatter_philox_vs_oracle(golden, store, hip, tag, N):
    cfg = ISO[tag]
    rs = np.random.RandomState(N + len(tag))
    st = {"r": cols(rs.uniform(-10, 10, (N, 3))), "v": [np.full(N, C_LIT), np.zeros(N), np.zeros(N)],
          "dr": [np.zeros(N)] * 3, "dv": [np.zeros(N)] * 3,
          "E": rs.uniform(2.8e-19, 9.9e-19, N), "id": np.arange(N, dtype=np.int64) + 5_000_000_000}
    if tag == "base":
        A_k, n_k, dt = 1e-3, 1e-3, 1e-3
    elif tag == "lambda":
        A_k, n_k, dt = 1e-15, 1e-19, 5e-3
    else:
        A_k, n_k, dt = 1e-15, 1e-19, 1e-9
    d = store(N)
    d.upload_state({"r": np.stack(st["r"], 1), "v": np.stack(st["v"], 1), "E": st["E"], "id_base": 5_000_000_000})
    seed = 0xDEADBEEF12345
    for step in range(3):
        d.step_newton(dt)
        orc.step_newton(st, dt)
        hits = d.step_scatter_isotropic(A_k, n_k, flags_of(hip, cfg), C_LIT, H_LIT, cfg["expr"],
                                        rng_mode=hip.RNG_PHILOX, seed=seed, step=step)
        draws = orc.philox_draws(seed, step, st["id"])
        v_before = [x.copy() for x in st["v"]]
        pc = orc.scatter_pcoll(*st["dr"], A_k, n_k, h=H_LIT, c=C_LIT, E=st["E"] if cfg["use_E"] else None,
                               n_expr=cfg["expr"], r=st["r"] if cfg["expr"] else None)
        hit = orc.step_scatter_isotropic(st, draws, A_k, n_k, C_LIT, h=H_LIT, use_E=cfg["use_E"], n_expr=cfg["expr"])
        s = d.download_state()
        v_dev, dv_dev = np.stack(s["v"], 1), np.stack(s["dv"], 1)
        hit_dev = np.any(dv_dev != 0.0, axis=1) | np.any(v_dev != np.stack(v_before, 1), axis=1)
        assert_hits(hit_dev, hit, pc, draws[2])
        same = hit_dev == hit
        assert same.all() or N > 1000
        assert hits == hit_dev.sum()
        assert np.max(np.abs(v_dev[same] - np.stack(st["v"], 1)[same])) <= V_ABS_TOL
        assert np.max(np.abs(dv_dev[same] - np.stack(st["dv"], 1)[same])) <= DV_ABS_TOL
        assert np.array_equal(np.stack(s["r"], 1), np.stack(st["r"], 1))
        # continue the oracle from the device's velocities so later steps compare like with like
        st["v"] = cols(v_dev)


def test_delete_chain_rng_input_bit_exact_vs_reference(golden, store, hip):
    """Newton + ScatterDelete until the store is empty with the reference's random stream:
    surviving ids, positions, alive counts and plane-crossing rows are all exact."""
    z = golden("g4_delete")
    N = int(z["N"])
    d = store(N)
    d.upload_state({"v": np.tile([C_LIT, 0.0, 0.0], (N, 1)), "E": np.ones(N)})
    for k in range(int(z["K"])):
        g = lambda nm: z["k%d_%s" % (k, nm)]
        n_before = d.count
        d.step_newton(float(z["dt"]))
        d.upload_rand(2, g("rand"))
        alive, removed = d.step_scatter_delete(float(z["n_user"]), float(z["A_user"]), rng_mode=hip.RNG_INPUT)
        assert alive + removed == n_before and alive == len(g("survivor_uid")) == d.count
        assert np.array_equal(d.last_delete_flags(n_before), g("flags"))
        assert np.array_equal(d.download_ids(), g("survivor_uid"))
        s = d.download_state()
        if k < 3:
            assert np.array_equal(np.stack(s["r"], 1), g("post_r"))
        cnt = d.step_counters(z["planes"])
        row, srow = z["measure_rows"][k], z["sign_rows"][k]
        assert cnt[hip.CNT_N] == row[1]
        assert [int(x) for x in cnt[hip.CNT_PLANE0:]] == [int(x) for x in row[2:]]
        assert [int(x) for x in cnt[hip.CNT_XP:hip.CNT_XP + 3]] == [int(x) for x in srow[2:5]]
    assert d.count == 0
    # an empty store is a valid input for every step
    d.step_newton(1e-3)
    assert d.step_scatter_delete(1e-3, 1e-3, rng_mode=hip.RNG_PHILOX) == (0, 0)
    assert d.step_scatter_isotropic(1e-3, 1e-3, 0, C_LIT, H_LIT) == 0
    assert list(d.step_counters([[0.0, np.nan, np.nan]])) == [0, 0, 0, 0, 0]


@pytest.mark.parametrize("N", [1, 64, 2049, 1_000_003])
def test_delete_philox_multi_step_vs_oracle(store, hip, N):
    rs = np.random.RandomState(N)
    st = {"r": cols(rs.normal(size=(N, 3))), "v": cols(rs.normal(size=(N, 3)) * 1e8), "dr": [np.zeros(N)] * 3,
          "dv": cols(rs.normal(size=(N, 3))), "E": rs.uniform(1, 2, N), "id": np.arange(N, dtype=np.int64) + 77}
    d = store(N)
    d.upload_state({"r": np.stack(st["r"], 1), "v": np.stack(st["v"], 1), "dv": np.stack(st["dv"], 1), "E": st["E"],
                    "id_base": 77})
    seed, A_k, n_k, dt = 99, 2e-3, 1e-3, 1e-3
    for step in range(6):
        d.step_newton(dt)
        orc.step_newton(st, dt)
        alive, removed = d.step_scatter_delete(A_k, n_k, rng_mode=hip.RNG_PHILOX, seed=seed, step=step)
        n_before = len(st["id"])
        flags, keep = orc.step_scatter_delete(st, orc.philox_draws(seed, step, st["id"])[2], A_k, n_k)
        assert (alive, removed) == (len(keep), n_before - len(keep))
        assert np.array_equal(d.last_delete_flags(n_before), flags)
        s = d.download_state()
        assert np.array_equal(s["id"], st["id"])
        for f in ("r", "v", "dr", "dv"):
            assert np.array_equal(np.stack(s[f], 1).reshape(-1, 3), np.stack(st[f], 1).reshape(-1, 3)), f
        assert np.array_equal(s["E"], st["E"])
        if alive == 0:
            break


def test_mixed_kinds_light_steps_skip_plain_objects(store, hip):
    """``if type(obj) != PhotonObject: continue`` (light.py:233, 283): plain Objects move but are never
    scattered or deleted."""
    N = 5000
    rs = np.random.RandomState(5)
    kind = (rs.random_sample(N) < 0.7).astype(np.uint8)
    d = store(N)
    v = np.tile([C_LIT, 0.0, 0.0], (N, 1))
    d.upload_state({"v": v, "dv": np.full((N, 3), 9.0), "E": np.ones(N), "kind": kind})
    d.step_newton(1e-3)
    hits = d.step_scatter_isotropic(1.0, 1.0, 0, C_LIT, H_LIT, rng_mode=hip.RNG_PHILOX, seed=1, step=0)  # pcoll >> 1
    s = d.download_state()
    assert hits == kind.sum()
    obj = kind == 0
    assert np.array_equal(np.stack(s["v"], 1)[obj], v[obj]) and np.all(np.stack(s["dv"], 1)[obj] == 9.0)
    assert np.all(np.stack(s["r"], 1)[:, 0] == C_LIT * 1e-3)
    alive, removed = d.step_scatter_delete(1.0, 1.0, rng_mode=hip.RNG_PHILOX, seed=1, step=1)
    assert removed == kind.sum() and alive == obj.sum()
    assert np.array_equal(d.download_ids(), np.flatnonzero(obj))
    assert np.all(d.download_kind() == 0)


def test_results_do_not_depend_on_sharding(store, dev, hip):
    """Philox is keyed by the global particle id: two shards give exactly the rows of the whole."""
    N, half = 200_001, 100_000
    rs = np.random.RandomState(3)
    E = rs.uniform(2.8e-19, 9.9e-19, N)
    r = rs.uniform(-10, 10, (N, 3))
    v = np.tile([C_LIT, 0.0, 0.0], (N, 1))

    def run(lo, hi):
        d = store(hi - lo)
        d.upload_state({"r": r[lo:hi], "v": v[lo:hi], "E": E[lo:hi], "id_base": lo})
        tot_hits = 0
        for step in range(3):
            d.step_newton(1e-9)
            tot_hits += d.step_scatter_isotropic(1e-15, 1e-19, 3, C_LIT, H_LIT, EXPR_EX, hip.RNG_PHILOX, 42, step)
        d.step_newton(1e-3)
        alive, _ = d.step_scatter_delete(1e-3, 1e-3, hip.RNG_PHILOX, 42, 3)
        s = d.download_state()
        cnt = d.step_counters([[0.0, np.nan, np.nan]])
        return s, tot_hits, cnt

    whole, hits_w, cnt_w = run(0, N)
    a, hits_a, cnt_a = run(0, half)
    b, hits_b, cnt_b = run(half, N)
    assert hits_w == hits_a + hits_b
    assert np.array_equal(cnt_w, cnt_a + cnt_b)          # the vector RCCL all-reduces
    assert np.array_equal(whole["id"], np.concatenate([a["id"], b["id"]]))
    for f in ("r", "v", "dr", "dv"):
        for k in range(3):
            assert np.array_equal(whole[f][k], np.concatenate([a[f][k], b[f][k]]))


def test_fill_photons_matches_oracle_energy_sampler(store, hip):
    N, base, seed = 100_000, 1 << 33, 2024
    e_lo, e_hi = 2.8378e-19, 9.9322e-19
    d = store(N)
    d.fill_photons(N, base, C_LIT, e_lo, e_hi, seed)
    s = d.download_state()
    want = orc.philox_energy(seed, np.arange(N, dtype=np.int64) + base, e_lo, e_hi)
    assert np.max(np.abs(s["E"] - want) / want) <= 1e-15          # pow(u, 1/3) within a few ulp
    assert np.array_equal(s["id"], np.arange(N, dtype=np.int64) + base)
    assert np.all(s["v"][0] == C_LIT) and not np.any(s["v"][1]) and not np.any(s["v"][2])
    for f in ("r", "dr", "dv"):
        assert not np.any(np.stack(s[f]))
    assert e_lo <= s["E"].min() and s["E"].max() <= e_hi
    # E ~ min + (max-min) * U^(1/3): P(U^(1/3) < x) = x^3
    frac = np.mean((s["E"] - e_lo) / (e_hi - e_lo) < 0.5)
    assert abs(frac - 0.125) < 0.005


def test_state_errors_are_reported(dev, hip):
    dev.store_free()
    with pytest.raises(hip.HipError, match="pcl_store_alloc"):
        dev.step_newton(1e-3)
    dev.store_alloc(10)
    dev.set_count(10)
    with pytest.raises(hip.HipError, match="upload_rand"):
        dev.step_scatter_isotropic(1.0, 1.0, 0, C_LIT, H_LIT, rng_mode=hip.RNG_INPUT)
    with pytest.raises(hip.HipError):
        dev.set_count(11)
    with pytest.raises(hip.HipError):
        dev.step_counters(np.zeros((13, 3)))
    dev.store_free()


# ============================================================================ full-size properties
@pytest.mark.parametrize("N", [10_000_000, 100_000_000])
def test_full_size_properties(dev, hip, N):
    """BASELINE.json sizes (1e7, 1e8): size-independent properties on the device.
    * Newton from r=0, v=(c,0,0): every r_x == c*dt exactly (one value), checked through counters.
    * Delete: alive + removed == N; survivors' ids strictly ascending (stable); the survival
      fraction is 1 - pcoll within 5 sigma; a second delete of an all-certain store empties it.
    * Scatter: hits == number of particles whose dv != 0; counters of a sharded run add up.
    """
    dev.store_alloc(N)
    try:
        e_lo, e_hi = 2.8378e-19, 9.9322e-19
        dev.fill_photons(N, 0, C_LIT, e_lo, e_hi, 7)
        dt = 1e-3
        dev.step_newton(dt)
        x = C_LIT * dt
        cnt = dev.step_counters([[x, np.nan, np.nan], [np.nextafter(x, np.inf), np.nan, np.nan], [np.nan, 0.0, np.nan]])
        assert list(cnt) == [N, N, 0, 0, N, 0, N]
        # scatter: pcoll = 1e-3*1e-3*c*dt = 0.2998
        hits = dev.step_scatter_isotropic(1e-3, 1e-3, 0, C_LIT, H_LIT, rng_mode=hip.RNG_PHILOX, seed=11, step=0)
        p = 1e-3 * 1e-3 * x
        assert abs(hits - N * p) < 5 * np.sqrt(N * p * (1 - p))
        # every scattered photon has |v| = c to rounding and dv = v - (c,0,0); sample a slice
        m = 1 << 20
        v = [dev.download(hip.V0 + k, m, N - m) for k in range(3)]
        dv = [dev.download(hip.DV0 + k, m, N - m) for k in range(3)]
        hit = (dv[0] != 0) | (dv[1] != 0) | (dv[2] != 0)
        assert abs(hit.mean() - p) < 5 * np.sqrt(p * (1 - p) / m)
        speed = np.sqrt(v[0] ** 2 + v[1] ** 2 + v[2] ** 2)
        assert np.max(np.abs(speed - C_LIT)) < 1e-6
        assert np.array_equal(dv[0][hit], v[0][hit] - C_LIT) and np.all(v[0][~hit] == C_LIT)
        # delete
        alive, removed = dev.step_scatter_delete(1e-3, 1e-3, rng_mode=hip.RNG_PHILOX, seed=11, step=1)
        assert alive + removed == N and dev.count == alive
        # |dr| = c*dt for all (scatter changed v, not dr) -> same pcoll
        assert abs(removed - N * p) < 5 * np.sqrt(N * p * (1 - p))
        ids = dev.download_ids(m, alive - m)
        assert np.all(np.diff(ids) > 0) and ids[-1] < N
        ids0 = dev.download_ids(m, 0)
        assert np.all(np.diff(ids0) > 0) and ids0[0] >= 0
        # survivors keep their own data: E is a pure function of id
        E = dev.download(hip.E, m, alive - m)
        want = orc.philox_energy(7, ids, e_lo, e_hi)
        assert np.max(np.abs(E - want) / want) <= 1e-15
        # certain deletion empties the store
        alive2, removed2 = dev.step_scatter_delete(1.0, 1.0, rng_mode=hip.RNG_PHILOX, seed=11, step=2)
        assert (alive2, removed2) == (0, alive)
    finally:
        dev.store_free()


# ============================================================================ fused loop body
def _run_separate(d, hip, dt, sc, planes):
    d.step_newton(dt)
    hits = d.step_scatter_isotropic(sc["A"], sc["n"], sc["flags"], sc["c"], sc["h"], sc.get("n_expr"),
                                    rng_mode=sc["rng_mode"], seed=sc.get("seed", 0), step=sc.get("step", 0))
    cnt = d.step_counters(planes)
    return hits, cnt


@pytest.mark.parametrize("tag", ["base", "lambda", "varn", "varn_radial"])
@pytest.mark.parametrize("N", [1, 2, 777, 200_001])
def test_fused_step_is_bit_identical_to_separate_steps(store, hip, tag, N):
    """pcl_step_fused == pcl_step_newton + pcl_step_scatter_isotropic + pcl_step_counters, bit for bit
    (same device arithmetic), for every variant, odd sizes, mixed kinds and both RNG modes."""
    cfg = ISO[tag]
    rs = np.random.RandomState(N + 31 * len(tag))
    init = {"r": rs.uniform(-8, 8, (N, 3)), "v": np.tile([C_LIT, 0.0, 0.0], (N, 1)),
            "dv": rs.normal(size=(N, 3)), "E": rs.uniform(2.8e-19, 9.9e-19, N), "id_base": 1 << 34,
            "kind": (rs.random_sample(N) < 0.9).astype(np.uint8)}
    if tag == "base":
        A_k, n_k, dt = 1e-3, 1e-3, 1e-3
    elif tag == "lambda":
        A_k, n_k, dt = 1e-15, 1e-19, 5e-3
    elif tag == "varn":
        A_k, n_k, dt = 1e-15, 1e-19, 1e-9
    else:
        A_k, n_k, dt = 0.5, 123.0, 1e-9
    planes = [[0.5, np.nan, np.nan], [np.nan, -1.0, np.nan], [np.nan, np.nan, 2.0]]
    results = {}
    for mode in ("separate", "fused"):
        for rng_mode in (hip.RNG_PHILOX, hip.RNG_INPUT):
            d = store(N)
            d.upload_state(init)
            rs2 = np.random.RandomState(99)
            log = []
            for step in range(3):
                sc = dict(A=A_k, n=n_k, flags=flags_of(hip, cfg), c=C_LIT, h=H_LIT, n_expr=cfg["expr"],
                          rng_mode=rng_mode, seed=4242, step=step)
                if rng_mode == hip.RNG_INPUT:
                    rt, rp, ra = orc.reference_draws(N, rs2)
                    d.upload_rand(0, rt), d.upload_rand(1, rp), d.upload_rand(2, ra)
                if mode == "separate":
                    hits, cnt = _run_separate(d, hip, dt, sc, planes)
                    log.append((hits, list(cnt)))
                else:
                    out = d.step_fused(dt, sc, planes)
                    log.append((out["hits"], [out["N"]] + list(out["sign"]) + list(out["planes"])))
                    assert d.last_scatter_hits() == out["hits"]
            results[(mode, rng_mode)] = (d.download_state(), log)
    for rng_mode in (hip.RNG_PHILOX, hip.RNG_INPUT):
        (sa, la), (sb, lb) = results[("separate", rng_mode)], results[("fused", rng_mode)]
        assert la == lb
        assert la[0][0] > 0 or N < 10
        for f in ("r", "v", "dr", "dv"):
            for k in range(3):
                assert np.array_equal(sa[f][k], sb[f][k]), (f, k)


def test_fused_newton_only_and_counters_only(store, hip):
    N = 4099
    rs = np.random.RandomState(8)
    r, v = rs.normal(size=(N, 3)), rs.normal(size=(N, 3))
    d = store(N)
    d.upload_state({"r": r, "v": v, "dv": np.full((N, 3), 5.0), "E": np.ones(N)})
    assert d.step_fused(0.25) is None                         # Newton only, no counters, no sync
    out = d.step_fused(0.25, None, [[np.nan, 0.1, np.nan]])    # Newton + counters
    rr, vv = cols(r), cols(v)
    for _ in range(2):
        rr, dr = orc.newton_euler(rr, vv, 0.25)
    s = d.download_state()
    assert np.array_equal(np.stack(s["r"], 1), np.stack(rr, 1)) and np.array_equal(np.stack(s["dr"], 1), np.stack(dr, 1))
    assert np.all(np.stack(s["dv"], 1) == 5.0)                # untouched without a scatter step
    assert tuple(out["sign"]) == orc.sign_counts(vv) and out["hits"] == 0 and out["N"] == N
    assert out["planes"][0] == orc.plane_crossings(rr, dr, [np.nan, 0.1, np.nan])


def test_fused_chain_vs_reference(golden, store, hip):
    """Fused loop body against the reference's golden chain (same check as the unfused chain test)."""
    z = golden("g2_iso_base")
    N = len(z["k0_rand"])
    d = store(N)
    d.upload_state({"r": z["init_r"], "v": np.tile([C_LIT, 0.0, 0.0], (N, 1)), "E": z["init_E"]})
    for k in range(int(z["K"])):
        g = lambda nm: z["k%d_%s" % (k, nm)]
        for w, nm in enumerate(("rtheta", "rphi", "rand")):
            d.upload_rand(w, g(nm))
        out = d.step_fused(float(z["dt"]), dict(A=float(g("A")), n=float(g("n")), flags=0, c=C_LIT, h=H_LIT,
                                                 rng_mode=hip.RNG_INPUT), z["planes"])
        s = d.download_state()
        assert np.max(np.abs(np.stack(s["v"], 1) - g("post_v"))) <= V_ABS_TOL
        assert out["hits"] == (~np.isnan(g("res0"))).sum()
        assert [int(x) for x in out["planes"]] == [int(x) for x in z["measure_rows"][k][2:]]
        assert [int(x) for x in out["sign"]] == [int(x) for x in z["sign_rows"][k][2:5]]


# ============================================================================ lazy dr/dv (PCL_FUSED_LAZY)
@pytest.mark.parametrize("tag", ["base", "varn"])
@pytest.mark.parametrize("N", [1, 333, 150_001])
def test_lazy_fused_chain_is_bit_identical_to_eager(store, hip, tag, N):
    """A chain of lazy fused steps, materialised by whatever touches the store next, leaves exactly the
    state of the eager chain -- r, v, dr, dv, counters -- including mixed kinds, a Newton-only pass in
    the middle, and a delete step (which needs the real dr) at the end."""
    cfg = ISO[tag]
    rs = np.random.RandomState(N + 7)
    init = {"r": rs.uniform(-8, 8, (N, 3)), "v": np.tile([C_LIT, 0.0, 0.0], (N, 1)), "dv": rs.normal(size=(N, 3)),
            "dr": rs.normal(size=(N, 3)), "E": rs.uniform(2.8e-19, 9.9e-19, N), "id_base": 10,
            "kind": (rs.random_sample(N) < 0.85).astype(np.uint8)}
    A_k, n_k, dt = (1e-3, 1e-3, 1e-3) if tag == "base" else (1e-15, 1e-19, 1e-9)
    planes = [[0.5, np.nan, np.nan], [np.nan, np.nan, -2.0]]
    out = {}
    for lazy in (False, True):
        d = store(N)
        d.upload_state(init)
        log = []
        for step in range(5):
            sc = dict(A=A_k, n=n_k, flags=flags_of(hip, cfg), c=C_LIT, h=H_LIT, n_expr=cfg["expr"],
                      rng_mode=hip.RNG_PHILOX, seed=31337, step=step)
            if step == 2:
                o = d.step_fused(dt * 0.5, None, planes, lazy=lazy)          # Newton-only pass
            else:
                o = d.step_fused(dt, sc, planes, lazy=lazy)
            log.append((o["N"], o["hits"], list(o["sign"]), list(o["planes"])))
            if step == 3:
                mid = d.download_state()                                      # forces a materialise mid-chain
        end = d.download_state()
        alive, removed = d.step_scatter_delete(A_k if tag == "base" else 1e-3, n_k if tag == "base" else 1e-3,
                                               hip.RNG_PHILOX, 31337, 99)
        out[lazy] = (log, mid, end, alive, removed, d.download_state())
    la, lb = out[False], out[True]
    assert la[0] == lb[0] and la[3:5] == lb[3:5]
    for sa, sb in ((la[1], lb[1]), (la[2], lb[2]), (la[5], lb[5])):
        assert np.array_equal(sa["id"], sb["id"]) and np.array_equal(sa["E"], sb["E"])
        for f in ("r", "v", "dr", "dv"):
            for k in range(3):
                assert np.array_equal(sa[f][k], sb[f][k]), (f, k)
    # plain Objects never get a dv from the light step, lazily or not
    obj = init["kind"] == 0
    assert np.array_equal(np.stack(lb[2]["dv"], 1)[obj], init["dv"][obj])


def test_lazy_state_is_materialised_for_every_consumer(store, hip):
    """Each consumer of dr/dv after a lazy step sees real arrays: separate scatter step, counters with
    planes, field pointers, a second store_alloc."""
    N = 10_000
    d = store(N)
    v = np.tile([C_LIT, 0.0, 0.0], (N, 1))
    sc = dict(A=1e-3, n=1e-3, flags=0, c=C_LIT, h=H_LIT, rng_mode=hip.RNG_PHILOX, seed=5, step=0)

    def fresh():
        d.upload_state({"v": v, "E": np.ones(N)})
        d.step_fused(1e-3, sc, None, lazy=True)

    fresh()
    cnt = d.step_counters([[C_LIT * 1e-3, np.nan, np.nan]])             # plane test reads r and dr
    assert cnt[hip.CNT_PLANE0] == N
    fresh()
    hits = d.step_scatter_isotropic(1.0, 1.0, 0, C_LIT, H_LIT, rng_mode=hip.RNG_PHILOX, seed=5, step=1)  # reads dr
    assert hits == N
    fresh()
    assert np.all(d.download(hip.DR0) == C_LIT * 1e-3)
    dv = np.stack([d.download(hip.DV0 + k) for k in range(3)], 1)
    vv = np.stack([d.download(hip.V0 + k) for k in range(3)], 1)
    assert np.array_equal(dv, vv - v)
    fresh()
    d.step_newton(1e-3)                                                  # overwrites dr; dv must survive
    assert np.array_equal(np.stack([d.download(hip.DV0 + k) for k in range(3)], 1),
                          np.stack([d.download(hip.V0 + k) for k in range(3)], 1) - v)


# ============================================================================ fused Newton + delete (+ counters)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("N", [1, 64, 2049, 500_003])
def test_fused_delete_is_bit_identical_to_separate_steps(dev, hip, N, dtype):
    """pcl_step_fused_delete (eager and lazy) == pcl_step_newton + pcl_step_scatter_delete + pcl_step_counters:
    same survivors, same state bit for bit, same counters; mixed kinds; both RNG modes; delete chains."""
    rs = np.random.RandomState(N)
    init = {"r": rs.normal(size=(N, 3)) * 1e5, "v": rs.normal(size=(N, 3)) * 1e8, "dv": rs.normal(size=(N, 3)),
            "dr": rs.normal(size=(N, 3)), "E": rs.uniform(1, 2, N), "id_base": 1000,
            "kind": (rs.random_sample(N) < 0.9).astype(np.uint8)}
    planes = [[1e5, np.nan, np.nan], [np.nan, -2e4, np.nan]]
    results = {}
    for mode in ("separate", "eager", "lazy"):
        for rng_mode in (hip.RNG_PHILOX, hip.RNG_INPUT):
            dev.store_alloc(N, dtype)
            dev.upload_state(init)
            rs2 = np.random.RandomState(7)
            log = []
            for step in range(4):
                n_now = dev.count
                if rng_mode == hip.RNG_INPUT:
                    dev.upload_rand(2, rs2.random_sample(max(n_now, 1)))
                if mode == "separate":
                    dev.step_newton(1e-3)
                    alive, removed = dev.step_scatter_delete(3e-6, 1e-3, rng_mode, 5, step)
                    cnt = dev.step_counters(planes)
                    log.append((alive, removed, list(cnt[1:])))
                else:
                    o = dev.step_fused_delete(1e-3, 3e-6, 1e-3, rng_mode, 5, step, planes, lazy=(mode == "lazy"))
                    log.append((o["N"], o["removed"], list(o["sign"]) + list(o["planes"])))
                flags = dev.last_delete_flags(n_now) if n_now else None
                log.append(None if flags is None else int(flags.sum()))
            results[(mode, rng_mode)] = (log, dev.download_state(), dev.download_kind())
    for rng_mode in (hip.RNG_PHILOX, hip.RNG_INPUT):
        la, sa, ka = results[("separate", rng_mode)]
        for mode in ("eager", "lazy"):
            lb, sb, kb = results[(mode, rng_mode)]
            assert la == lb, (mode, la[:2], lb[:2])
            assert np.array_equal(sa["id"], sb["id"]) and np.array_equal(sa["E"], sb["E"]) and np.array_equal(ka, kb)
            for f in ("r", "v", "dr", "dv"):
                for k in range(3):
                    assert np.array_equal(sa[f][k], sb[f][k]), (mode, f, k)
    assert results[("separate", hip.RNG_PHILOX)][0][0][1] > 0 or N < 100


def test_fused_delete_after_lazy_scatter_keeps_dv(dev, hip):
    """A lazy scatter step leaves dv implicit; the fused delete that follows must move the real dv."""
    N = 50_000
    outs = []
    for lazy in (False, True):
        dev.store_alloc(N)
        dev.fill_photons(N, 0, C_LIT, 1.0, 1.0, 3)
        sc = dict(A=1e-3, n=1e-3, flags=0, c=C_LIT, h=H_LIT, rng_mode=hip.RNG_PHILOX, seed=3, step=0)
        dev.step_fused(1e-3, sc, (), lazy=lazy)
        o = dev.step_fused_delete(1e-3, 1e-3, 1e-3, hip.RNG_PHILOX, 3, 1, [[0.0, np.nan, np.nan]], lazy=lazy)
        outs.append((o["N"], o["removed"], list(o["sign"]), list(o["planes"]), dev.download_state()))
    a, b = outs
    assert a[:4] == b[:4] and a[1] > 0
    for f in ("r", "v", "dr", "dv"):
        for k in range(3):
            assert np.array_equal(a[4][f][k], b[4][f][k]), (f, k)
    assert np.any(a[4]["dv"][0] != 0)


def test_fill_photons_from_planck_table_bit_exact(store, hip):
    N, base, seed = 300_000, 7, 99
    cdf, grid = orc.planck_table(7.9e-20, 9.9e-19, 5778.0, 1000)
    d = store(N)
    d.fill_photons_table(N, base, C_LIT, cdf, grid, seed)
    s = d.download_state()
    assert np.array_equal(s["E"], orc.philox_table_energy(seed, np.arange(N) + base, cdf, grid))   # integer work: exact
    assert np.all(s["v"][0] == C_LIT) and not np.any(s["r"][0]) and np.array_equal(s["id"], np.arange(N) + base)


def test_async_fused_steps_are_pipelined_and_read_in_order(store, hip):
    """Two asynchronous fused steps may be in flight; pcl_step_fused_read returns them oldest first and the
    values equal those of synchronous calls."""
    N = 300_001
    sc = lambda k: dict(A=1e-3, n=1e-3, flags=0, c=C_LIT, h=H_LIT, rng_mode=hip.RNG_PHILOX, seed=8, step=k)
    d = store(N)
    d.fill_photons(N, 0, C_LIT, 1.0, 1.0, 8)
    sync_rows = [d.step_fused(1e-3, sc(k), (), lazy=True) for k in range(5)]
    d.fill_photons(N, 0, C_LIT, 1.0, 1.0, 8)
    rows = []
    assert d.step_fused(1e-3, sc(0), (), sync=False, lazy=True) is None
    for k in range(1, 5):
        d.step_fused(1e-3, sc(k), (), sync=False, lazy=True)      # step k enqueued ...
        rows.append(d.step_fused_read(0))                           # ... while step k-1 is read
    assert d.last_scatter_hits() == sync_rows[4]["hits"]            # most recent step, not yet read
    rows.append(d.step_fused_read(0))
    for a, b in zip(sync_rows, rows):
        assert a["hits"] == b["hits"] and list(a["sign"]) == list(b["sign"]) and a["N"] == b["N"] == N
    with pytest.raises(hip.HipError, match="no asynchronous"):
        d.step_fused_read(0)
    d.step_fused(1e-3, sc(5), (), sync=False, lazy=True)
    d.step_fused(1e-3, sc(6), (), sync=False, lazy=True)
    with pytest.raises(hip.HipError, match="outstanding"):
        d.step_fused(1e-3, sc(7), (), sync=False, lazy=True)
    d.step_fused_read(0), d.step_fused_read(0)


# ============================================================================ on-disk cache of hipRTC code objects
_RTC_CACHE_WORKER = r"""
import sys, time
import numpy as np
sys.path.insert(0, %(root)r)
from physicl_amd import _hip
d = _hip.Device(0)
d.store_alloc(1000)
d.upload_state({"v": np.tile([299792458.0, 0.0, 0.0], (1000, 1)), "E": np.full(1000, 3e-19)})
t0 = time.perf_counter()
hits = d.step_scatter_isotropic(1e-3, 1.0, _hip.SCATTER_VARIABLE_N, 299792458.0, 6.62607015e-34, %(expr)r, _hip.RNG_PHILOX, 3, 0)
print(hits, time.perf_counter() - t0)
d.close()
"""


def test_rtc_code_objects_are_cached_on_disk_and_survive_a_corrupt_file(tmp_path):
    """A second process finds the specialised kernels of an expression in $PCL_RTC_CACHE instead of recompiling them;
    a truncated cache file is ignored (recompiled and overwritten), never trusted."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    expr = "0.5 * exp(-1 * pow(r0[gid], 2) / %d.0)" % (1000 + os.getpid() % 1000)     # not compiled by anyone before
    env = dict(os.environ, PCL_RTC_CACHE=str(tmp_path))

    def run():
        out = subprocess.check_output([sys.executable, "-c", _RTC_CACHE_WORKER % {"root": root, "expr": expr}], env=env)
        h, t = out.decode().split()[-2:]
        return int(h), float(t)
    h1, t1 = run()
    files = list(tmp_path.glob("*.hsaco"))
    assert len(files) == 1 and files[0].stat().st_size > 10_000
    h2, t2 = run()
    assert h2 == h1 and t2 < 0.5 * t1, (t1, t2)                    # no compile the second time
    size = files[0].stat().st_size
    files[0].write_bytes(files[0].read_bytes()[: size // 3])        # corrupt it
    h3, _ = run()
    assert h3 == h1 and files[0].stat().st_size == size and len(list(tmp_path.glob("*"))) == 1


# ============================================================================ ScatterMeasureStep(measure_E=True)
@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("N", [1, 64, 2049, 300_001])
def test_plane_energies_match_oracle_in_particle_order(dev, hip, N, dtype):
    npdt = np.float64 if dtype == "f64" else np.float32
    rs = np.random.RandomState(N)
    r = rs.uniform(-2, 2, (N, 3)).astype(npdt)
    dr = rs.uniform(-1, 1, (N, 3)).astype(npdt)
    E = rs.uniform(1, 2, N).astype(npdt)
    kind = (rs.random_sample(N) < 0.8).astype(np.uint8)
    d = hip.Device(0)
    try:
        d.store_alloc(N, dtype)
        d.upload_state({"r": r, "dr": dr, "v": np.zeros((N, 3)), "E": E, "kind": kind})
        for loc in ([0.25, np.nan, np.nan], [np.nan, -0.5, np.nan], [np.nan, np.nan, 1.0], [0.1, 0.2, np.nan], [50.0, np.nan, np.nan]):
            want = orc.plane_crossing_energies([r[:, k].astype(np.float64) for k in range(3)],
                                               [dr[:, k].astype(np.float64) for k in range(3)], E, loc, kind)
            if dtype == "f32":                       # the device subtracts in float32
                ax = 0 if not np.isnan(loc[0]) else (1 if not np.isnan(loc[1]) else 2)
                x, L = r[:, ax], np.float32(loc[ax])
                p = x - dr[:, ax]
                want = E[(((p <= L) & (L <= x)) | ((p >= L) & (L >= x))) & (kind != 0)]
            got = d.plane_energies(loc)
            assert got.dtype == npdt and np.array_equal(got, want), (loc, len(got), len(want))
            cnt = d.step_counters([loc])
            if kind.all():
                assert cnt[hip.CNT_PLANE0] == len(want)
            assert np.array_equal(d.plane_energies(loc, n_hint=len(want)), want)
    finally:
        d.close()
