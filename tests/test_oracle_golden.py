"""Pin the CPU oracle against the golden vectors produced by running the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import physicl_oracle as orc

C_LIT = 299792458.0          # str(c)           light.py:14, 301
H_LIT = 6.62607015e-34       # str(h).upper()   light.py:15, 301

# Stated tolerance for transcendental results: <= 4 ulp of |v| = c per component.
V_ABS_TOL = 4 * np.spacing(C_LIT)


def cols(a):
    return [np.ascontiguousarray(a[:, i]) for i in range(3)]


# ------------------------------------------------------------------ G1 newton: bit-exact
@pytest.mark.parametrize("case", [0, 1, 2])
def test_newton_bit_exact(golden, case):
    z = golden("g1_newton")
    r, v = cols(z["r_init"]), cols(z["v_init"])
    dt = float(z["c%d_dt" % case])
    for k in range(1, 11):
        r, dr = orc.newton_euler(r, v, dt)
        if k in (1, 10):
            assert np.array_equal(np.stack(r, 1), z["c%d_r_after%d" % (case, k)])
            assert np.array_equal(np.stack(dr, 1), z["c%d_dr_after%d" % (case, k)])


def test_newton_config1(golden):
    z = golden("g1_newton")
    r = [np.zeros(4)] * 3
    v = [np.full(4, C_LIT), np.zeros(4), np.zeros(4)]
    for _ in range(100):
        r, dr = orc.newton_euler(r, v, 0.001)
    assert np.array_equal(np.stack(r, 1)[0], z["cfg1_r_after100"])
    assert np.array_equal(np.stack(dr, 1)[0], z["cfg1_dr_after100"])


# ------------------------------------------------------------------ G2 isotropic scatter, CL path
ISO = {
    "base": dict(use_E=False, expr=None),
    "lambda": dict(use_E=True, expr=None),
    "varn": dict(use_E=True, expr="0.000000001 * exp(r0[gid] - 5)"),
    "varn_radial": dict(use_E=False,
                        expr="2.5 * exp(-1 * (sqrt(pow(r0[gid], 2) + pow(r1[gid], 2) + pow(r2[gid], 2)) - 6.0)/(3.5))"),
    "varn_overflow": dict(use_E=True, expr="0.000000001 * exp(r0[gid] - 5)"),
}


def tie_ok(pcoll, rand, mism):
    """Hit-mask mismatches are tolerated only where pcoll and rand agree to 1e-14 relative."""
    return np.all(np.abs(pcoll[mism] - rand[mism]) <= 1e-14 * np.abs(pcoll[mism]))


@pytest.mark.parametrize("tag", sorted(ISO))
def test_iso_kernel_and_writeback(golden, tag):
    z = golden("g2_iso_" + tag)
    cfg = ISO[tag]
    K = int(z["K"])
    # quirk 1: the kernel's A is the user's n and vice versa (light.py:287)
    assert float(z["k0_A"]) == float(z["n_user"]) and float(z["k0_n"]) == float(z["A_user"])
    prev_v = None
    for k in range(K):
        g = lambda nm: z["k%d_%s" % (k, nm)]
        E = g("E") if cfg["use_E"] else None
        r = [g("r0"), g("r1"), g("r2")] if cfg["expr"] else None
        hit, r0, r1, r2 = orc.scatter_sphere_kernel(
            g("d0"), g("d1"), g("d2"), g("rtheta"), g("rphi"), g("rand"), float(g("A")), float(g("n")), C_LIT,
            h=H_LIT if cfg["use_E"] else None, E=E, n_expr=cfg["expr"], r=r, fill=-7.0)
        ref_hit = ~np.isnan(g("res0"))
        mism = hit != ref_hit
        if mism.any():
            pc = orc.scatter_pcoll(g("d0"), g("d1"), g("d2"), float(g("A")), float(g("n")), h=H_LIT, c=C_LIT,
                                   E=E, n_expr=cfg["expr"], r=r)
            assert tie_ok(pc, g("rand"), mism)
        both = hit & ref_hit
        for mine, ref in ((r0, g("res0")), (r1, g("res1")), (r2, g("res2"))):
            assert np.max(np.abs(mine[both] - ref[both]), initial=0.0) <= V_ABS_TOL
        # res1/res2 untouched on a miss (quirk 3): generator pre-filled them with -7
        miss = ~ref_hit
        assert np.all(g("res1")[miss] == -7.0) and np.all(g("res2")[miss] == -7.0)
        # host write-back (light.py:325-331) from the REFERENCE's kernel outputs: exact
        if prev_v is None:
            prev_v = [np.full(len(hit), C_LIT), np.zeros(len(hit)), np.zeros(len(hit))]
        vn, dv = orc.scatter_apply(prev_v, ref_hit, (g("res0"), g("res1"), g("res2")))
        assert np.array_equal(np.stack(vn, 1), g("post_v"))
        assert np.array_equal(np.stack(dv, 1), g("post_dv"))
        prev_v = vn
        # the kernel's d inputs are the stored dr of the Newton step (bit-exact chain)
        assert np.array_equal(np.stack([g("d0"), g("d1"), g("d2")], 1), g("post_dr"))


@pytest.mark.parametrize("tag", sorted(ISO))
def test_iso_rng_order(golden, tag):
    """Per-photon RNG order rtheta, rphi, rand from the global MT19937 stream (__init__.py:606-619)."""
    z = golden("g2_iso_" + tag)
    rs = np.random.RandomState(int(z["seed"]))
    for k in range(int(z["K"])):
        rt, rp, ra = orc.reference_draws(len(z["k%d_rand" % k]), rs)
        assert np.array_equal(rt, z["k%d_rtheta" % k])
        assert np.array_equal(rp, z["k%d_rphi" % k])
        assert np.array_equal(ra, z["k%d_rand" % k])


@pytest.mark.parametrize("tag", ["base", "varn"])
def test_iso_chain_with_newton(golden, tag):
    """Newton -> scatter chained over K steps from the initial state, using the reference's own
    kernel outputs for v (so the chain stays bit-exact): r, dr must match exactly."""
    z = golden("g2_iso_" + tag)
    n = len(z["k0_rand"])
    r = cols(z["init_r"])
    v = [np.full(n, C_LIT), np.zeros(n), np.zeros(n)]
    dt = float(z["dt"])
    for k in range(int(z["K"])):
        r, dr = orc.newton_euler(r, v, dt)
        assert np.array_equal(np.stack(r, 1), z["k%d_post_r" % k])
        assert np.array_equal(np.stack(dr, 1), z["k%d_post_dr" % k])
        v = cols(z["k%d_post_v" % k])


# ------------------------------------------------------------------ G5 counters (exact integers)
@pytest.mark.parametrize("tag", sorted(ISO))
def test_sign_and_plane_counters(golden, tag):
    z = golden("g2_iso_" + tag)
    planes = z["planes"]
    for k in range(int(z["K"])):
        v, r, dr = (cols(z["k%d_post_%s" % (k, f)]) for f in ("v", "r", "dr"))
        row = z["sign_rows"][k]
        assert row[1] == len(v[0])
        assert tuple(int(x) for x in row[2:5]) == orc.sign_counts(v)
        mrow = z["measure_rows"][k]
        for pi, loc in enumerate(planes):
            assert int(mrow[2 + pi]) == orc.plane_crossings(r, dr, loc)


# ------------------------------------------------------------------ G4 delete + stable compaction
def test_delete_flags_and_compaction(golden):
    z = golden("g4_delete")
    assert list(z["argnames_clprogram"]) == ["d0", "d1", "d2", "rand", "A", "n", "res"]
    assert list(z["argnames_reference"]) == ["dx", "dy", "dz", "rand", "n", "A", "result"]
    uid = np.arange(int(z["N"]), dtype=np.int64)
    rs = np.random.RandomState(int(z["seed"]))
    A_k, n_k = float(z["n_user"]), float(z["A_user"])  # swapped (light.py:236)
    for k in range(int(z["K"])):
        g = lambda nm: z["k%d_%s" % (k, nm)]
        assert np.array_equal(rs.random_sample(len(uid)), g("rand"))       # one draw per photon
        flags = orc.delete_flags(g("d0"), g("d1"), g("d2"), g("rand"), A_k, n_k)
        assert flags.dtype == np.int32 and np.array_equal(flags, g("flags"))
        keep = orc.survivors(flags)
        uid = uid[keep]
        assert np.array_equal(uid, g("survivor_uid"))
        assert z["measure_rows"][k][1] == len(uid)
    assert len(uid) == 0


def test_delete_full_chain(golden):
    """Newton + delete from the initial state with the reference's random stream: survivor ids,
    positions and the plane-crossing rows, all exact."""
    z = golden("g4_delete")
    n = int(z["N"])
    st = {"r": [np.zeros(n)] * 3, "v": [np.full(n, C_LIT), np.zeros(n), np.zeros(n)],
          "dr": [np.zeros(n)] * 3, "dv": [np.zeros(n)] * 3, "E": np.ones(n), "id": np.arange(n, dtype=np.int64)}
    rs = np.random.RandomState(int(z["seed"]))
    for k in range(int(z["K"])):
        orc.step_newton(st, float(z["dt"]))
        orc.step_scatter_delete(st, rs.random_sample(len(st["id"])), float(z["n_user"]), float(z["A_user"]))
        assert np.array_equal(st["id"], z["k%d_survivor_uid" % k])
        if k < 3:
            assert np.array_equal(np.stack(st["r"], 1), z["k%d_post_r" % k])
        row = z["measure_rows"][k]
        assert row[1] == len(st["id"])
        for pi, loc in enumerate(z["planes"]):
            assert int(row[2 + pi]) == orc.plane_crossings(st["r"], st["dr"], loc)
        srow = z["sign_rows"][k]
        assert tuple(int(x) for x in srow[2:5]) == orc.sign_counts(st["v"])


# ------------------------------------------------------------------ Philox known answers
def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10 (Salmon et al., SC'11 distribution)."""
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF), (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0),
         (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for ctr, key, want in kat:
        got = orc.philox4x32_10(*[np.array([x], dtype=np.uint64) for x in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == want


def test_philox_draws_ranges_and_independence():
    ids = np.arange(100000, dtype=np.int64)
    rt, rp, ra = orc.philox_draws(1234, 5, ids)
    assert 0 <= rt.min() and rt.max() < 2 * np.pi and 0 <= rp.min() and rp.max() < np.pi
    assert 0 <= ra.min() and ra.max() < 1 and abs(ra.mean() - 0.5) < 5e-3
    # keyed by id: a shard sees the same numbers as the whole
    rt2, rp2, ra2 = orc.philox_draws(1234, 5, ids[40000:60000])
    assert np.array_equal(rt2, rt[40000:60000]) and np.array_equal(ra2, ra[40000:60000])
    # different step or seed -> different stream
    assert not np.array_equal(orc.philox_draws(1234, 6, ids)[2], ra)
    assert not np.array_equal(orc.philox_draws(1235, 5, ids)[2], ra)
    # ids above 2**32 use the high counter word
    hi = orc.philox_draws(1, 0, np.array([1 << 32, 0], dtype=np.int64))[2]
    assert hi[0] != hi[1]


def test_planck_table_matches_the_reference_density():
    """Closed-form bin masses == numerical integration of the reference's planck_distribution formula
    (physicl/light.py:53-60); sampled energies follow that table."""
    import scipy.integrate
    kB, T, lo, hi = 1.380649e-23, 5778.0, 7.9e-20, 9.9e-19
    cdf, grid = orc.planck_table(lo, hi, T, 40)
    dens = lambda E: 15 / (np.pi ** 4 * kB * T) * (E / (kB * T)) ** 3 / np.e ** (E / (kB * T))
    edges = np.linspace(lo, hi, 40)
    mass = np.array([scipy.integrate.quad(dens, edges[k], edges[k + 1])[0] for k in range(39)])
    assert np.allclose(np.cumsum(mass / mass.sum()), cdf, rtol=1e-10, atol=1e-13)
    assert np.array_equal(grid, edges[:-1]) and cdf[-1] == 1.0 and np.all(np.diff(cdf) > 0)
    E = orc.philox_table_energy(3, np.arange(200000), cdf, grid)
    counts = np.array([(E == g).sum() for g in grid]) / len(E)
    assert np.max(np.abs(counts - np.diff(np.concatenate([[0.0], cdf])))) < 4e-3


# ------------------------------------------------------------------ G3 the reference's CPU paths (cl_on=False)
@pytest.mark.parametrize("tag,use_E", [("base", False), ("lambda", True), ("varn_ignored", False)])
def test_iso_py_path_bit_exact(golden, tag, use_E):
    """ScatterIsotropicStep.__run_py (light.py:335-350) restated: the fixture ran the reference unmodified under
    cl_on=False with numpy's own sin / cos / power, so everything -- hit decisions, new velocities, dv = v_old, the
    Euler moves fed by them, and the position of the np.random stream afterwards -- is bit-exact."""
    z = golden("g3_iso_py_" + tag)
    N, K, dt = len(z["init_E"]), int(z["K"]), float(z["dt"])
    U = np.random.RandomState(int(z["seed"])).random_sample(3 * N * K + 1)
    ph = z["is_photon"]
    st = {"r": [np.zeros(N) for _ in range(3)], "v": cols(z["init_v"]), "dr": [np.zeros(N) for _ in range(3)],
          "dv": [np.zeros(N) for _ in range(3)], "E": z["init_E"].copy(), "id": np.arange(N)}
    pos = 0
    for k in range(K):
        orc.step_newton(st, dt)
        before = [x.copy() for x in st["dv"]]
        hit, used = orc.step_scatter_isotropic_py(st, U[pos:], float(z["n_user"]), float(z["A_user"]), C_LIT, h=H_LIT,
                                                  use_E=use_E, is_photon=ph)
        pos += used
        for f in ("r", "v", "dr"):
            assert np.array_equal(np.stack(st[f], 1), z["k%d_post_%s" % (k, f)]), (k, f)
        dv = np.stack(st["dv"], 1)
        assert np.array_equal(dv[ph], z["k%d_post_dv" % k][ph])          # plain Objects keep their own dv (all zero here)
        assert hit[~ph].sum() == 0 and 0 < hit.sum() < ph.sum()
        assert orc.sign_counts(st["v"]) == tuple(int(x) for x in z["sign_rows"][k][2:5])
    assert U[pos] == float(z["next_draw"])                                # the stream stands where the reference left it


def test_delete_reference_py_path_skips_the_object_after_every_removal(golden):
    """ScatterDeleteStepReference.__run_py (light.py:216-223) restated, survivor ids per step and stream position."""
    z = golden("g3_delete_py")
    N, K, dt = int(z["N"]), int(z["K"]), float(z["dt"])
    U = np.random.RandomState(int(z["seed"])).random_sample(N * K + 1)
    ph_all = z["is_photon"]
    st = {"r": [np.zeros(N) for _ in range(3)], "v": cols(z["init_v"]), "dr": [np.zeros(N) for _ in range(3)],
          "dv": [np.zeros(N) for _ in range(3)], "E": np.ones(N), "id": np.arange(N)}
    pos = 0
    for k in range(K):
        orc.step_newton(st, dt)
        removed, used = orc.step_scatter_delete_reference_py(st, U[pos:], float(z["n_user"]), float(z["A_user"]),
                                                             is_photon=ph_all[st["id"]])
        pos += used
        assert np.array_equal(st["id"], z["k%d_survivor_uid" % k])
        assert np.array_equal(np.stack(st["r"], 1), z["k%d_post_r" % k])
    assert U[pos] == float(z["next_draw"])
    # the quirk itself: far fewer photons go than pcoll = 0.2998 would remove from 1818 photons in the first step
    assert N - len(z["k0_survivor_uid"]) < 0.85 * 0.2998 * ph_all.sum()


# ------------------------------------------------------------------ G5 TracePathMeasureStep's table (light.py:433-483)
def _g5_rows(z, pre):
    at = 0
    for i in range(len(z[pre + "info"])):
        n = int(z[pre + "pos_len"][i])
        yield i, z[pre + "pos"][at:at + n]
        at += n


def test_trace_table_delete_run_is_the_oracle_chain(golden):
    """What the reference traced in a delete run until empty == the positions of the oracle's Newton + delete chain under
    the same np.random stream: a row per photon that survived the first pass's delete (the step runs behind it), named by
    uid, a position for every pass it was in the list at the trace step, 3 NaN scalars per pass after that; exact."""
    z = golden("g5_trace")
    n, dt = int(z["del_N"]), float(z["del_dt"])
    st = {"r": [np.zeros(n)] * 3, "v": [np.full(n, C_LIT), np.zeros(n), np.zeros(n)],
          "dr": [np.zeros(n)] * 3, "dv": [np.zeros(n)] * 3, "E": np.ones(n), "id": np.arange(n, dtype=np.int64)}
    rs = np.random.RandomState(int(z["del_seed"]))
    seen, passes = {}, len(z["del_t_row"])
    for k in range(passes):
        orc.step_newton(st, dt)
        orc.step_scatter_delete(st, rs.random_sample(len(st["id"])), float(z["del_n_user"]), float(z["del_A_user"]))
        assert len(st["id"]) == z["del_alive"][k]
        for j, uid in enumerate(st["id"]):
            seen.setdefault(int(uid), []).append([st["r"][a][j] for a in range(3)])
    assert len(st["id"]) == 0 and np.array_equal(z["del_t_row"], np.cumsum(np.full(passes, dt)))
    order = sorted(seen)                                   # trace ids are handed out in list order at the first sight: uid order
    assert [str(x) for x in z["del_info"]] == ["photon %d" % u for u in order]
    for i, pos in _g5_rows(z, "del_"):
        assert np.array_equal(pos, np.array(seen[order[i]]))
        assert z["del_lead_scalars"][i] == 0 and z["del_trail_scalars"][i] == 3 * (passes - len(pos))


def test_trace_table_isotropic_run_is_the_oracle_chain(golden):
    """The isotropic run with trace_dv: freq == the number of passes the oracle's chain scattered the photon in (dv is the
    zero vector on a miss, light.py:331), positions within the chain's 4 ulp(c) dt per scattering before."""
    z = golden("g5_trace")
    n, dt, K = int(z["iso_N"]), float(z["iso_dt"]), int(z["iso_K"])
    st = {"r": [np.zeros(n)] * 3, "v": [np.full(n, C_LIT), np.zeros(n), np.zeros(n)],
          "dr": [np.zeros(n)] * 3, "dv": [np.zeros(n)] * 3, "E": np.ones(n), "id": np.arange(n, dtype=np.int64)}
    rs = np.random.RandomState(int(z["iso_seed"]))
    freq, pos = np.zeros(n, dtype=np.int64), []
    for k in range(K):
        orc.step_newton(st, dt)
        hit = orc.step_scatter_isotropic(st, orc.reference_draws(n, rs), float(z["iso_n_user"]), float(z["iso_A_user"]), C_LIT)
        freq += hit.astype(np.int64)
        pos.append(np.stack(st["r"], 1))
    pos = np.stack(pos, 1)                                 # (photon, pass, 3)
    assert np.array_equal(z["iso_freq"], freq) and freq.sum() > 0
    assert np.all(z["iso_pos_len"] == K) and not z["iso_lead_scalars"].any() and not z["iso_trail_scalars"].any()
    assert len(set(str(x) for x in z["iso_info"])) == 1 and str(z["iso_info"][0]) == "<class 'physicl.light.PhotonObject'>"
    for i, p in _g5_rows(z, "iso_"):
        assert np.max(np.abs(p - pos[i])) <= K * dt * 4 * np.spacing(C_LIT)
