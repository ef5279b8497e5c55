"""CPU: the C/OpenMP restatement (oracle/c) against the numpy oracle (which is pinned to the reference)."""
import numpy as np
import pytest

from oracle import c_oracle as co
from oracle import physicl_oracle as orc

C, H = 299792458.0, 6.62607015e-34
TOL = 4 * np.spacing(C)


def make_state(N, seed):
    rs = np.random.RandomState(seed)
    return {"r": [np.ascontiguousarray(x) for x in rs.uniform(-10, 10, (3, N))],
            "v": [np.full(N, C), np.zeros(N), np.zeros(N)], "dr": [np.zeros(N) for _ in range(3)],
            "dv": [np.zeros(N) for _ in range(3)], "E": rs.uniform(2.8e-19, 9.9e-19, N),
            "id": np.arange(N, dtype=np.int64) + (1 << 33)}


def copy_state(st):
    return {k: ([a.copy() for a in v] if isinstance(v, list) else v.copy()) for k, v in st.items()}


def test_philox_words_match_numpy_oracle():
    ids = np.array([0, 1, 2, (1 << 32) + 5, (1 << 40) + 123456789], dtype=np.int64)
    w = co.philox_words(ids, 7, 1, 0x1234567890ABCDEF)
    ref = orc.philox4x32_10(ids.astype(np.uint64) & np.uint64(0xFFFFFFFF), ids.astype(np.uint64) >> np.uint64(32),
                            np.uint64(7), np.uint64(1), 0x90ABCDEF, 0x12345678)
    assert np.array_equal(w, np.stack(ref, 1))


def test_newton_bit_exact():
    st = make_state(10007, 1)
    st["v"] = [np.ascontiguousarray(x) for x in np.random.RandomState(2).normal(size=(3, 10007)) * 1e8]
    ref = copy_state(st)
    for _ in range(4):
        co.newton(st, 1.25e-4)
        orc.step_newton(ref, 1.25e-4)
    for k in range(3):
        assert np.array_equal(st["r"][k], ref["r"][k]) and np.array_equal(st["dr"][k], ref["dr"][k])


def test_delete_flags_and_compaction_bit_exact(golden):
    z = golden("g4_delete")
    for k in range(int(z["K"])):
        g = lambda nm: z["k%d_%s" % (k, nm)]
        flags = co.delete_flags([g("d0"), g("d1"), g("d2")], g("rand"), float(z["n_user"]), float(z["A_user"]))
        assert np.array_equal(flags, g("flags"))
        assert np.array_equal(co.compact_indices(flags), orc.survivors(flags))


@pytest.mark.parametrize("use_E,profile", [(0, 0), (1, 0), (1, 1)])
def test_scatter_isotropic_philox_vs_numpy_oracle(use_E, profile):
    N, seed = 50000, 77
    st = make_state(N, 3)
    dt = 1e-9 if profile else (5e-3 if use_E else 1e-3)
    A, n = (1e-15, 1e-19) if use_E else (1e-3, 1e-3)
    expr = "0.000000001 * exp(r0[gid] - 5)" if profile else None
    ref = copy_state(st)
    for step in range(3):
        co.newton(st, dt)
        orc.step_newton(ref, dt)
        hits = co.scatter_isotropic(st, A, n, C, H, use_E, profile, 0.000000001, 5.0, seed, step, ids=st["id"])
        pc = orc.scatter_pcoll(*ref["dr"], A, n, h=H, c=C, E=ref["E"] if use_E else None, n_expr=expr,
                               r=ref["r"] if expr else None)
        draws = orc.philox_draws(seed, step, ref["id"])
        hit = orc.step_scatter_isotropic(ref, draws, A, n, C, h=H, use_E=bool(use_E), n_expr=expr)
        mine = np.any(np.stack(st["dv"], 1) != 0, axis=1)
        mism = mine != hit
        assert np.all(np.abs(pc[mism] - draws[2][mism]) <= 1e-14 * np.abs(pc[mism]))
        assert abs(hits - hit.sum()) <= mism.sum()
        ok = ~mism
        for k in range(3):
            assert np.max(np.abs(st["v"][k][ok] - ref["v"][k][ok])) <= TOL
            assert np.max(np.abs(st["dv"][k][ok] - ref["dv"][k][ok])) <= 2 * TOL
        ref["v"] = [a.copy() for a in st["v"]]      # keep the chains on identical inputs


def test_scatter_isotropic_input_randoms_vs_reference(golden):
    z = golden("g2_iso_varn")
    N = len(z["k0_rand"])
    st = {"r": [np.ascontiguousarray(z["k0_post_r"][:, k]) for k in range(3)],
          "dr": [np.ascontiguousarray(z["k0_post_dr"][:, k]) for k in range(3)],
          "v": [np.full(N, C), np.zeros(N), np.zeros(N)], "dv": [np.zeros(N) for _ in range(3)], "E": z["init_E"].copy()}
    co.scatter_isotropic(st, float(z["k0_A"]), float(z["k0_n"]), C, H, 1, 1, 0.000000001, 5.0,
                         draws=(z["k0_rtheta"], z["k0_rphi"], z["k0_rand"]))
    assert np.max(np.abs(np.stack(st["v"], 1) - z["k0_post_v"])) <= TOL
    assert np.max(np.abs(np.stack(st["dv"], 1) - z["k0_post_dv"])) <= 2 * TOL


def test_counters_match():
    st = make_state(20011, 9)
    st["v"] = [np.ascontiguousarray(x) for x in np.random.RandomState(4).normal(size=(3, 20011))]
    co.newton(st, 0.5)
    out = co.counters(st, 1, 0.25)
    assert tuple(out[:3]) == orc.sign_counts(st["v"])
    assert out[3] == orc.plane_crossings(st["r"], st["dr"], [np.nan, 0.25, np.nan])


@pytest.mark.parametrize("step0", [0, 3])
def test_delete_chain_is_the_numpy_oracle_body_after_body(step0):
    """orc_delete_chain (the body that removes each photon of a delete run) against Newton + step_scatter_delete of the numpy
    oracle applied body after body: the survivor ids after every body, in order."""
    N, K, seed = 30_011, 14, 99
    rs = np.random.RandomState(4)
    ang = rs.uniform(0, np.pi, N)
    v = [np.ascontiguousarray(C * np.cos(ang)), np.ascontiguousarray(C * np.sin(ang)), np.zeros(N)]
    ids = np.arange(N, dtype=np.int64) + (1 << 32) - 7
    death = co.delete_chain(v, 1e-3, 1e-3, 0.7e-3, seed, step0, K, ids=ids)
    death_base = co.delete_chain(v, 1e-3, 1e-3, 0.7e-3, seed, step0, K, id_base=int(ids[0]))
    assert np.array_equal(death, death_base)
    st = {"r": [np.zeros(N) for _ in range(3)], "v": [x.copy() for x in v], "dr": [np.zeros(N) for _ in range(3)],
          "dv": [np.zeros(N) for _ in range(3)], "E": np.ones(N), "id": ids.copy()}
    for k in range(K):
        orc.step_newton(st, 1e-3)
        orc.step_scatter_delete(st, orc.philox_draws(seed, step0 + k, st["id"])[2], 1e-3, 0.7e-3)
        assert np.array_equal(st["id"], ids[death > k])
    assert 0 < (death == K).sum() < N
    a, b = ids[death == K], ids[death == K][::-1]
    assert co.order_checksum(a) != co.order_checksum(b) and co.order_checksum(a) == co.order_checksum(a.copy())
