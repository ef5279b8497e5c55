"""GPU at BASELINE.json's full sizes against the C oracle over EVERY photon (oracle/c/physicl_oracle.c, pinned to the numpy
oracle -- itself pinned to the reference's fixtures -- by tests/test_oracle_c.py).

tests/test_gpu_full_size_oracle.py compares id windows (4 096 ids) with the oracle and whole-store rows between device
formulations; here the oracle itself runs the whole store:

* configs[2] (1e8 photons, variable-n + wavelength scatter, the example's constants), 8 steps: every row [N, hits, xp, yp, zp] of
  the K-steps-per-launch kernel (k_multi) and of the one-launch-per-step kernel (k_fast) == the C oracle's row over all 1e8
  ids; v of all 1e8 photons within 4 ulp(c), r within 8 * dt * 4 ulp(c).  Reference semantics: physicl/light.py:303-331, 414-431.
* configs[1](ii) (delete until empty, A = n = 1e-3, dt = 1e-3) at 1e7 and 1e8: the alive count after EVERY body, the survivor
  ids (whole arrays, and an order-sensitive checksum sum(id * (position + 1)) mod 2^64) at three bodies, the number of bodies
  until the store is empty -- one call per body and 16 bodies per call.  Reference semantics: physicl/light.py:239-260,
  physicl/__init__.py:455-459 (stable removal).
"""
import numpy as np
import pytest

from oracle import c_oracle as co

pytestmark = pytest.mark.gpu

C_LIT = 299792458.0
H_LIT = 6.62607015e-34
V_ABS_TOL = 4 * np.spacing(C_LIT)
EXPR = "0.000000001 * exp(r0[gid] - 5)"
E_LO, E_HI = H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9


@pytest.fixture(scope="module")
def hip():
    from physicl_amd import _hip
    return _hip


@pytest.fixture()
def dev(hip):
    d = hip.Device(0)
    yield d
    d.close()
    hip.pool_trim()


def test_config2_rows_of_the_whole_store_equal_the_c_oracle_over_all_1e8_ids(dev, hip):
    N, K, dt, seed = 100_000_000, 8, 0.005, 1234
    A, n = 1e-15, 1e-19                                    # kernel constants after the reference's swap (light.py:287)
    sc = dict(A=A, n=n, flags=hip.SCATTER_WAVELENGTH | hip.SCATTER_VARIABLE_N, c=C_LIT, h=H_LIT, n_expr=EXPR,
              rng_mode=hip.RNG_PHILOX, seed=seed, step=0)
    dev.store_alloc(N)
    dev.fill_photons(N, 0, C_LIT, E_LO, E_HI, seed)
    st = {"r": [np.zeros(N) for _ in range(3)], "v": [np.full(N, C_LIT), np.zeros(N), np.zeros(N)],
          "dr": [np.zeros(N) for _ in range(3)], "dv": [np.zeros(N) for _ in range(3)], "E": dev.download(hip.E)}
    co.set_threads(co.usable_cores())
    ref = []
    for k in range(K):
        co.newton(st, dt)
        hits = co.scatter_isotropic(st, A, n, C_LIT, H_LIT, 1, 1, 0.000000001, 5.0, seed, k, ids=None, id_base=0)
        sign = co.counters(st)
        ref.append([N, int(hits), int(sign[0]), int(sign[1]), int(sign[2])])
    assert ref[0][1] == N and 0 < ref[-1][1] < N            # exp() saturates: everybody scatters in step 0, fewer and fewer later
    # K steps per launch
    rows = dev.step_fused_multi(dt, K, sc)
    got = [[o["N"], o["hits"]] + [int(x) for x in o["sign"]] for o in rows]
    assert got == ref
    for k, (fv, fr) in enumerate(zip((hip.V0, hip.V1, hip.V2), (hip.R0, hip.R1, hip.R2))):
        v = dev.download(fv)
        assert np.max(np.abs(v - st["v"][k])) <= V_ABS_TOL
        del v
        r = dev.download(fr)
        assert np.max(np.abs(r - st["r"][k])) <= K * dt * V_ABS_TOL + 4 * np.spacing(np.max(np.abs(st["r"][k])))
        del r
    # one launch per step (k_fast)
    dev.fill_photons(N, 0, C_LIT, E_LO, E_HI, seed)
    got = []
    for k in range(K):
        o = dev.step_fused(dt, dict(sc, step=k), (), lazy=True)
        got.append([o["N"], o["hits"]] + [int(x) for x in o["sign"]])
    assert got == ref
    v0 = dev.download(hip.V0)
    assert np.max(np.abs(v0 - st["v"][0])) <= V_ABS_TOL


@pytest.mark.parametrize("N", [10_000_000, 100_000_000])
def test_delete_until_empty_alive_counts_and_survivor_order_equal_the_c_oracle(dev, hip, N):
    dt, A, n, seed, K = 1e-3, 1e-3, 1e-3, 4321, 64
    co.set_threads(co.usable_cores())
    death = co.delete_chain([np.full(N, C_LIT), np.zeros(N), np.zeros(N)], dt, A, n, seed, 0, K, id_base=0)
    alive_after = N - np.cumsum(np.bincount(death, minlength=K + 1))[:K]        # photons left after body k
    bodies_until_empty = int(np.argmax(alive_after == 0)) + 1
    assert alive_after[-1] == 0 and 30 < bodies_until_empty <= K
    checkpoints = (2, 9, 19)
    ids_all = np.arange(N, dtype=np.int64)
    # one call per loop body (the alive mask, bodies worked out ahead of their calls, compactions when half of the slots are dead)
    dev.store_alloc(N)
    dev.fill_photons(N, 0, C_LIT, 1.0, 1.0, seed)
    plane = [[1.0 / (A * n), np.nan, np.nan]]
    k = 0
    while True:
        o = dev.step_fused_delete(dt, A, n, hip.RNG_PHILOX, seed, k, plane, lazy=True)
        assert o["N"] == alive_after[k], k
        if k in checkpoints:
            have = dev.download_ids()
            want = ids_all[death > k]
            assert np.array_equal(have, want)
            assert co.order_checksum(have) == co.order_checksum(want)
        k += 1
        if o["N"] == 0:
            break
    assert k == bodies_until_empty
    # sixteen bodies per call
    dev.fill_photons(N, 0, C_LIT, 1.0, 1.0, seed)
    rows, k = [], 0
    while k < bodies_until_empty:
        rows += [o["N"] for o in dev.step_fused_delete_multi(dt, 16, A, n, seed, k, plane)]
        if k + 16 == 32:
            have = dev.download_ids()
            assert np.array_equal(have, ids_all[death > 31]) and co.order_checksum(have) == co.order_checksum(ids_all[death > 31])
        k += 16
    assert rows[:bodies_until_empty] == alive_after[:bodies_until_empty].tolist()


def test_tracked_subset_of_a_1e8_photon_store_over_32_steps_equals_the_oracle_chain(dev, hip):
    """SURVEY 8(f)-4 at full size: the positions pcl_store_trace_ahead works out for ids {0 .. 999, a window across a tile
    boundary, a window across the 2^26 boundary, the last 64} of the 1e8-photon example workload over 32 steps == the numpy
    oracle's chain on those ids (same decisions: the NaN-free rows' dv flags are the oracle's hits; positions within
    32 * dt * 4 ulp(c) + the rounding of r at 1e7 m), and the K-pass launch that follows leaves them where the last row says.
    Reference semantics: physicl/light.py:447-458 (what is traced), 303-331 (the step that moves them)."""
    from oracle import physicl_oracle as orc
    N, K, dt, seed = 100_000_000, 32, 0.005, 1234
    A, n = 1e-15, 1e-19
    sc = dict(A=A, n=n, flags=hip.SCATTER_WAVELENGTH | hip.SCATTER_VARIABLE_N, c=C_LIT, h=H_LIT, n_expr=EXPR, rng_mode=hip.RNG_PHILOX,
              seed=seed, step=0)
    dev.store_alloc(N)
    dev.fill_photons(N, 0, C_LIT, E_LO, E_HI, seed)
    ids = np.unique(np.concatenate([np.arange(1000), np.arange(2040, 2056), np.arange((1 << 26) - 8, (1 << 26) + 8), np.arange(N - 64, N)]))
    E = np.concatenate([dev.download(hip.E, int(hi - lo + 1), int(lo)) for lo, hi in
                        ((0, 999), (2040, 2055), ((1 << 26) - 8, (1 << 26) + 7), (N - 64, N - 1))])
    rows = dev.trace_ahead(ids, dt, K, ("iso",), 0, sc, None, seed, 0)
    m = len(ids)
    st = {"r": [np.zeros(m) for _ in range(3)], "v": [np.full(m, C_LIT), np.zeros(m), np.zeros(m)], "dr": [np.zeros(m) for _ in range(3)],
          "dv": [np.zeros(m) for _ in range(3)], "E": E.copy(), "id": ids.copy()}
    assert not np.isnan(rows).any()
    for k in range(K):
        orc.step_newton(st, dt)
        hit = orc.step_scatter_isotropic(st, orc.philox_draws(seed, k, st["id"]), A, n, C_LIT, h=H_LIT, use_E=True, n_expr=EXPR)
        want = np.stack(st["r"], 1)
        assert np.max(np.abs(rows[k, :, :3] - want)) <= (k + 1) * dt * V_ABS_TOL + (k + 1) * np.spacing(np.max(np.abs(want)))
        assert np.array_equal(rows[k, :, 3] != 0, np.any(np.stack(st["dv"], 1) != 0, axis=1)) and int(hit.sum()) == int((rows[k, :, 3] != 0).sum())
    got = dev.step_fused_multi(dt, K, sc)
    assert got[0]["hits"] == N
    for lo, cnt in ((0, 1000), (2040, 16), ((1 << 26) - 8, 16), (N - 64, 64)):
        sel = (ids >= lo) & (ids < lo + cnt)
        for k, f in enumerate((hip.R0, hip.R1, hip.R2)):
            assert np.array_equal(dev.download(f, cnt, lo), rows[-1, sel, k])
