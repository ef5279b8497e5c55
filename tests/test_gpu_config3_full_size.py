"""GPU: BASELINE configs[3] at its REAL size -- 8e8 photons -- on the one MI355X a test box has.

configs[3] shards 8e8 photons over eight GPUs by contiguous index blocks [g * 1e8, (g + 1) * 1e8): no halo,
no migration, the device RNG keyed by the GLOBAL photon id, one all-reduce of the counter rows.  Whether that partition is
right does not need eight GPUs: 8e8 photons fit ONE MI355X (109 GB slab), so the unsharded run and the eight shards -- each
filled with ``id_base = g * 1e8`` exactly as rank g of ``bench.py --gpus 8`` fills its store -- can be run on the same
device and compared at size:

* the measure rows of the unsharded store == the element-wise SUM of the eight shards' rows (what the all-reduce computes),
  every step, exactly;
* the photons of id windows at the start, across the 4e8 boundary (shard 3 | shard 4) and at the very end are bit-identical
  between the unsharded store and the shards that own them, and agree with the oracle's step-by-step chain on those ids;
* for the delete loop: the global survivor order of the unsharded compaction == the concatenation of the shards' survivor
  orders (ids and positions, bit for bit).

The reference has nothing of this (one context, one queue: physicl/__init__.py:427-429); the semantics compared are its
kernels' (physicl/light.py:303-315, 239-249, physicl/newton.py:15-16) through the oracle.
"""
import numpy as np
import pytest

from oracle import physicl_oracle as orc

pytestmark = pytest.mark.gpu

C_LIT = 299792458.0
H_LIT = 6.62607015e-34
G = 8
SHARD = 100_000_000
N_ALL = G * SHARD
W = 4096
# [first, count): the start of the store, a window astride the 4e8 boundary between shards 3 and 4, the very end
WINDOWS = [(0, W), (4 * SHARD - W // 2, W), (N_ALL - W, W)]


@pytest.fixture(scope="module")
def hip():
    from physicl_amd import _hip
    return _hip


def need_memory(d, gib):
    free, total = d.mem_info()
    if total < gib * 2**30:
        pytest.skip("device memory %.0f GiB < %.0f GiB" % (total / 2**30, gib))


def window_pieces(off, cnt):
    """The parts of the id window [off, off + cnt) by owning shard: [(shard, local offset, count)]."""
    out = []
    lo = off
    while lo < off + cnt:
        g = lo // SHARD
        hi = min(off + cnt, (g + 1) * SHARD)
        out.append((g, lo - g * SHARD, hi - lo))
        lo = hi
    return out


def test_config3_scatter_8e8_unsharded_equals_the_sum_of_eight_shards(hip):
    K, SEED, DT = 8, 1234, 5e-3
    EXPR, A_K, N_K = "0.000000001 * exp(r0[gid] - 5)", 1e-15, 1e-19          # examples/variable_n_scattering.ipynb:30,52-56
    e_lo, e_hi = H_LIT * C_LIT / 700e-9, H_LIT * C_LIT / 200e-9
    sc = dict(A=A_K, n=N_K, flags=hip.SCATTER_WAVELENGTH | hip.SCATTER_VARIABLE_N, c=C_LIT, h=H_LIT, n_expr=EXPR,
              rng_mode=hip.RNG_PHILOX, seed=SEED, step=0)
    fields = [hip.R0, hip.R0 + 1, hip.R0 + 2, hip.V0, hip.V0 + 1, hip.V0 + 2, hip.E]

    def row(o):
        return [o["N"], o["hits"]] + [int(x) for x in o["sign"]]

    with hip.Device(0) as d:
        need_memory(d, 200)
        # ---- one store of 8e8 photons
        d.store_alloc(N_ALL)
        d.fill_photons(N_ALL, 0, C_LIT, e_lo, e_hi, SEED)
        E0 = [d.download(hip.E, cnt, off) for off, cnt in WINDOWS]
        rows_all = np.array([row(o) for o in d.step_fused_multi(DT, K, sc)], dtype=np.int64)
        whole = [[d.download(f, cnt, off) for f in fields] for off, cnt in WINDOWS]
        d.store_free()
        # ---- eight stores of 1e8, ids g * 1e8 ... : what rank g of the 8-GPU job holds
        d.store_alloc(SHARD)
        rows_sum = np.zeros_like(rows_all)
        parts = {}
        for g in range(G):
            d.fill_photons(SHARD, g * SHARD, C_LIT, e_lo, e_hi, SEED)
            rows_sum += np.array([row(o) for o in d.step_fused_multi(DT, K, sc)], dtype=np.int64)
            for w, (off, cnt) in enumerate(WINDOWS):
                for (pg, loc, n) in window_pieces(off, cnt):
                    if pg == g:
                        parts[(w, g)] = [d.download(f, n, loc) for f in fields]
    # the all-reduce's result == the unsharded rows, every step, every column
    assert np.array_equal(rows_all, rows_sum), (rows_all.tolist(), rows_sum.tolist())
    assert rows_all[0, 0] == N_ALL and rows_all[:, 1].min() > 0
    ulp_c = float(np.spacing(C_LIT))
    for w, (off, cnt) in enumerate(WINDOWS):
        pieces = window_pieces(off, cnt)
        assert len(pieces) == (2 if w == 1 else 1)
        for j in range(len(fields)):                                           # bit-identical, shard by shard
            assert np.array_equal(whole[w][j], np.concatenate([parts[(w, g)][j] for g, _, _ in pieces])), (off, j)
        # ... and the oracle's chain on exactly those global ids
        ids = np.arange(off, off + cnt, dtype=np.int64)
        want_E = orc.philox_energy(SEED, ids, e_lo, e_hi)
        assert np.max(np.abs(E0[w] - want_E) / want_E) <= 4e-16
        z = lambda: np.zeros(cnt)
        st = {"r": [z(), z(), z()], "v": [np.full(cnt, C_LIT), z(), z()], "dr": [z(), z(), z()], "dv": [z(), z(), z()], "E": E0[w].copy(), "id": ids}
        hits = 0
        for k in range(K):
            orc.step_newton(st, DT)
            hits += int(orc.step_scatter_isotropic(st, orc.philox_draws(SEED, k, ids), A_K, N_K, C_LIT, h=H_LIT, use_E=True, n_expr=EXPR).sum())
        assert hits > cnt                                                      # (the window did scatter)
        r, v = np.stack(whole[w][0:3], 1), np.stack(whole[w][3:6], 1)
        r_ref, v_ref = np.stack(st["r"], 1), np.stack(st["v"], 1)
        assert np.max(np.abs(v - v_ref)) <= 4 * ulp_c, off
        slack = K * float(np.spacing(np.max(np.abs(r_ref))))
        assert np.max(np.abs(r - r_ref)) <= K * DT * 4 * ulp_c + slack + 1e-12, off


def test_config3_delete_8e8_global_survivor_order_equals_the_concatenation_of_shard_orders(hip):
    dt, A, n, seed, bodies = 1e-3, 1e-3, 1e-3, 1234, 6            # pcoll = 0.2998 per body: 0.7^6 = 11.8 % survive, several compactions
    plane = np.array([[1.0 / (1e-3 * 1e-3), np.nan, np.nan]])      # test/test_light.py:58

    def row(o):
        return [o["N"], o["removed"]] + [int(x) for x in o["sign"]] + [int(x) for x in o["planes"]]

    with hip.Device(0) as d:
        need_memory(d, 250)                                        # the unsharded store and the slab its survivors are compacted into
        d.store_alloc(N_ALL)
        d.fill_photons(N_ALL, 0, C_LIT, 1.0, 1.0, seed)
        rows_all = np.array([row(d.step_fused_delete(dt, A, n, hip.RNG_PHILOX, seed, step, plane, lazy=True)) for step in range(bodies)],
                            dtype=np.int64)
        ids_all = d.download_ids()
        x_all = d.download(hip.R0)
        vx_all = d.download(hip.V0)
        d.store_free()
        hip.pool_trim()
        d.store_alloc(SHARD)
        rows_sum = np.zeros_like(rows_all)
        ids_parts, x_parts, vx_parts = [], [], []
        for g in range(G):
            d.fill_photons(SHARD, g * SHARD, C_LIT, 1.0, 1.0, seed)
            # (the shards take the other formulation -- K bodies per call -- so the comparison also crosses the two)
            rows_sum += np.array([row(o) for o in d.step_fused_delete_multi(dt, bodies, A, n, seed, 0, plane)], dtype=np.int64)
            ids_parts.append(d.download_ids())
            x_parts.append(d.download(hip.R0))
            vx_parts.append(d.download(hip.V0))
    assert np.array_equal(rows_all, rows_sum), (rows_all.tolist(), rows_sum.tolist())
    ids_cat = np.concatenate(ids_parts)
    assert len(ids_all) == rows_all[-1, 0] == len(ids_cat)
    assert np.array_equal(ids_all, ids_cat)                        # global survivor order == concatenation of the shards' orders
    assert np.all(np.diff(ids_all) > 0)                            # ... which is ascending ids: stable
    assert np.array_equal(x_all, np.concatenate(x_parts)) and np.array_equal(vx_all, np.concatenate(vx_parts))
    for g in range(G):
        assert ids_parts[g][0] >= g * SHARD and ids_parts[g][-1] < (g + 1) * SHARD
    p = 1e-3 * 1e-3 * C_LIT * dt
    assert abs(rows_all[0, 1] - N_ALL * p) < 5 * np.sqrt(N_ALL * p * (1 - p))
    assert rows_all[:, 5].max() > 0                                # the plane at x = 1e6 was crossed (after four moves)
    # the oracle's chain on the id windows: the very photons it keeps, their positions bit for bit (IEEE mul / add only)
    for off, cnt in WINDOWS:
        ids = np.arange(off, off + cnt, dtype=np.int64)
        z = lambda: np.zeros(cnt)
        st = {"r": [z(), z(), z()], "v": [np.full(cnt, C_LIT), z(), z()], "dr": [z(), z(), z()], "dv": [z(), z(), z()], "E": np.ones(cnt), "id": ids}
        for step in range(bodies):
            orc.step_newton(st, dt)
            orc.step_scatter_delete(st, orc.philox_draws(seed, step, st["id"])[2], A, n)
        lo, hi = np.searchsorted(ids_all, off), np.searchsorted(ids_all, off + cnt)
        assert np.array_equal(ids_all[lo:hi], st["id"]) and 0 < hi - lo < cnt, off
        assert np.array_equal(x_all[lo:hi], st["r"][0]) and np.array_equal(vx_all[lo:hi], st["v"][0]), off
