"""GPU: a call the library refuses (bad argument, missing randoms) leaves the store exactly as it was -- including the
state the lazy steps keep implicit (dr = v*dt, dv = v - vprev), which a refused call must not forget."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C_LIT = 299792458.0
H_LIT = 6.62607015e-34


def test_refused_calls_do_not_change_the_store():
    from physicl_amd import _hip as hip
    N = 5000
    rs = np.random.RandomState(5)
    vdir = rs.normal(size=(N, 3))
    vdir /= np.linalg.norm(vdir, axis=1)[:, None]
    init = {"r": rs.uniform(-8, 8, (N, 3)), "v": vdir * C_LIT, "dv": rs.normal(size=(N, 3)), "E": rs.uniform(2.8e-19, 9.9e-19, N)}
    sc = dict(A=1e-3, n=1e-3, flags=0, c=C_LIT, h=H_LIT, n_expr=None, rng_mode=hip.RNG_PHILOX, seed=3, step=1)
    states = []
    for poke in (False, True):
        with hip.Device(0) as d:
            d.store_alloc(N)
            d.upload_state(init)
            o = d.step_fused(1e-3, sc, [], lazy=True)                    # leaves dr and dv implicit
            assert 0 < o["hits"] < N
            if poke:
                with pytest.raises(hip.HipError, match="upload_rand"):
                    d.step_fused_delete(1e-3, 1e-3, 1e-3, hip.RNG_INPUT, 3, 2, [], lazy=True)      # no randoms uploaded
                with pytest.raises(hip.HipError, match="rng_mode"):
                    d.step_fused_delete(1e-3, 1e-3, 1e-3, 7, 3, 2, [], lazy=True)
                with pytest.raises(hip.HipError, match="n_planes"):
                    d.step_fused_delete(1e-3, 1e-3, 1e-3, hip.RNG_PHILOX, 3, 2, [[1.0, np.nan, np.nan]] * 13, lazy=True)
                with pytest.raises(hip.HipError, match="k_steps"):
                    d.step_fused_delete_multi(1e-3, 65, 1e-3, 1e-3, 3, 2, [])
                with pytest.raises(hip.HipError, match="capacity"):
                    d.fill_photons(N + 1, 0, C_LIT, 1.0, 1.0, 3)
                with pytest.raises((hip.HipError, KeyError)):
                    d.step_mixed_multi(1e-3, 2, ("iso", "iso"), sc, None, [], 3, 2)
                with pytest.raises(hip.HipError, match="k_passes"):
                    d.step_mixed_multi(1e-3, 40, ("iso", "delete"), sc, (1e-3, 1e-3), [], 3, 2)
                assert d.count == N
            s = d.download_state()
            states.append(s)
    a, b = states
    assert np.array_equal(a["E"], b["E"]) and np.array_equal(a["id"], b["id"])
    for f in ("r", "v", "dr", "dv"):
        for k in range(3):
            assert np.array_equal(a[f][k], b[f][k]), (f, k)
    # and the implicit fields really were the lazy ones: dr = v_before * dt, dv = v - v_before on a hit
    assert np.array_equal(np.stack(a["dr"], 1), init["v"] * 1e-3)
    hit = np.any(np.stack(a["v"], 1) != init["v"], axis=1)
    assert 0 < hit.sum() < N and np.array_equal(np.stack(a["dv"], 1)[~hit], np.zeros((int((~hit).sum()), 3)))
