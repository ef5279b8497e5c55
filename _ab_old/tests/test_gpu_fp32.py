"""GPU: the fp32 instantiations (BASELINE.json configs[4], "fp32 vs fp64 tolerance sweep").

The reference is fp64-only, so there is no reference vector for fp32: the checker is the oracle's
float32 restatement (oracle/physicl_oracle.py, dtype=np.float32).  Bars: bit-exact for Newton r/dr,
delete flags, survivor ids and compacted state; <= 4 ulp(float32) of |v| = c for scattered velocities;
hit-mask mismatches only where |pcoll - rand| <= 1e-6 * pcoll; and, against the fp64 path on the same
photons and the same random stream, the errors stated in test_fp32_vs_fp64_sweep."""
import numpy as np
import pytest

from oracle import physicl_oracle as orc

pytestmark = pytest.mark.gpu

C_LIT, H_LIT = 299792458.0, 6.62607015e-34
F32 = np.float32
V_TOL32 = 4 * float(np.spacing(F32(C_LIT)))          # 128 m/s = 4.3e-7 relative
EXPR_EX = "0.000000001 * exp(r0[gid] - 5)"


@pytest.fixture(scope="module")
def hip():
    from physicl_amd import _hip
    return _hip


@pytest.fixture()
def dev(hip):
    d = hip.Device(0)
    yield d
    d.close()


def cols32(a):
    return [np.ascontiguousarray(a[:, i]).astype(F32) for i in range(3)]


@pytest.mark.parametrize("N", [1, 3, 4, 5, 1023, 100_003])
def test_newton_fp32_bit_exact(dev, N):
    rs = np.random.RandomState(N)
    r, v = (rs.normal(size=(N, 3)) * 1e5).astype(F32), (rs.normal(size=(N, 3)) * 1e8).astype(F32)
    dev.store_alloc(N, "f32")
    assert dev.np_dtype == np.float32
    dev.upload_state({"r": r, "v": v, "E": np.ones(N)})
    rr, vv = cols32(r), cols32(v)
    for _ in range(3):
        dev.step_newton(1.25e-4)
        rr, dr = orc.newton_euler(rr, vv, 1.25e-4, F32)
    s = dev.download_state()
    assert s["r"][0].dtype == np.float32
    for k in range(3):
        assert np.array_equal(s["r"][k], rr[k]) and np.array_equal(s["dr"][k], dr[k])


@pytest.mark.parametrize("N", [1, 2049, 300_001])
def test_delete_fp32_bit_exact(dev, hip, N):
    rs = np.random.RandomState(N)
    st = {"r": cols32(rs.normal(size=(N, 3))), "v": cols32(rs.normal(size=(N, 3)) * 1e8),
          "dr": [np.zeros(N, F32)] * 3, "dv": cols32(rs.normal(size=(N, 3))), "E": rs.uniform(1, 2, N).astype(F32),
          "id": np.arange(N, dtype=np.int64) + 9}
    dev.store_alloc(N, "f32")
    dev.upload_state({"r": np.stack(st["r"], 1), "v": np.stack(st["v"], 1), "dv": np.stack(st["dv"], 1), "E": st["E"],
                      "id_base": 9})
    for step in range(4):
        dev.step_newton(1e-3)
        orc.step_newton(st, 1e-3, F32)
        alive, removed = dev.step_scatter_delete(2e-3, 1e-3, hip.RNG_PHILOX, 5, step)
        n_before = len(st["id"])
        flags, keep = orc.step_scatter_delete(st, orc.philox_draws(5, step, st["id"], F32)[2], 2e-3, 1e-3, F32)
        assert (alive, removed) == (len(keep), n_before - len(keep))
        assert np.array_equal(dev.last_delete_flags(n_before), flags)
        s = dev.download_state()
        assert np.array_equal(s["id"], st["id"]) and np.array_equal(s["E"], st["E"])
        for f in ("r", "v", "dr", "dv"):
            for k in range(3):
                assert np.array_equal(s[f][k], st[f][k]), (f, k)
        if alive == 0:
            break


@pytest.mark.parametrize("tag", ["base", "lambda", "varn"])
@pytest.mark.parametrize("path", ["separate", "fused", "lazy"])
def test_scatter_fp32_vs_oracle(dev, hip, tag, path):
    N = 200_003
    rs = np.random.RandomState(len(tag) + len(path))
    use_E, expr = tag != "base", (EXPR_EX if tag == "varn" else None)
    A_k, n_k, dt = {"base": (1e-3, 1e-3, 1e-3), "lambda": (1e-15, 1e-19, 5e-3), "varn": (1e-15, 1e-19, 1e-9)}[tag]
    st = {"r": cols32(rs.uniform(-10, 10, (N, 3))), "v": [np.full(N, C_LIT, F32), np.zeros(N, F32), np.zeros(N, F32)],
          "dr": [np.zeros(N, F32)] * 3, "dv": [np.zeros(N, F32)] * 3, "E": rs.uniform(2.8e-19, 9.9e-19, N).astype(F32),
          "id": np.arange(N, dtype=np.int64) + (1 << 35)}
    dev.store_alloc(N, "f32")
    dev.upload_state({"r": np.stack(st["r"], 1), "v": np.stack(st["v"], 1), "E": st["E"], "id_base": 1 << 35})
    flags = (hip.SCATTER_WAVELENGTH if use_E else 0) | (hip.SCATTER_VARIABLE_N if expr else 0)
    for step in range(3):
        sc = dict(A=A_k, n=n_k, flags=flags, c=C_LIT, h=H_LIT, n_expr=expr, rng_mode=hip.RNG_PHILOX, seed=77, step=step)
        if path == "separate":
            dev.step_newton(dt)
            hits = dev.step_scatter_isotropic(A_k, n_k, flags, C_LIT, H_LIT, expr, hip.RNG_PHILOX, 77, step)
            sign = dev.step_counters()[1:4]
        else:
            o = dev.step_fused(dt, sc, (), lazy=(path == "lazy"))
            hits, sign = o["hits"], o["sign"]
        orc.step_newton(st, dt, F32)
        draws = orc.philox_draws(77, step, st["id"], F32)
        v_before = np.stack(st["v"], 1)
        pc = orc.scatter_pcoll(*st["dr"], A_k, n_k, h=H_LIT, c=C_LIT, E=st["E"] if use_E else None, n_expr=expr,
                               r=st["r"] if expr else None, dtype=F32)
        hit = orc.step_scatter_isotropic(st, draws, A_k, n_k, C_LIT, h=H_LIT, use_E=use_E, n_expr=expr, dtype=F32)
        s = dev.download_state()
        v_dev, dv_dev = np.stack(s["v"], 1), np.stack(s["dv"], 1)
        hit_dev = np.any(dv_dev != 0, axis=1) | np.any(v_dev != v_before, axis=1)
        mism = hit_dev != hit
        assert np.all(np.abs(pc[mism] - draws[2][mism]) <= 1e-6 * np.abs(pc[mism])), mism.sum()
        assert mism.mean() < 1e-4 and hits == hit_dev.sum()
        ok = ~mism
        assert np.max(np.abs(v_dev[ok].astype(np.float64) - np.stack(st["v"], 1)[ok])) <= V_TOL32
        assert np.max(np.abs(dv_dev[ok].astype(np.float64) - np.stack(st["dv"], 1)[ok])) <= 2 * V_TOL32
        for k in range(3):
            assert np.array_equal(s["r"][k], st["r"][k]) and np.array_equal(s["dr"][k], st["dr"][k])
        assert list(sign) == [int((v_dev[:, k] > 0).sum()) for k in range(3)]
        st["v"] = [np.ascontiguousarray(v_dev[:, k]) for k in range(3)]     # keep both chains on the same inputs


def test_fp32_paths_agree_bit_for_bit(dev, hip):
    """separate == fused-eager == fused-lazy (fast path) in fp32 too."""
    N = 100_001
    rs = np.random.RandomState(1)
    init = {"r": rs.uniform(-8, 8, (N, 3)), "v": np.tile([C_LIT, 0.0, 0.0], (N, 1)), "E": rs.uniform(2.8e-19, 9.9e-19, N),
            "id_base": 123}
    out = {}
    for path in ("separate", "fused", "lazy"):
        dev.store_alloc(N, "f32")
        dev.upload_state(init)
        log = []
        for step in range(4):
            sc = dict(A=1e-15, n=1e-19, flags=3, c=C_LIT, h=H_LIT, n_expr=EXPR_EX, rng_mode=hip.RNG_PHILOX, seed=3, step=step)
            if path == "separate":
                dev.step_newton(1e-9)
                hits = dev.step_scatter_isotropic(1e-15, 1e-19, 3, C_LIT, H_LIT, EXPR_EX, hip.RNG_PHILOX, 3, step)
                log.append((hits, list(dev.step_counters()[1:4])))
            else:
                o = dev.step_fused(1e-9, sc, (), lazy=(path == "lazy"))
                log.append((o["hits"], list(o["sign"])))
        out[path] = (log, dev.download_state())
    for path in ("fused", "lazy"):
        assert out[path][0] == out["separate"][0]
        for f in ("r", "v", "dr", "dv"):
            for k in range(3):
                assert np.array_equal(out[path][1][f][k], out["separate"][1][f][k]), (path, f, k)


def test_fp32_vs_fp64_sweep(dev, hip):
    """configs[4] in small: step list [Newton, ScatterIsotropic(base), Newton, ScatterDelete], same photons
    and the same Philox stream in both precisions.  Hit/delete decisions may differ only in the 2^-24
    sliver between the two uniforms (plus fp32 rounding of pcoll); where they agree, positions agree to
    fp32 rounding accumulated over the steps."""
    N, K = 1_000_000, 10
    res = {}
    for dt_name in ("f64", "f32"):
        dev.store_alloc(N, dt_name)
        dev.fill_photons(N, 0, C_LIT, 2.84e-19, 9.93e-19, 11)
        hist = []
        for k in range(K):
            o = dev.step_fused(1e-3, dict(A=1e-3, n=1e-3, flags=0, c=C_LIT, h=H_LIT, rng_mode=hip.RNG_PHILOX, seed=11,
                                          step=2 * k), (), lazy=True)
            dev.step_newton(1e-3)
            alive, removed = dev.step_scatter_delete(1e-4, 1e-3, hip.RNG_PHILOX, 11, 2 * k + 1)
            hist.append((o["hits"], alive))
        res[dt_name] = (hist, dev.download_ids(), np.stack([dev.download(hip.R0 + k) for k in range(3)], 1).astype(np.float64),
                        np.stack([dev.download(hip.V0 + k) for k in range(3)], 1).astype(np.float64))
    h64, id64, r64, v64 = res["f64"]
    h32, id32, r32, v32 = res["f32"]
    # decisions: fewer than 1e-5 of all hit / delete decisions differ
    for (a, b), (c_, d) in zip(h64, h32):
        assert abs(a - c_) <= max(20, 1e-5 * N) and abs(b - d) <= max(20, 1e-5 * N)
    common, i64, i32 = np.intersect1d(id64, id32, return_indices=True)
    assert len(common) > 0.99 * min(len(id64), len(id32))
    same_path = np.max(np.abs(v64[i64] - v32[i32]), axis=1) <= 4 * V_TOL32       # same scatter history
    assert same_path.mean() > 0.9999
    a, b = r64[i64][same_path], r32[i32][same_path]
    rel = np.linalg.norm(a - b, axis=1) / np.linalg.norm(a, axis=1)
    assert rel.max() < 2e-6 * K, rel.max()        # 2K Euler adds of fp32-rounded terms; |r| ~ 1e6 m, ulp32 ~ 0.1 m
