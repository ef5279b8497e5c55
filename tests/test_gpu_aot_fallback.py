"""GPU: variable_n_fn without hipRTC ("AOT profiles for the three expression shapes of the examples").

libphysicl_hip.so loads libhiprtc at run time (dlopen); when it is not there -- simulated with PCL_NO_RTC -- an
expression of one of the three shapes the reference's examples use
    K * exp(rA[gid] - X)                     examples/variable_n_scattering.ipynb:30
    K * exp(-1 * (sqrt(pow(r0[gid], 2) + pow(r1[gid], 2) + pow(r2[gid], 2)) - R)/(S))     presentation_example.ipynb:31
    K * exp(rA[gid] / Z)                     presentation_example_2.ipynb:41
runs on ahead-of-time kernels that take the text's literals as arguments.  Bar: every kernel family gives the SAME
bits (state and counters) as its hipRTC specialisation of the same text, fp64 and fp32; anything else fails loudly.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C_LIT = 299792458.0
H_LIT = 6.62607015e-34
SHAPES = {
    "offset": ("0.000000001 * exp(r0[gid] - 5)", 1e-15, 1e-9, True),
    "radial": ("2.5 * exp(-1 * (sqrt(pow(r0[gid], 2) + pow(r1[gid], 2) + pow(r2[gid], 2)) - 6.0)/(3.5))", 1.0, 1e-9, False),
    "scale": ("250000.0 * exp(r2[gid] / 8600.0)", 4.08e-36, 1e-5, True),       # "{} * exp(r2[gid] / {})".format(n_0, z_0)
    "offset_spaced": ("  7.5e-10*exp( r1[gid]-2.25 )", 1e-15, 1e-9, True),
}


@pytest.fixture(scope="module")
def hip():
    from physicl_amd import _hip
    return _hip


def run_everything(hip, expr, A, dt, use_e, dtype, N=30_011):
    rs = np.random.RandomState(11)
    npdt = np.float64 if dtype == "f64" else np.float32
    init = {"r": rs.uniform(-8, 8, (N, 3)).astype(npdt), "v": np.tile([C_LIT, 0.0, 0.0], (N, 1)).astype(npdt),
            "E": rs.uniform(2.8e-19, 9.9e-19, N).astype(npdt), "id_base": 1234567}
    flags = (hip.SCATTER_WAVELENGTH if use_e else 0) | hip.SCATTER_VARIABLE_N
    sc = lambda k: dict(A=A, n=1.0, flags=flags, c=C_LIT, h=H_LIT, n_expr=expr, rng_mode=hip.RNG_PHILOX, seed=5, step=k)
    log = []
    with hip.Device(0) as d:
        d.store_alloc(N, dtype)
        d.upload_state(init)
        o = d.step_fused(dt, sc(1), [], lazy=True)                                   # fast kernel
        log.append(("fast", o["hits"], list(o["sign"])))
        o = d.step_fused(dt, sc(2), [[0.5, np.nan, np.nan]], lazy=False)             # generic fused kernel
        log.append(("fused", o["hits"], list(o["sign"]), list(o["planes"])))
        log += [("multi", o["hits"], list(o["sign"])) for o in d.step_fused_multi(dt, 5, sc(3))]      # K-step kernel
        d.step_newton(dt)
        log.append(("scatter", d.step_scatter_isotropic(A, 1.0, flags, C_LIT, H_LIT, expr, hip.RNG_PHILOX, 5, 9)))   # separate step
        log += [(o["phase"], o["N"], o.get("hits", o.get("removed")), list(o["sign"]))
                for o in d.step_mixed_multi(dt, 3, ("iso", "delete"), sc(0), (1e-3, 0.4e-3 * 1e-3 / dt), (), 5, 10)]   # mixed kernel
        o = d.step_fused(dt, sc(20), [], lazy=True)                                  # fast kernel, explicit ids
        log.append(("fastg", o["hits"], list(o["sign"])))
        st = d.download_state()
        if dtype == "f64":                                                           # Level-1 kernel (the reference's ABI is fp64)
            n = len(st["E"])
            dr = [d.array(x) for x in st["dr"]]
            r = [d.array(x) for x in st["r"]]
            u = [d.array(rs.random_sample(n) * s) for s in (2 * np.pi, np.pi, 1.0)]
            E = d.array(st["E"])
            res = [d.empty(n) for _ in range(3)]
            for x in res:
                x.fill_bytes(0)
            d.k_light_scatter_step_sphere(*dr, *u, A, 1.0, E if use_e else None, r, *res, n, flags, C_LIT, H_LIT, expr)
            st["sphere"] = [x.get() for x in res]
            for x in dr + r + u + [E] + res:
                x.free()
    return log, st


@pytest.mark.parametrize("dtype", ["f64", "f32"])
@pytest.mark.parametrize("tag", sorted(SHAPES))
def test_builtin_shapes_without_hiprtc_equal_their_hiprtc_specialisation(hip, monkeypatch, tag, dtype):
    expr, A, dt, use_e = SHAPES[tag]
    monkeypatch.delenv("PCL_NO_RTC", raising=False)
    log_rtc, st_rtc = run_everything(hip, expr, A, dt, use_e, dtype)
    monkeypatch.setenv("PCL_NO_RTC", "1")
    log_aot, st_aot = run_everything(hip, expr, A, dt, use_e, dtype)
    assert log_aot == log_rtc
    assert sum(x[1] for x in log_rtc if x[0] in ("fast", "fused", "multi")) > 0          # the expressions do scatter photons
    for f in ("r", "v", "dr", "dv"):
        for k in range(3):
            assert np.array_equal(st_aot[f][k], st_rtc[f][k]), (f, k)
    assert np.array_equal(st_aot["id"], st_rtc["id"]) and np.array_equal(st_aot["E"], st_rtc["E"])
    if dtype == "f64":
        for a, b in zip(st_aot["sphere"], st_rtc["sphere"]):
            assert np.array_equal(a, b, equal_nan=True)


def test_without_hiprtc_other_expressions_and_user_kernels_fail_loudly(hip, monkeypatch):
    monkeypatch.setenv("PCL_NO_RTC", "1")
    with hip.Device(0) as d:
        d.store_alloc(100)
        d.upload_state({"v": np.tile([C_LIT, 0.0, 0.0], (100, 1)), "E": np.ones(100)})
        for expr in ("0.5 * exp(r0[gid] - 5) + 1.0", "exp(r0[gid])", "2.0 * exp(d0[gid] - 1.0)"):
            with pytest.raises(hip.ExpressionError, match="needs hipRTC"):
                d.step_scatter_isotropic(1e-3, 1.0, hip.SCATTER_VARIABLE_N, C_LIT, H_LIT, expr, hip.RNG_PHILOX, 1, 1)
        with pytest.raises(hip.ExpressionError):                                       # still validated first
            d.step_scatter_isotropic(1e-3, 1.0, hip.SCATTER_VARIABLE_N, C_LIT, H_LIT, "system(1)", hip.RNG_PHILOX, 1, 1)
        with pytest.raises(hip.HipError, match="need hipRTC"):
            d.user_kernel("k", [("double", "x", True)], "x[get_global_id(0)] = 1.0;")
        # constant-n steps do not care
        assert d.step_scatter_isotropic(1e-3, 1e-3, 0, C_LIT, H_LIT, None, hip.RNG_PHILOX, 1, 1) >= 0
