"""Driver entry points: build() compiles every native piece; smoke() runs one tiny step on cuda:0."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def build():
    """Compile libphysicl_hip.so for gfx950 (hipcc cross-compiles without a GPU), the oracle's C
    restatement, and import the package."""
    from physicl_amd import build as b
    lib = b.build_lib()
    assert os.path.exists(lib)
    mk = os.path.join(ROOT, "oracle", "Makefile")
    if os.path.exists(mk):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    import physicl_amd  # noqa: F401
    from physicl_amd import _hip
    _hip.load()
    return lib


def smoke():
    """One small invocation of the hot path on device 0 (Newton + variable-n scatter + delete +
    counters), checked against the CPU oracle."""
    import numpy as np
    from oracle import physicl_oracle as orc
    from physicl_amd import _hip

    C, H = 299792458.0, 6.62607015e-34
    expr = "0.000000001 * exp(r0[gid] - 5)"
    N, seed = 20000, 5
    rs = np.random.RandomState(0)
    r = rs.uniform(-10, 10, (N, 3))
    E = rs.uniform(2.8e-19, 9.9e-19, N)
    st = {"r": [r[:, k].copy() for k in range(3)], "v": [np.full(N, C), np.zeros(N), np.zeros(N)],
          "dr": [np.zeros(N)] * 3, "dv": [np.zeros(N)] * 3, "E": E, "id": np.arange(N, dtype=np.int64)}
    with _hip.Device(0) as d:
        d.store_alloc(N)
        d.upload_state({"r": r, "v": np.stack(st["v"], 1), "E": E})
        d.step_newton(1e-9)
        orc.step_newton(st, 1e-9)
        hits = d.step_scatter_isotropic(1e-15, 1e-19, 3, C, H, expr, _hip.RNG_PHILOX, seed, 0)
        hit = orc.step_scatter_isotropic(st, orc.philox_draws(seed, 0, st["id"]), 1e-15, 1e-19, C, h=H, use_E=True,
                                         n_expr=expr)
        s = d.download_state()
        assert abs(hits - int(hit.sum())) <= 1, (hits, int(hit.sum()))
        assert np.max(np.abs(np.stack(s["v"], 1) - np.stack(st["v"], 1))) <= 1e-6
        assert np.array_equal(np.stack(s["r"], 1), np.stack(st["r"], 1))
        d.step_newton(1e-3)
        orc.step_newton(st, 1e-3)
        # dr now depends on the scattered v (<= 4 ulp apart): compare the delete on the device's own dr
        st["dr"] = [d.download(_hip.DR0 + k) for k in range(3)]
        alive, removed = d.step_scatter_delete(1e-3, 1e-3, _hip.RNG_PHILOX, seed, 1)
        flags, keep = orc.step_scatter_delete(st, orc.philox_draws(seed, 1, st["id"])[2], 1e-3, 1e-3)
        assert alive == len(keep) and np.array_equal(d.download_ids(), st["id"])
        cnt = d.step_counters([[0.0, np.nan, np.nan]])
        assert cnt[0] == alive
        # the production path: the whole loop body as one kernel (dr/dv implicit), then the fused delete pipeline;
        # checked against the separate-step oracle chain on the survivors
        st["dr"] = [d.download(_hip.DR0 + k) for k in range(3)]
        st["r"] = [d.download(_hip.R0 + k) for k in range(3)]
        st["v"] = [d.download(_hip.V0 + k) for k in range(3)]
        st["E"] = d.download(_hip.E)
        o = d.step_fused(1e-9, dict(A=1e-15, n=1e-19, flags=3, c=C, h=H, n_expr=expr, rng_mode=_hip.RNG_PHILOX,
                                    seed=seed, step=2), (), lazy=True)
        orc.step_newton(st, 1e-9)
        hit2 = orc.step_scatter_isotropic(st, orc.philox_draws(seed, 2, st["id"]), 1e-15, 1e-19, C, h=H, use_E=True,
                                          n_expr=expr)
        assert abs(o["hits"] - int(hit2.sum())) <= 1 and o["N"] == alive
        assert np.array_equal(d.download(_hip.R0), st["r"][0])
        assert np.max(np.abs(d.download(_hip.V0) - st["v"][0])) <= 1e-6
        o2 = d.step_fused_delete(1e-3, 1e-3, 1e-3, _hip.RNG_PHILOX, seed, 3, [[0.0, np.nan, np.nan]], lazy=True)
        assert o2["N"] + o2["removed"] == alive and np.all(np.diff(d.download_ids()) > 0)
        print("smoke ok: N=%d hits=%d alive=%d fused hits=%d then alive=%d device=%s"
              % (N, hits, alive, o["hits"], o2["N"], d.info()["name"]))


if __name__ == "__main__":
    build()
    if "--smoke" in sys.argv:
        smoke()
