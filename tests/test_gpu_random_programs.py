"""GPU: randomly generated step programs, every formulation against the plain one and against the oracle.

The store carries a good deal of implicit state between launches -- dr = v*dt and dv = v - vprev left unwritten by the
lazy steps, the knowledge that the dv rows hold nothing but +0.0, the two slabs a compaction alternates between,
explicit ids after the first removal, the per-photon wavelength cache -- and every entry point has to hand that state
over correctly to every other one.  The hand-written chains of test_gpu_multi.py / test_gpu_mixed.py cover the
transitions somebody thought of; this file draws programs at random:

  a program   = a list of loop bodies [Newton, ScatterIsotropic | ScatterDelete | nothing, counters]
                (physicl/__init__.py:512-516 runs the steps in order; light.py:231-260, 281-331, 374-431), interleaved
                with partial downloads and host uploads of single field groups;
  plain run   = one launch per Step: pcl_step_newton, pcl_step_scatter_isotropic (eager) / pcl_step_scatter_delete,
                pcl_step_counters -- the kernels pinned to the reference's goldens in test_gpu_parity.py;
  grouped run = the same program cut at random into pcl_step_fused (lazy or eager), pcl_step_fused_delete,
                pcl_step_fused_multi, pcl_step_fused_delete_multi and pcl_step_mixed_multi launches.

Bars: rows (alive, hits | removed, sign counts, plane crossings) equal and the whole state r, v, dr, dv, E, id, kind
BIT-identical between the two runs, fp64 and fp32, uniform / explicit-id / mixed-kind stores; for all-photon fp64
stores the plain run is also compared with the numpy oracle stepping the same program: rows equal, survivor ids
identical, v within 4 ulp(c), r within (moves) * dt * 4 ulp(c).
"""
import os

import numpy as np
import pytest

from oracle import physicl_oracle as orc

pytestmark = pytest.mark.gpu

C_LIT = 299792458.0
H_LIT = 6.62607015e-34
V_ABS_TOL = 4 * np.spacing(C_LIT)
CASES = {
    # tag: (use_E, expr, A, n, dt)           kernel constants as in test_gpu_mixed.py
    "base": (False, None, 1e-3, 1e-3, 1e-3),
    "lambda": (True, None, 1e-15, 1e-19, 5e-3),
    "varn": (True, "0.000000001 * exp(r0[gid] - 5)", 1e-15, 1e-19, 1e-9),
}
A_DEL, N_DEL = 1e-3, 0.4e-3            # pcoll ~ 0.12 per delete body at dt = 1e-3 (scaled with 1e-3 / dt below)
GROUPS = ("r", "v", "dr", "dv")


@pytest.fixture(scope="module")
def hip():
    from physicl_amd import _hip
    return _hip


def initial(N, dtype, rs, store):
    npdt = np.float64 if dtype == "f64" else np.float32
    vdir = rs.normal(size=(N, 3))
    vdir /= np.linalg.norm(vdir, axis=1)[:, None]
    st = {"r": rs.uniform(-8, 8, (N, 3)).astype(npdt), "v": (vdir * C_LIT).astype(npdt),
          "dr": np.zeros((N, 3), npdt), "dv": (rs.normal(size=(N, 3)) if rs.random_sample() < 0.5 else np.zeros((N, 3))).astype(npdt),
          "E": rs.uniform(2.8e-19, 9.9e-19, N).astype(npdt), "id_base": 3_000_000_001}
    if store in ("ids", "both"):
        st["id"] = np.sort(rs.choice(50 * N + 100, N, replace=False)).astype(np.int64) + (1 << 33)
    if store in ("kinds", "both"):
        st["kind"] = (rs.random_sample(N) < 0.85).astype(np.uint8)
    return st


def make_program(rs, length):
    """('body', 'iso' | 'delete' | 'newton') | ('download', group) | ('upload', group, salt)"""
    prog = []
    while len(prog) < length:
        u = rs.random_sample()
        if u < 0.30:                               # a run of scatter bodies
            prog += [("body", "iso")] * rs.randint(1, 6)
        elif u < 0.50:
            prog += [("body", "delete")] * rs.randint(1, 5)
        elif u < 0.70:                             # the configs[4] loop, either order
            pair = [("body", "iso"), ("body", "delete")][::rs.choice([1, -1])]
            prog += pair * rs.randint(1, 4)
        elif u < 0.78:
            prog.append(("body", "newton"))
        elif u < 0.90:
            prog.append(("download", GROUPS[rs.randint(4)]))
        else:
            prog.append(("upload", GROUPS[rs.randint(1, 4)], int(rs.randint(1 << 30))))
    return prog


def upload_values(group, salt, n, npdt):
    rs = np.random.RandomState(salt)
    if group == "v":
        d = rs.normal(size=(n, 3))
        return (d / np.linalg.norm(d, axis=1)[:, None] * C_LIT).astype(npdt)
    if group == "dv":
        return np.zeros((n, 3), npdt) if salt % 3 == 0 else rs.normal(size=(n, 3)).astype(npdt)
    return rs.normal(size=(n, 3)).astype(npdt) * 1e-3          # dr


def scatter_dict(hip, tag, seed, step):
    use_e, expr, A, n, dt = CASES[tag]
    flags = (hip.SCATTER_WAVELENGTH if use_e else 0) | (hip.SCATTER_VARIABLE_N if expr else 0)
    return dict(A=A, n=n, flags=flags, c=C_LIT, h=H_LIT, n_expr=expr, rng_mode=hip.RNG_PHILOX, seed=seed, step=step)


def do_side_op(hip, d, op, log):
    npdt = d.np_dtype
    if op[0] == "download":
        log.append(("download", op[1], [d.download(f, d.count) for f in hip.FIELD_GROUPS[op[1]]] if d.count else []))
    else:
        vals = upload_values(op[1], op[2], d.count, npdt)
        for k, f in enumerate(hip.FIELD_GROUPS[op[1]]):
            if d.count:
                d.upload(f, np.ascontiguousarray(vals[:, k]))


def run_plain(hip, d, prog, tag, seed, planes):
    dt = CASES[tag][4]
    A_d, n_d = A_DEL, N_DEL * 1e-3 / dt
    log, step = [], 1
    for op in prog:
        if op[0] != "body":
            do_side_op(hip, d, op, log)
            continue
        d.step_newton(dt)
        second = 0
        if op[1] == "iso":
            sc = scatter_dict(hip, tag, seed, step)
            second = d.step_scatter_isotropic(sc["A"], sc["n"], sc["flags"], C_LIT, H_LIT, sc["n_expr"], hip.RNG_PHILOX, seed, step)
        elif op[1] == "delete":
            second = d.step_scatter_delete(A_d, n_d, hip.RNG_PHILOX, seed, step)[1]
        c = d.step_counters(planes)
        log.append((op[1], int(c[hip.CNT_N]), int(second), [int(x) for x in c[1:4]], [int(x) for x in c[4:4 + len(planes)]]))
        step += 1
    return log


def run_grouped(hip, d, prog, tag, seed, planes, rs):
    """The same program, cut into launches of randomly chosen formulations."""
    dt = CASES[tag][4]
    dele = (A_DEL, N_DEL * 1e-3 / dt)
    log, step, i, used = [], 1, 0, set()
    while i < len(prog):
        op = prog[i]
        if op[0] != "body":
            do_side_op(hip, d, op, log)
            i += 1
            continue
        kinds = []
        while i + len(kinds) < len(prog) and prog[i + len(kinds)][0] == "body":
            kinds.append(prog[i + len(kinds)][1])
        first = kinds[0]
        run = 1
        while run < len(kinds) and kinds[run] == first:
            run += 1
        pairs = 0
        if len(kinds) >= 2 and first != "newton" and kinds[1] not in (first, "newton"):
            while 2 * pairs + 1 < len(kinds) and kinds[2 * pairs] == first and kinds[2 * pairs + 1] == kinds[1]:
                pairs += 1
        sc = scatter_dict(hip, tag, seed, step)
        choice = rs.random_sample()
        if first == "newton":
            o = d.step_fused(dt, None, planes, lazy=bool(rs.randint(2)))
            log.append(("newton", o["N"], 0, [int(x) for x in o["sign"]], [int(x) for x in o["planes"]]))
            used.add("fused-newton")
            n_done = 1
        elif pairs and choice < 0.6:
            K = int(rs.randint(1, pairs + 1))
            phases = (first, kinds[1])
            rows = d.step_mixed_multi(dt, K, phases, sc, dele, planes, seed, step)
            log += [(o["phase"], o["N"], o["hits"] if o["phase"] == "iso" else o["removed"], [int(x) for x in o["sign"]],
                     [int(x) for x in o["planes"]]) for o in rows]
            used.add("mixed-pair")
            n_done = 2 * K
        elif choice < 0.8 or run > 1 and choice < 0.9:
            K = int(rs.randint(1, run + 1))
            if first == "iso" and d.is_uniform() and rs.random_sample() < 0.6:
                rows = d.step_fused_multi(dt, K, sc, planes)
                log += [("iso", o["N"], o["hits"], [int(x) for x in o["sign"]], [int(x) for x in o["planes"]]) for o in rows]
                used.add("fused_multi")
            elif first == "delete" and rs.random_sample() < 0.6:
                rows = d.step_fused_delete_multi(dt, K, dele[0], dele[1], seed, step, planes)
                log += [("delete", o["N"], o["removed"], [int(x) for x in o["sign"]], [int(x) for x in o["planes"]]) for o in rows]
                used.add("fused_delete_multi")
            else:
                rows = d.step_mixed_multi(dt, K, (first,), sc if first == "iso" else None, dele if first == "delete" else None, planes, seed, step)
                log += [(first, o["N"], o["hits"] if first == "iso" else o["removed"], [int(x) for x in o["sign"]],
                         [int(x) for x in o["planes"]]) for o in rows]
                used.add("mixed-" + first)
            n_done = K
        else:
            lazy = bool(rs.randint(2))
            if first == "iso":
                o = d.step_fused(dt, sc, planes, lazy=lazy)
                log.append(("iso", o["N"], o["hits"], [int(x) for x in o["sign"]], [int(x) for x in o["planes"]]))
            else:
                o = d.step_fused_delete(dt, dele[0], dele[1], hip.RNG_PHILOX, seed, step, planes, lazy=lazy)
                log.append(("delete", o["N"], o["removed"], [int(x) for x in o["sign"]], [int(x) for x in o["planes"]]))
            used.add("fused-%s-%s" % (first, "lazy" if lazy else "eager"))
            n_done = 1
        step += n_done
        i += n_done
    return log, used


def run_oracle(prog, init, tag, seed, planes):
    use_e, expr, A, n, dt = CASES[tag]
    A_d, n_d = A_DEL, N_DEL * 1e-3 / dt
    N = len(init["E"])
    st = {g: [np.ascontiguousarray(init[g][:, k]).astype(np.float64) for k in range(3)] for g in GROUPS}
    st["E"] = init["E"].astype(np.float64)
    st["id"] = init["id"].copy() if init.get("id") is not None else init["id_base"] + np.arange(N, dtype=np.int64)
    log, step, moves = [], 1, 0
    for op in prog:
        if op[0] == "download":
            log.append(("download", op[1], [a.copy() for a in st[op[1]]] if len(st["E"]) else []))
            continue
        if op[0] == "upload":
            vals = upload_values(op[1], op[2], len(st["E"]), np.float64)
            st[op[1]] = [np.ascontiguousarray(vals[:, k]) for k in range(3)]
            continue
        orc.step_newton(st, dt)
        moves += 1
        second = 0
        if op[1] == "iso" and len(st["E"]):
            hit = orc.step_scatter_isotropic(st, orc.philox_draws(seed, step, st["id"]), A, n, C_LIT, h=H_LIT, use_E=use_e, n_expr=expr)
            second = int(hit.sum())
        elif op[1] == "delete" and len(st["E"]):
            flags, _ = orc.step_scatter_delete(st, orc.philox_draws(seed, step, st["id"])[2], A_d, n_d)
            second = int(flags.sum())
        log.append((op[1], len(st["E"]), second, list(orc.sign_counts(st["v"])), [orc.plane_crossings(st["r"], st["dr"], p) for p in planes]))
        step += 1
    return log, st, moves


def full_state(d):
    if d.count == 0:
        return None
    s = d.download_state()
    s["kind"] = d.download_kind(d.count)
    return s


def assert_logs_identical(a, b):
    assert len(a) == len(b)
    for k, (x, y) in enumerate(zip(a, b)):
        if x[0] == "download":
            assert y[0] == "download" and x[1] == y[1] and len(x[2]) == len(y[2]), k
            for p, q in zip(x[2], y[2]):
                assert np.array_equal(p, q), (k, x[1])
        else:
            assert x == y, (k, x, y)


N_SEEDS = int(os.environ.get("PCL_RANDOM_SEEDS", "30"))      # a soak run sets this to hundreds
PROGRAMS = [(seed, dtype, store) for seed in range(N_SEEDS) for dtype, store in
            ((("f64", "uniform"), ("f32", "ids"), ("f64", "both")) if seed % 2 else (("f64", "uniform"), ("f32", "uniform"), ("f64", "kinds")))]


@pytest.mark.parametrize("seed,dtype,store", PROGRAMS)
def test_random_program_every_formulation_equals_one_launch_per_step(hip, seed, dtype, store):
    rs = np.random.RandomState(1000 + seed)
    N = int([1, 63, 129, 2049, 4100, 20_011, 70_001][seed % 7])
    tag = sorted(CASES)[seed % 3]
    planes = [[[0.5, np.nan, np.nan]], [], [[np.nan, -1.0, np.nan], [np.nan, np.nan, 2.0]]][seed % 3]
    init = initial(N, dtype, rs, store)
    prog = make_program(rs, 32)
    a, b = hip.Device(0), hip.Device(0)
    try:
        a.store_alloc(N, dtype)
        b.store_alloc(N, dtype)
        a.upload_state(init)
        b.upload_state(init)
        plain = run_plain(hip, a, prog, tag, 77 + seed, planes)
        grouped, used = run_grouped(hip, b, prog, tag, 77 + seed, planes, rs)
        assert_logs_identical(plain, grouped)
        sa, sb = full_state(a), full_state(b)
        assert (sa is None) == (sb is None)
        if sa is not None:
            assert np.array_equal(sa["E"], sb["E"]) and np.array_equal(sa["id"], sb["id"]) and np.array_equal(sa["kind"], sb["kind"])
            for g in GROUPS:
                for k in range(3):
                    assert np.array_equal(sa[g][k], sb[g][k]), (g, k, sorted(used))
        if dtype == "f64" and store == "uniform" and N <= 20_011:
            ref, st, moves = run_oracle(prog, init, tag, 77 + seed, planes)
            dt = CASES[tag][4]
            rtol = 2 * max(moves, 1) * dt * V_ABS_TOL + (moves + 4) * np.spacing(8.0 + moves * dt * C_LIT)   # one rounding per move on top
            assert len(ref) == len(plain)
            for k, (x, y) in enumerate(zip(ref, plain)):
                if x[0] == "download":
                    tol = {"r": rtol, "v": V_ABS_TOL, "dv": 2 * V_ABS_TOL, "dr": dt * V_ABS_TOL + 1e-18}[x[1]]
                    for p, q in zip(x[2], y[2]):
                        assert p.shape == q.shape and (p.size == 0 or np.max(np.abs(p - q)) <= tol), (k, x[1])
                else:
                    assert x == y, (k, x, y)
            if sa is not None:
                assert np.array_equal(sa["id"], st["id"])
                assert np.max(np.abs(np.stack(sa["v"]) - np.stack(st["v"]))) <= V_ABS_TOL
                assert np.max(np.abs(np.stack(sa["r"]) - np.stack(st["r"]))) <= rtol
            else:
                assert len(st["E"]) == 0
    finally:
        a.close()
        b.close()


def test_the_random_programs_reach_every_formulation(hip):
    """The generator is only worth its run time if the cuts it draws exercise every entry point."""
    seen = set()
    for seed in range(14):
        rs = np.random.RandomState(1000 + seed)
        N = 2049
        init = initial(N, "f64", rs, "uniform")
        prog = make_program(rs, 32)
        d = hip.Device(0)
        try:
            d.store_alloc(N)
            d.upload_state(init)
            seen |= run_grouped(hip, d, prog, "base", 5, [], rs)[1]
        finally:
            d.close()
    assert {"fused_multi", "fused_delete_multi", "mixed-pair", "mixed-iso", "mixed-delete", "fused-iso-lazy", "fused-iso-eager",
            "fused-delete-lazy", "fused-delete-eager", "fused-newton"} <= seen, sorted(seen)
