"""CPU oracle for the per-particle time-step hot path of PhysiCL.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm, written from the maths of the
reference (file:line citations on every function; paths relative to /root/reference).  It is the
*checker* for the HIP kernels: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  Nothing under ``physicl_amd/`` imports it and
there is no code path by which the product falls back to it.

Parity pin: PINNED.  ``tests/test_oracle_golden.py`` checks every function below against the
golden vectors in ``tests/golden/*.npz``, which were produced by running the reference itself
(``tests/golden/make_golden.py``: reference CPU paths unmodified; reference OpenCL paths with the
reference's generated kernel text compiled by gcc against glibc libm).  Integer / IEEE add, mul,
sqrt results are pinned bit-exactly; transcendental results (sin, cos, exp, pow) are pinned to the
stated ulp tolerance only, because the reference itself leaves them to whichever OpenCL device
maths library is present ("Third-party arithmetic").

Conventions
-----------
* All particle data is structure-of-arrays, float64, one 1-D array per component.
* Arithmetic follows the reference kernel text *left to right, unfused*: ``pow(x, 2)`` is ``x*x``
  (what every optimising OpenCL/C compiler emits, verified for gcc), no FMA contraction.
"""
import math
import re

import numpy as np

TWO_PI_FACTOR = 2  # rtheta = random() * 2 * np.pi   (light.py:285)

# ----------------------------------------------------------------------------------------------
# a1  NewtonianKinematicsStep.run                                   newton.py:10-16
# ----------------------------------------------------------------------------------------------
def newton_euler(r, v, dt, dtype=np.float64):
    """Explicit Euler.  ``dr = v*dt`` is rounded and stored, then ``r = r + dr`` (newton.py:15-16).

    r, v: sequences of three arrays.  Returns (r_new[3], dr[3]).  ``dtype=np.float32`` restates the
    same arithmetic in single precision (the reference itself is fp64-only); every function below that
    takes ``dtype`` does the same -- it is what the fp32 device kernels are compared with bit for bit.
    """
    dt = dtype(dt)
    dr = [np.multiply(np.asarray(vc, dtype=dtype), dt) for vc in v]
    rn = [np.add(np.asarray(rc, dtype=dtype), drc) for rc, drc in zip(r, dr)]
    return rn, dr


# ----------------------------------------------------------------------------------------------
# kernel maths shared by a2 / a4 / a5
# ----------------------------------------------------------------------------------------------
def step_norm(d0, d1, d2, dtype=np.float64):
    """``sqrt(pow(d0,2) + pow(d1,2) + pow(d2,2))`` (light.py:149, 241, 305), left to right."""
    d0, d1, d2 = (np.asarray(x, dtype=dtype) for x in (d0, d1, d2))
    return np.sqrt((d0 * d0 + d1 * d1) + d2 * d2)


# ----------------------------------------------------------------------------------------------
# a4 / a5  delete-flag kernels                          light.py:146-158 and light.py:239-249
# ----------------------------------------------------------------------------------------------
def delete_flags(d0, d1, d2, rand, A, n, dtype=np.float64):
    """``pcoll = A * n * norm; flag = pcoll >= rand ? 1 : 0`` -> int32 (OpenCL ``int``).

    ``A`` and ``n`` are the KERNEL arguments, i.e. after the reference's swap (light.py:236).
    """
    pcoll = (dtype(A) * dtype(n)) * step_norm(d0, d1, d2, dtype)
    return (pcoll >= np.asarray(rand, dtype=dtype)).astype(np.int32)


def survivors(flags):
    """Stable compaction: indices (ascending) of the particles that stay.

    The reference removes flagged photons from ``sim.objects`` in ascending index order with
    ``list.remove`` (light.py:258-260, __init__.py:455-459), so survivors keep their order.
    """
    return np.flatnonzero(np.asarray(flags) == 0).astype(np.int64)


def compact(arrays, flags):
    """Apply ``survivors`` to every array in ``arrays``."""
    keep = survivors(flags)
    return [np.asarray(a)[keep] for a in arrays]


# ----------------------------------------------------------------------------------------------
# a2  kernel light_scatter_step_sphere                                 light.py:299-315
# ----------------------------------------------------------------------------------------------
_ALLOWED_FUNCS = {
    "exp": np.exp, "sqrt": np.sqrt, "pow": np.power, "log": np.log, "sin": np.sin, "cos": np.cos,
    "fabs": np.fabs, "fmin": np.fmin, "fmax": np.fmax, "tanh": np.tanh, "log10": np.log10,
    "exp2": np.exp2, "log2": np.log2,
}
_ALLOWED_ARRAYS = ("r0", "r1", "r2", "d0", "d1", "d2", "E")
_TOKEN = re.compile(r"\s*(?:(\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?)|([A-Za-z_]\w*)|(.))")


def eval_n_expr(expr, arrays, dtype=np.float64):
    """Evaluate an OpenCL-C ``variable_n_fn`` expression (light.py:299) with numpy.

    ``arrays`` maps the kernel's array names (r0, r1, r2, d0.., E) to float64 arrays; ``x[gid]``
    selects the work-item's element.  Integer literals behave as in C (``-1 * x`` etc. promote to
    double because an operand is double; the examples never divide two integer literals).
    """
    for num, ident, other in _TOKEN.findall(expr):
        if ident and ident not in _ALLOWED_FUNCS and ident not in _ALLOWED_ARRAYS and ident != "gid":
            raise ValueError("identifier %r not allowed in variable_n_fn" % ident)
        if other and other not in "+-*/()[], ":
            raise ValueError("character %r not allowed in variable_n_fn" % other)
    env = dict(_ALLOWED_FUNCS)
    env["gid"] = slice(None)
    for k in _ALLOWED_ARRAYS:
        if k in arrays and arrays[k] is not None:
            env[k] = np.asarray(arrays[k], dtype=dtype)
    with np.errstate(all="ignore"):
        # numpy >= 2 (NEP 50): Python literals are "weak", so with float32 arrays the whole expression stays
        # float32 -- the device's fp32 spelling puts an f suffix on every floating literal for the same effect
        val = eval(expr, {"__builtins__": {}}, env)  # noqa: S307  (token-checked above)
    n = len(next(v for v in env.values() if isinstance(v, np.ndarray)))
    return np.broadcast_to(np.asarray(val, dtype=dtype), (n,)).copy()


def scatter_pcoll(d0, d1, d2, A, n, *, h=None, c=None, E=None, n_expr=None, r=None, dtype=np.float64):
    """Collision probability exactly as the generated kernel text multiplies it (light.py:299-306).

    ``pcoll = A * n * norm``                                  (base)
    ``pcoll = A * (<n_expr>) * norm``                         (variable_n: kernel arg ``n`` unused)
    ``... * pow((h * c) / E[gid], -4)``                       (wavelength_dep_scattering)
    Multiplication is left to right.  ``h``/``c`` are the literals pasted into the source
    (``str(h).upper()``, ``str(c)``; light.py:301), i.e. their code-unit values.
    """
    norm = step_norm(d0, d1, d2, dtype)
    with np.errstate(all="ignore"):
        if n_expr is None:
            p = (dtype(A) * dtype(n)) * norm
        else:
            arrs = {"d0": d0, "d1": d1, "d2": d2, "E": E}
            if r is not None:
                arrs.update(r0=r[0], r1=r[1], r2=r[2])
            p = (dtype(A) * eval_n_expr(n_expr, arrs, dtype)) * norm
        if E is not None and h is not None:
            hc = dtype(h) * dtype(c)
            p = p * np.power(hc / np.asarray(E, dtype=dtype), dtype(-4.0))
    return p


def scatter_sphere_kernel(d0, d1, d2, rtheta, rphi, rand, A, n, c, *, h=None, E=None, n_expr=None,
                          r=None, fill=np.nan, dtype=np.float64):
    """Kernel ``light_scatter_step_sphere`` (light.py:303-315).

    Returns (hit mask, res0, res1, res2).  On a miss ``res0`` is NaN and ``res1``/``res2`` are left
    untouched by the reference (uninitialised device memory); they are returned as ``fill``.
    """
    pcoll = scatter_pcoll(d0, d1, d2, A, n, h=h, c=c, E=E, n_expr=n_expr, r=r, dtype=dtype)
    hit = pcoll >= np.asarray(rand, dtype=dtype)
    c = dtype(c)
    rtheta, rphi = np.asarray(rtheta, dtype=dtype), np.asarray(rphi, dtype=dtype)
    st, ct = np.sin(rtheta), np.cos(rtheta)
    sp, cp = np.sin(rphi), np.cos(rphi)
    res0 = np.where(hit, (c * st) * cp, np.nan)
    res1 = np.where(hit, (c * st) * sp, fill)
    res2 = np.where(hit, c * ct, fill)
    return hit, res0, res1, res2


def scatter_apply(v, hit, res, dtype=np.float64):
    """Host write-back of the OpenCL path (light.py:325-331): hit -> ``v = res``, ``dv = v - vold``;
    miss -> ``dv = 0``.  Returns (v_new[3], dv[3])."""
    vn, dv = [], []
    for vc, rc in zip(v, res):
        vc = np.asarray(vc, dtype=dtype)
        new = np.where(hit, rc, vc).astype(dtype)
        vn.append(new)
        dv.append(np.where(hit, new - vc, dtype(0.0)).astype(dtype))
    return vn, dv


def reference_draws(n, random_state=None):
    """Per-photon host RNG consumption of the OpenCL path: three ``np.random.random()`` per photon in
    the order rtheta, rphi, rand (``prep_metadata`` order light.py:285,291; gather loop
    __init__.py:606-619).  ``rtheta = u*2*pi``, ``rphi = u*pi`` evaluated left to right."""
    rs = np.random if random_state is None else random_state
    u = rs.random_sample((n, 3))
    return u[:, 0] * 2 * np.pi, u[:, 1] * np.pi, u[:, 2].copy()


# ----------------------------------------------------------------------------------------------
# a3 / a6  the reference's CPU paths of the light steps (``cl_on=False``)
#          ScatterIsotropicStep.__run_py light.py:335-350, ScatterDeleteStepReference.__run_py light.py:216-223
# Literal per-object loops: the order in which np.random is consumed depends on the outcome for every earlier object.
# ``U`` is the np.random stream from where the step finds it (e.g. RandomState(seed).random_sample(big)); every
# function returns how many numbers it consumed.
# ----------------------------------------------------------------------------------------------
def norm_py(d0, d1, d2):
    """``np_lin.norm(obj.dr)`` (light.py:220, 339): sqrt(dot(x, x)) through BLAS -- NOT the kernels' left-to-right
    form; the two disagree in the last bit for ~11 % of vectors, which only matters when
    ``pcoll`` and ``rand`` tie."""
    return float(np.linalg.norm(np.array([d0, d1, d2], dtype=np.float64)))


def step_scatter_isotropic_py(state, U, n_user, A_user, c, *, h=None, use_E=False, is_photon=None):
    """One ScatterIsotropicStep.__run_py pass over the objects of ``state`` (in place).  Per PhotonObject:
    ``p = n * A * |dr|`` [``*= ((h*c)/E) ** -4``], one draw; on a hit phi = draw * pi, THEN theta = draw * pi * 2,
    ``v = c (sin t cos p, sin t sin p, cos t)`` and ``dv = v_old`` (sic, light.py:346-348); on a miss ``dv = 0``.
    ``variable_n`` does not exist on this path (light.py:334).  Returns (hit mask, numbers consumed)."""
    N = len(state["E"])
    v = [np.array(x, dtype=np.float64) for x in state["v"]]
    dv = [np.array(x, dtype=np.float64) for x in state["dv"]]
    hit = np.zeros(N, dtype=bool)
    pos = 0
    nA = np.float64(n_user) * np.float64(A_user)
    for i in range(N):
        if is_photon is not None and not is_photon[i]:
            continue
        p = nA * norm_py(state["dr"][0][i], state["dr"][1][i], state["dr"][2][i])
        if use_E:
            p = p * np.power((np.float64(h) * np.float64(c)) / np.float64(state["E"][i]), -4)
        u = U[pos]
        pos += 1
        if p >= u:
            phi = U[pos] * np.pi
            theta = U[pos + 1] * np.pi * 2
            pos += 2
            old = (v[0][i], v[1][i], v[2][i])
            v[0][i] = c * np.sin(theta) * np.cos(phi)
            v[1][i] = c * np.sin(theta) * np.sin(phi)
            v[2][i] = c * np.cos(theta)
            dv[0][i], dv[1][i], dv[2][i] = old
            hit[i] = True
        else:
            dv[0][i] = dv[1][i] = dv[2][i] = 0.0
    state["v"], state["dv"] = v, dv
    return hit, pos


def step_scatter_delete_reference_py(state, U, n_user, A_user, is_photon=None):
    """One ScatterDeleteStepReference.__run_py pass: ``for obj in sim.objects: ... sim.remove_obj(obj)`` removes from
    the list being iterated, so the object behind every removed photon is skipped -- not tested, no draw.  ``state``
    is compacted in place.  Returns (removed mask over the incoming objects, numbers consumed)."""
    N = len(state["id"])
    removed = np.zeros(N, dtype=bool)
    nA = np.float64(n_user) * np.float64(A_user)
    pos, i = 0, 0
    while i < N:
        if is_photon is None or is_photon[i]:
            p = nA * norm_py(state["dr"][0][i], state["dr"][1][i], state["dr"][2][i])
            u = U[pos]
            pos += 1
            if p >= u:
                removed[i] = True
                i += 1                      # the list shrank under the iterator: the next object is never visited
        i += 1
    keep = np.flatnonzero(~removed)
    for f in ("r", "v", "dr", "dv"):
        state[f] = [np.asarray(a)[keep] for a in state[f]]
    for f in ("E", "id"):
        if f in state and state[f] is not None:
            state[f] = np.asarray(state[f])[keep]
    return removed, pos


# ----------------------------------------------------------------------------------------------
# counters (#10, #11)                                   light.py:414-431 and light.py:374-404
# ----------------------------------------------------------------------------------------------
def sign_counts(v):
    """(#v_x>0, #v_y>0, #v_z>0) -- strict, zero is not positive (light.py:415, 424-426)."""
    return tuple(int(np.count_nonzero(np.asarray(vc) > 0)) for vc in v)


def plane_crossings(r, dr, loc):
    """Photons whose last move crossed the plane ``loc`` (light.py:385-399).

    ``loc`` has NaN in the coordinates that do not define the plane; the first non-NaN of x, y
    decides the axis, else z.  Crossing test ``(r-dr <= L <= r) or (r-dr >= L >= r)``."""
    ax = 0 if not math.isnan(loc[0]) else (1 if not math.isnan(loc[1]) else 2)
    L = np.float64(loc[ax])
    x = np.asarray(r[ax], dtype=np.float64)
    p = x - np.asarray(dr[ax], dtype=np.float64)
    return int(np.count_nonzero(((p <= L) & (L <= x)) | ((p >= L) & (L >= x))))


def plane_crossing_energies(r, dr, E, loc, kind=None):
    """``ScatterMeasureStep(measure_E=True)``: the energies of the photons whose last move crossed ``loc``, in object
    order (light.py:383-399; plain Objects carry no E and are left out)."""
    ax = 0 if not math.isnan(loc[0]) else (1 if not math.isnan(loc[1]) else 2)
    L = np.float64(loc[ax])
    x = np.asarray(r[ax], dtype=np.float64)
    p = x - np.asarray(dr[ax], dtype=np.float64)
    m = ((p <= L) & (L <= x)) | ((p >= L) & (L >= x))
    if kind is not None:
        m &= np.asarray(kind) != 0
    return np.asarray(E)[m]


# ----------------------------------------------------------------------------------------------
# device RNG mode (NEW functionality; defines what the HIP Philox path must reproduce bit-exactly)
# Philox4x32-10: Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11.
# ----------------------------------------------------------------------------------------------
PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
_U32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  Inputs broadcastable uint32-valued arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & _U32 for x in np.broadcast_arrays(c0, c1, c2, c3))
    k0 = np.uint64(int(k0) & 0xFFFFFFFF)
    k1 = np.uint64(int(k1) & 0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(PHILOX_M0) * c0
        p1 = np.uint64(PHILOX_M1) * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _U32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _U32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & _U32, lo1, (hi0 ^ c3 ^ k1) & _U32, lo0
        k0 = (k0 + np.uint64(PHILOX_W0)) & _U32
        k1 = (k1 + np.uint64(PHILOX_W1)) & _U32
    return tuple(x.astype(np.uint32) for x in (c0, c1, c2, c3))


def u53(a, b):
    """Two uint32 words -> double in [0,1) with 53 random bits: ``((a>>5)*2**26 + (b>>6)) / 2**53``
    (the recipe numpy/MT19937 uses for ``random()``, applied to Philox words)."""
    a = np.asarray(a, dtype=np.uint64) >> np.uint64(5)
    b = np.asarray(b, dtype=np.uint64) >> np.uint64(6)
    return (a * np.uint64(67108864) + b).astype(np.float64) * (1.0 / 9007199254740992.0)


def philox_draws(seed, step, ids, dtype=np.float64):
    """Device-RNG equivalent of ``reference_draws``: keyed by (seed, step, global particle id), so the
    stream of a photon does not depend on how particles are sharded over GPUs or compacted.

    key = (seed_lo, seed_hi)
    decision block  counter (id_lo, id_hi, step >> 1, 0): rand of an even step = u53(w0,w1), of an odd step = u53(w2,w3)
                    (one block serves the hit decisions of two consecutive steps)
    direction block counter (id_lo, id_hi, step, 1):      rtheta = u53(w0,w1) * 2 * pi, rphi = u53(w2,w3) * pi
    """
    ids = np.asarray(ids, dtype=np.uint64)
    lo, hi = ids & _U32, ids >> np.uint64(32)
    k0, k1 = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    st = np.uint64(int(step) & 0xFFFFFFFF)
    odd = int(step) & 1
    a = philox4x32_10(lo, hi, np.uint64((int(step) & 0xFFFFFFFF) >> 1), np.uint64(0), k0, k1)
    b = philox4x32_10(lo, hi, st, np.uint64(1), k0, k1)
    ra, rb = (a[2], a[3]) if odd else (a[0], a[1])
    if dtype == np.float32:
        # the top 24 bits of the same words: u32 <= u64 < u32 + 2**-24, so both precisions follow one stream
        u24 = lambda w: (w >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        pi32 = np.float32(np.pi)
        return (u24(b[0]) * np.float32(2)) * pi32, u24(b[2]) * pi32, u24(ra)
    rand = u53(ra, rb)
    rtheta = u53(b[0], b[1]) * 2 * np.pi
    rphi = u53(b[2], b[3]) * np.pi
    return rtheta, rphi, rand


def philox_energy(seed, ids, e_min, e_max, power=3.0):
    """Bulk photon energies ``E = min + (max-min) * U**(1/power)`` -- the default sampler of
    ``generate_photons`` (``np.random.power(3)``, light.py:112,126) driven by Philox block 2 of
    step 0xFFFFFFFF so it never collides with a step's scatter stream."""
    ids = np.asarray(ids, dtype=np.uint64)
    lo, hi = ids & _U32, ids >> np.uint64(32)
    k0, k1 = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    w = philox4x32_10(lo, hi, np.uint64(0xFFFFFFFF), np.uint64(2), k0, k1)
    u = u53(w[0], w[1])
    return np.float64(e_min) + (np.float64(e_max) - np.float64(e_min)) * np.power(u, 1.0 / power)


def planck_table(e_min, e_max, T, bins, kB=1.380649e-23):
    """Binned Planck CDF of planck_phot_distribution (light.py:73-96): grid = linspace(min, max, bins); bin x
    carries the integral of planck_distribution (light.py:53-60: 15/(pi^4 kT) * x^3 * exp(-x), x = E/kT) over
    [grid[x], grid[x+1]], normalised.  The reference integrates with scipy.quad; the integrand has the closed
    antiderivative -exp(-x)(x^3 + 3x^2 + 6x + 6), used here (agrees with quad to its own tolerance).
    Returns (cdf[bins-1], grid[bins-1]): the value for bin x is grid[x]."""
    grid = np.linspace(e_min, e_max, int(bins))
    x = grid / (kB * T)
    F = -np.exp(-x) * (x ** 3 + 3 * x ** 2 + 6 * x + 6)
    mass = np.diff(F)
    cdf = np.cumsum(mass / mass.sum())
    cdf[-1] = 1.0
    return cdf, grid[:-1].copy()


def philox_table_energy(seed, ids, cdf, grid):
    """Device table sampler (Philox block 3 of step 0xFFFFFFFF): first x with cdf[x] >= u -> grid[x]."""
    ids = np.asarray(ids, dtype=np.uint64)
    lo, hi = ids & _U32, ids >> np.uint64(32)
    w = philox4x32_10(lo, hi, np.uint64(0xFFFFFFFF), np.uint64(3), int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    u = u53(w[0], w[1])
    return np.asarray(grid)[np.minimum(np.searchsorted(cdf, u, side="left"), len(cdf) - 1)]


# ----------------------------------------------------------------------------------------------
# whole steps on an SoA state dict {r:[3], v:[3], dr:[3], dv:[3], E, id}
# ----------------------------------------------------------------------------------------------
def step_newton(state, dt, dtype=np.float64):
    state["r"], state["dr"] = newton_euler(state["r"], state["v"], dt, dtype)


def step_scatter_isotropic(state, draws, A_kernel, n_kernel, c, *, h=None, use_E=False, n_expr=None, dtype=np.float64):
    rtheta, rphi, rand = draws
    hit, r0, r1, r2 = scatter_sphere_kernel(
        state["dr"][0], state["dr"][1], state["dr"][2], rtheta, rphi, rand, A_kernel, n_kernel, c,
        h=h if use_E else None, E=state["E"] if use_E else None, n_expr=n_expr,
        r=state["r"] if n_expr is not None else None, dtype=dtype)
    state["v"], state["dv"] = scatter_apply(state["v"], hit, (r0, r1, r2), dtype)
    return hit


def step_scatter_delete(state, rand, A_kernel, n_kernel, dtype=np.float64):
    flags = delete_flags(state["dr"][0], state["dr"][1], state["dr"][2], rand, A_kernel, n_kernel, dtype)
    keep = survivors(flags)
    for f in ("r", "v", "dr", "dv"):
        state[f] = [np.asarray(a)[keep] for a in state[f]]
    for f in ("E", "id"):
        if f in state and state[f] is not None:
            state[f] = np.asarray(state[f])[keep]
    return flags, keep
