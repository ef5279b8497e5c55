"""Reference-SHAPED CPU baseline: one Python object per photon, one Python loop per step -- the
structure of the reference's CPU path (physicl/newton.py:14-16, physicl/light.py:335-350), restated
with plain numpy 3-vectors.  TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline_python leg); it exists
so that the per-object cost the reference pays (~1e4..1e5 particle-steps/s, BASELINE.md sections 1-2)
can be quoted from the same box as the GPU number.  Single-threaded by construction, like the
reference (one threading.Thread + GIL, physicl/__init__.py:400)."""
import time

import numpy as np

C, H = 299792458.0, 6.62607015e-34


class _Photon:
    __slots__ = ("r", "v", "dr", "dv", "E")

    def __init__(self, E):
        self.r = np.zeros(3)
        self.v = np.array([C, 0.0, 0.0])
        self.dr = np.zeros(3)
        self.dv = np.zeros(3)
        self.E = E


def newton(objs, dt):                                   # physicl/newton.py:14-16
    for o in objs:
        o.dr = o.v * dt
        o.r += o.dr


def scatter_isotropic(objs, A, n, use_E, profile):     # physicl/light.py:336-350 (+ the variable-n factor)
    hits = 0
    for o in objs:
        d = o.dr
        norm = np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
        p = A * (profile[1] * np.exp(o.r[0] - profile[2])) * norm if profile else A * n * norm
        if use_E:
            p *= ((H * C) / o.E) ** -4
        if p >= np.random.random():
            phi = np.random.random() * np.pi
            theta = np.random.random() * np.pi * 2
            vold = o.v
            o.v = np.array([C * np.sin(theta) * np.cos(phi), C * np.sin(theta) * np.sin(phi), C * np.cos(theta)])
            o.dv = o.v - vold
            hits += 1
        else:
            o.dv = np.zeros(3)
    return hits


def sign_counts(objs):                                  # physicl/light.py:423-426
    xp = yp = zp = 0
    for o in objs:
        xp += int(o.v[0] > 0)
        yp += int(o.v[1] > 0)
        zp += int(o.v[2] > 0)
    return xp, yp, zp


def time_steps(E, dt, A, n, use_E, profile, seconds=3.0):
    """particle-steps/s of [Newton, ScatterIsotropic, sign counters] over the photons with energies E."""
    objs = [_Photon(float(e)) for e in E]
    np.random.seed(0)
    with np.errstate(all="ignore"):
        steps, t0 = 0, time.perf_counter()
        while True:
            newton(objs, dt)
            scatter_isotropic(objs, A, n, use_E, profile)
            sign_counts(objs)
            steps += 1
            el = time.perf_counter() - t0
            if el >= seconds:
                break
    return len(objs) * steps / el, steps, el
