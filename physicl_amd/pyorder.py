"""Host side of the reference's CPU ("cl_on=False") light steps: the order in which they consume ``np.random``.

The OpenCL paths draw a fixed number of randoms per photon (physicl/__init__.py:606-619), so the host can draw them
all ahead of the kernel.  The CPU paths do not:

* ``ScatterIsotropicStep.__run_py`` (physicl/light.py:335-350) draws ``rand`` for every photon and, ONLY if the photon
  is hit, two more numbers -- first phi (``* np.pi``), then theta (``* np.pi * 2``) -- before it goes on to the next
  photon;
* ``ScatterDeleteStepReference.__run_py`` (physicl/light.py:216-223) removes photons from the very list it iterates
  over, so the object behind every removed photon is skipped: it is neither tested nor does it draw a number.

Where a photon's numbers sit in the stream therefore depends on what happened to the photons before it.  The device
computes every photon's collision probability (``pcl_step_scatter_pcoll``), these functions walk the photons in
order against the global ``np.random`` stream exactly as the reference's loops do, and the device then applies the
outcome (``pcl_step_scatter_isotropic`` with the drawn numbers as inputs, ``pcl_step_delete_flags``).  Afterwards the
global stream stands exactly where the reference would have left it, so a seeded script keeps reproducing the
reference run -- including whatever it draws later.  Nothing here computes physics: comparisons and bookkeeping only.
"""
import numpy as np


def _stream():
    """``draw(k)``: the next k numbers of the global stream, counted from where it stands NOW however often it is
    called.  The walkers below first draw an upper bound, find out how many numbers the reference's loop would have
    consumed, and call ``draw(consumed)`` last -- which leaves the global stream exactly there."""
    state = np.random.get_state()

    def draw(k):
        np.random.set_state(state)
        return np.random.random(k)          # == k consecutive np.random.random() calls (legacy RandomState)
    return draw


def draw_isotropic_py(pcoll, photon=None):
    """RNG consumption of ScatterIsotropicStep.__run_py.  pcoll: (n,) collision probabilities in object order;
    photon: (n,) bool mask of the PhotonObjects (None = all).  Returns (rtheta, rphi, rand, hits): full-length float64
    arrays (entries of skipped objects and of misses stay 0) and the number of hits."""
    pcoll = np.asarray(pcoll, dtype=np.float64)
    n = pcoll.size
    idx = np.arange(n) if photon is None else np.flatnonzero(photon)
    draw = _stream()
    U = draw(3 * idx.size).tolist()
    p = pcoll[idx].tolist()
    rand, rphi, rtheta = [0.0] * idx.size, [0.0] * idx.size, [0.0] * idx.size
    pos = hits = 0
    pi = np.pi
    for k in range(idx.size):
        u = U[pos]
        pos += 1
        rand[k] = u
        if p[k] >= u:                        # light.py:343 (NaN compares false, +inf true)
            rphi[k] = U[pos] * pi            # light.py:344
            rtheta[k] = U[pos + 1] * pi * 2  # light.py:345
            pos += 2
            hits += 1
    draw(pos)                                # the stream now stands where the reference's loop leaves it
    out = [np.zeros(n), np.zeros(n), np.zeros(n)]
    for a, vals in zip(out, (rtheta, rphi, rand)):
        a[idx] = vals
    return out[0], out[1], out[2], hits


def flags_delete_reference_py(pcoll, photon=None):
    """Outcome of ScatterDeleteStepReference.__run_py: int32 flags (1 = removed) in object order.  The list iterator
    advances by one per visit while a removal shifts everything behind it one place forward, so the object that
    follows a removed photon is never visited (whatever its type) and draws nothing."""
    pcoll = np.asarray(pcoll, dtype=np.float64)
    n = pcoll.size
    is_ph = [True] * n if photon is None else np.asarray(photon, dtype=bool).tolist()
    draw = _stream()
    U = draw(n).tolist()
    p = pcoll.tolist()
    flags = np.zeros(n, dtype=np.int32)
    pos, i = 0, 0
    while i < n:
        if is_ph[i]:                          # light.py:218-219
            u = U[pos]
            pos += 1
            if p[i] >= u:                     # light.py:222
                flags[i] = 1
                i += 1                        # the next object slides into the slot just visited: skipped
        i += 1
    draw(pos)
    return flags
