"""What ``exit(sim)`` and ``UpdateTimeStep.fn(sim)`` may depend on when the host part of several passes of the loop is
evaluated AHEAD of the one launch that runs them (``Simulation.steps_per_launch``).

The reference's loop (physicl/__init__.py:512-516) calls ``exit`` between any two passes and the time-step function at
the start of each, on the live simulation.  Running K passes per launch is invisible only if both functions would have
returned the same values anyway, i.e. if they look at nothing the K passes produce: the clock (``t``, ``dt``, ``ts`` --
computed ahead, exactly), the object count (unchanged by scatter steps; for delete steps see
``Simulation._plan_passes``) and constants.  Two guards decide that, and any doubt means one launch per light step:

* ``clock_only(fn, steps)`` -- before the first launch: what the function can reach WITHOUT going through its ``sim``
  argument (closure cells, globals it names) must be plain data.  A closure over a measure step, over its ``data``
  list, over a simulation or over an object of an unknown class fails (``exit=lambda s: len(m.data) >= 10`` reads rows
  the launch has not produced yet, and no view of ``sim`` would ever notice).
* ``AheadView`` -- at every evaluation: the stand-in for ``sim`` exposes the clock, the count and the run's constants
  and raises ``NotAhead`` for anything else (``sim.hits``, ``sim.steps[...]``, ``sim.objects[0]``, a write).
"""
import types

import numpy as np

from .units import Measurement


class NotAhead(BaseException):
    """exit(sim) / UpdateTimeStep.fn(sim) looked at something that is not known ahead of a launch (args[0] = its name).
    Not an Exception: a ``try / except Exception`` or ``hasattr`` inside the user's function must not swallow it."""


class _CountOnly:
    """``view.objects``: the count and nothing else."""
    __slots__ = ("_n",)

    def __init__(self, n):
        self._n = n

    def __len__(self):
        return self._n

    def __bool__(self):
        return self._n > 0

    def __iter__(self):
        raise NotAhead("objects[...]")

    def __getitem__(self, i):
        raise NotAhead("objects[...]")

    def __getattr__(self, name):
        raise NotAhead("objects." + name)


class AheadView:
    """The simulation as the two functions may see it ahead of a launch.  The clock and the count are plain attributes
    (``refresh()`` copies them from the simulation: they are read once per planned pass), everything else goes through
    ``__getattr__``."""
    __slots__ = ("_sim", "_count", "t", "dt", "ts", "objects")
    READABLE = frozenset(("bounds", "cl_on", "seed", "rng", "device", "devices", "start_time", "running", "steps_per_launch", "fuse",
                          "state_need_lock"))

    def __init__(self, sim, count):
        set_ = object.__setattr__
        set_(self, "_sim", sim)
        set_(self, "_count", int(count))
        set_(self, "objects", _CountOnly(int(count)))
        self.refresh()

    def refresh(self):
        set_, sim = object.__setattr__, self._sim
        set_(self, "t", sim.t)
        set_(self, "dt", sim.dt)
        set_(self, "ts", sim.ts)

    def __getattr__(self, name):
        if name in AheadView.READABLE:
            return getattr(self._sim, name)
        raise NotAhead(name)

    def __setattr__(self, name, value):
        raise NotAhead(name + " = ...")


_SCALARS = (type(None), bool, int, float, complex, str, bytes, np.generic, np.ndarray, Measurement)
_CODE_OK = (types.ModuleType, types.BuiltinFunctionType, type, np.ufunc)
_MAX_ITEMS, _MAX_DEPTH = 4096, 4


def _plain(v, live, depth):
    """True if ``v`` cannot hand the function anything a launch produces: numbers, strings, arrays, Measurements,
    modules / builtins / classes, containers of those (but never a container that IS a step's ``data``), and plain
    functions that are themselves clock-only."""
    if isinstance(v, _SCALARS) or isinstance(v, _CODE_OK):
        return True
    if depth >= _MAX_DEPTH or any(v is x for x in live):
        return False
    if isinstance(v, (list, tuple, set, frozenset)):
        return len(v) <= _MAX_ITEMS and all(_plain(x, live, depth + 1) for x in v)
    if isinstance(v, dict):
        return len(v) <= _MAX_ITEMS and all(_plain(k, live, depth + 1) and _plain(x, live, depth + 1) for k, x in v.items())
    if isinstance(v, types.FunctionType):
        return _function_ok(v, live, depth + 1)[0]
    return False


def _names(code):
    """Every name the code object (and the code objects nested in it) may look up as a global."""
    out = set(code.co_names)
    for c in code.co_consts:
        if isinstance(c, types.CodeType):
            out |= _names(c)
    return out


def _function_ok(fn, live, depth):
    if not isinstance(fn, types.FunctionType):
        return False, "%s is not a plain function" % type(fn).__name__
    for name, cell in zip(fn.__code__.co_freevars, fn.__closure__ or ()):
        try:
            v = cell.cell_contents
        except ValueError:                       # not bound yet: cannot be judged
            return False, "closes over the unbound name %r" % name
        if not _plain(v, live, depth):
            return False, "closes over %r (%s)" % (name, type(v).__name__)
    g = fn.__globals__
    for name in sorted(_names(fn.__code__)):
        if name in g and not _plain(g[name], live, depth):
            return False, "uses the global %r (%s)" % (name, type(g[name]).__name__)
    for d in (fn.__defaults__ or ()) + tuple((fn.__kwdefaults__ or {}).values()):
        if not _plain(d, live, depth):
            return False, "has a default argument of type %s" % type(d).__name__
    return True, None


def clock_only(fn, steps=()):
    """(ok, why_not): can ``fn(sim)`` reach nothing but plain data except through its argument?  ``steps``: the
    simulation's steps -- they, and the ``data`` containers of the measure steps among them, are what a launch
    changes."""
    live = []
    for st in steps:
        live.append(st)
        d = getattr(st, "data", None)
        if d is not None:
            live.append(d)
    return _function_ok(fn, live, 0)
