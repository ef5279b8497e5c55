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
import builtins
import dis
import functools
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
_MAX_ITEMS, _MAX_DEPTH = 4096, 4

# Code the functions may call.  A module is "plain" only if it is on this list: what ``time``, ``random``, ``datetime``,
# ``os`` ... return is not a function of the run's clock, so K evaluations ahead of a launch and one per pass between the
# passes (physicl/__init__.py:512-516) are different programs (a wall-clock exit would overshoot by up to K - 1 passes,
# a ``np.random`` exit would consume the global stream in another order than the reference loop).
_MODULE_ROOTS = frozenset(("builtins", "math", "cmath", "operator", "numpy", "physicl_amd", "physicl", "phys"))
_MODULE_DENY = ("numpy.random", "numpy.testing", "numpy.ctypeslib", "numpy.distutils", "numpy.f2py")
# names that are never plain, whether they come up as a global, a builtin or an attribute (``np.random``)
_NAME_DENY = frozenset(("random", "open", "input", "eval", "exec", "compile", "__import__", "globals", "locals", "vars", "setattr",
                        "delattr", "getattr", "breakpoint", "load", "fromfile", "loadtxt", "genfromtxt", "datetime64", "memmap"))
# methods that change the object they are called on: a function that counts its own calls in a captured list is not a
# function of the clock
_MUTATORS = frozenset(("append", "extend", "insert", "pop", "remove", "clear", "update", "add", "discard", "setdefault", "popitem",
                       "sort", "reverse", "fill", "put", "itemset", "resize", "partition", "setflags", "setfield", "__setitem__",
                       "__delitem__", "__setattr__", "__delattr__", "__iadd__", "appendleft", "popleft", "rotate"))
_WRITES = frozenset(("STORE_GLOBAL", "STORE_SUBSCR", "STORE_ATTR", "STORE_NAME", "DELETE_SUBSCR", "DELETE_ATTR", "DELETE_GLOBAL",
                     "DELETE_NAME", "DELETE_DEREF"))


def _module_ok(name):
    if not isinstance(name, str):
        return False
    if any(name == d or name.startswith(d + ".") for d in _MODULE_DENY):
        return False
    return name.split(".", 1)[0] in _MODULE_ROOTS


def _code_ok(v):
    """Modules, builtin functions, classes and ufuncs of the allow-listed modules (``math.sqrt``, ``len``, ``np.double``,
    ``Measurement``); ``time.time``, ``random.random`` (a method of a hidden ``Random``), ``np.random.random`` (of the global
    ``RandomState``), ``datetime.datetime`` are not."""
    if isinstance(v, types.ModuleType):
        return _module_ok(v.__name__)
    if isinstance(v, np.ufunc):
        return True
    if isinstance(v, (types.BuiltinFunctionType, types.MethodDescriptorType, types.WrapperDescriptorType)):
        owner = getattr(v, "__self__", None)
        if owner is not None and not (isinstance(owner, types.ModuleType) and _module_ok(owner.__name__)):
            return False
        mod = getattr(v, "__module__", None)
        if mod is None:
            mod = getattr(owner, "__name__", None) or getattr(getattr(v, "__objclass__", None), "__module__", None)
        return _module_ok(mod) and getattr(v, "__name__", "") not in _NAME_DENY
    if isinstance(v, type):
        return _module_ok(getattr(v, "__module__", None))
    return False


def _plain(v, live, depth):
    """True if ``v`` cannot hand the function anything a launch produces, nor anything that differs from one call to the
    next: numbers, strings, arrays, Measurements, allow-listed modules / builtins / classes, containers of those (but
    never a container that IS a step's ``data``), and plain functions that are themselves clock-only."""
    if isinstance(v, _SCALARS) or _code_ok(v):
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


@functools.lru_cache(maxsize=256)
def _instrs(code):
    """(opname, argval, arg) of a code object's instructions, decoded once per code object (both analyses below read them;
    a script's exit lambda is judged again for every Simulation it creates)."""
    return tuple((i.opname, i.argval, i.arg) for i in dis.get_instructions(code))


def _codes(code):
    """The code object and every code object nested in it (lambdas, comprehensions, inner functions)."""
    yield code
    for c in code.co_consts:
        if isinstance(c, types.CodeType):
            yield from _codes(c)


def _names(code):
    """Every name the code object (and the code objects nested in it) may look up as a global or as an attribute."""
    out = set()
    for c in _codes(code):
        out |= set(c.co_names)
    return out


def _writes_state(code):
    """Name of the first instruction that changes something outliving the call (a global, an item or attribute of any
    object, a variable of an enclosing scope), or of a disallowed ``import``; None if there is none.  Locals -- and cell
    variables the code object owns -- are the function's own business."""
    for c in _codes(code):
        for opname, argval, _ in _instrs(c):
            if opname in _WRITES:
                return "%s %s" % (opname, argval if isinstance(argval, str) else "")
            if opname == "STORE_DEREF" and argval in c.co_freevars:
                return "STORE_DEREF %s" % argval
            if opname == "IMPORT_NAME" and not _module_ok(argval):
                return "import %s" % argval
    return None


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
    b = g.get("__builtins__", builtins)
    b = b.__dict__ if isinstance(b, types.ModuleType) else b
    names = _names(fn.__code__)
    for name in sorted(names):
        if name in _NAME_DENY:
            return False, "names %r, whose value is not a function of the run's clock" % name
        if name in _MUTATORS:
            return False, "calls .%s(), which changes an object between two calls" % name
        if name in g:
            if not _plain(g[name], live, depth):
                return False, "uses the global %r (%s)" % (name, type(g[name]).__name__)
        elif name in b and not _plain(b[name], live, depth):
            return False, "uses the builtin %r" % name
    w = _writes_state(fn.__code__)
    if w is not None:
        return False, "keeps state between its calls (%s)" % w.strip()
    for d in (fn.__defaults__ or ()) + tuple((fn.__kwdefaults__ or {}).values()):
        if not _plain(d, live, depth):
            return False, "has a default argument of type %s" % type(d).__name__
    return True, None


def clock_only(fn, steps=()):
    """(ok, why_not): can ``fn(sim)`` reach nothing but plain data except through its argument, and does it keep no
    state of its own?  ``steps``: the simulation's steps -- they, and the ``data`` containers of the measure steps among
    them, are what a launch changes."""
    live = []
    for st in steps:
        live.append(st)
        d = getattr(st, "data", None)
        if d is not None:
            live.append(d)
    return _function_ok(fn, live, 0)


# ---------------------------------------------------------------------------------------------------------------------
# how does the function use the object count?  (loops with a ScatterDeleteStep: the count of a later pass is not known
# ahead of the launch, only that a non-empty store stays non-empty until the rows say otherwise)
# ---------------------------------------------------------------------------------------------------------------------
_EMPTY_TESTS = frozenset((("==", 0), ("<=", 0), ("<", 1), ("!=", 0), (">", 0), (">=", 1)))
_TRUTH_OPS = frozenset(("UNARY_NOT", "POP_JUMP_IF_FALSE", "POP_JUMP_IF_TRUE", "JUMP_IF_FALSE_OR_POP", "JUMP_IF_TRUE_OR_POP",
                        "POP_JUMP_FORWARD_IF_FALSE", "POP_JUMP_FORWARD_IF_TRUE", "POP_JUMP_BACKWARD_IF_FALSE",
                        "POP_JUMP_BACKWARD_IF_TRUE"))
_LOADS = frozenset(("LOAD_FAST", "LOAD_DEREF", "LOAD_GLOBAL", "LOAD_NAME", "LOAD_CLOSURE"))


def _is_builtin(fn, name, what):
    g = fn.__globals__
    if name in g:
        return g[name] is what
    b = g.get("__builtins__", builtins)
    b = b.__dict__ if isinstance(b, types.ModuleType) else b
    return b.get(name) is what


def _count_use_code(fn, code):
    """'none' | 'emptiness' | 'other' for one code object: every ``<x>.objects`` must be consumed by a truth test
    (``not s.objects``, ``if s.objects``, ``bool(s.objects)``) or by ``len(s.objects) <op> <const>`` that only asks
    whether the list is empty (``== 0``, ``< 1``, ``> 0`` ...).  Recognised on the bytecode; anything else -- another
    comparison, arithmetic on the length, passing the list on -- is 'other'."""
    ins = [i for i in _instrs(code) if i[0] not in ("PRECALL", "CACHE", "PUSH_NULL", "RESUME", "EXTENDED_ARG")]
    use = "none"
    for k, (opname, argval, arg) in enumerate(ins):
        if argval != "objects" or opname not in ("LOAD_ATTR", "LOAD_METHOD", "STORE_ATTR", "DELETE_ATTR"):
            continue                             # (a variable that happens to be called ``objects`` is judged as a value elsewhere)
        if opname != "LOAD_ATTR":
            return "other"
        nxt = ins[k + 1] if k + 1 < len(ins) else None
        if nxt is not None and nxt[0] in _TRUTH_OPS:
            use = "emptiness"
            continue
        call = nxt is not None and nxt[0] in ("CALL_FUNCTION", "CALL") and nxt[2] == 1
        if call and k >= 2 and ins[k - 1][0] in _LOADS and ins[k - 2][0] in ("LOAD_GLOBAL", "LOAD_NAME"):
            callee = ins[k - 2][1]
            if _is_builtin(fn, callee, bool):
                use = "emptiness"
                continue
            if _is_builtin(fn, callee, len) and k + 3 < len(ins) and ins[k + 2][0] == "LOAD_CONST" and \
                    type(ins[k + 2][1]) is int and ins[k + 3][0] == "COMPARE_OP" and (ins[k + 3][1], ins[k + 2][1]) in _EMPTY_TESTS:
                use = "emptiness"
                continue
        return "other"
    return use


def count_use(fn, _depth=0):
    """How ``fn(sim)`` -- and every plain function it can reach through its closure or the globals it names -- uses
    ``sim.objects``: 'none', 'emptiness' (only whether the list is empty) or 'other'."""
    if not isinstance(fn, types.FunctionType) or _depth > _MAX_DEPTH:
        return "other"
    rank = {"none": 0, "emptiness": 1, "other": 2}
    use = "none"
    for c in _codes(fn.__code__):
        u = _count_use_code(fn, c)
        use = u if rank[u] > rank[use] else use
    reach = [cell.cell_contents for cell in (fn.__closure__ or ()) if _bound(cell)]
    reach += [fn.__globals__[n] for n in _names(fn.__code__) if n in fn.__globals__]
    reach += list(fn.__defaults__ or ()) + list((fn.__kwdefaults__ or {}).values())
    for v in reach:
        if isinstance(v, types.FunctionType):
            u = count_use(v, _depth + 1)
            use = u if rank[u] > rank[use] else use
    return use


def _bound(cell):
    try:
        cell.cell_contents
        return True
    except ValueError:
        return False
