"""Code-units system: ``Measurement``, a float64 ndarray that stores values in *code units*.

Drop-in for the class of the same name in the reference (physicl/__init__.py:18-291): the same unit table and the
same wrapping rules, restated compactly and kept bug-compatible where scripts can see the difference (each such place
cites the reference line it follows).  tests/test_units_parity.py replays 113 expressions under two code scales
against results recorded from the reference itself (tests/golden/g6_unit_ops.json).
The kernels only ever see the code-unit numbers (``float(m)``, ``str(m)``), so this module is host
plumbing: it is NOT accelerated and has no device counterpart.

Model
-----
A ``Measurement`` is the numeric array (already multiplied by ``scale``) plus three attributes:

``scale``           factor from the units it was written in to code units
``units``           {code dimension: power}, e.g. ``{"L": 1, "T": -1}``
``original_units``  {unit name: power} as written, e.g. ``{"m": 1, "s": -1}``

Arrays derived by slicing / copying carry no attributes at all (the reference's "phantom"
measurements); any ufunc re-wraps such operands, and plain numbers, *in the units of the other
operand* (physicl/__init__.py:215-216).  That rule, the first-operand-wins rule for + and -, and
the way named units are merged for * and / (physicl/__init__.py:243-250) are reproduced as they
are, including their surprises (``v * 2`` squares the dimensions; a unit name of the second factor
replaces, not adds to, the first factor's power) because scripts written against the reference see
them through ``units``, ``unitstr()`` and ``repr()``.
"""
import copy
import re

import numpy as np

__all__ = ["Measurement", "MeasurementError"]


class MeasurementError(ArithmeticError):
    """Kept for API compatibility (physicl/__init__.py:11); the reference never raises it."""


# unit name -> [factor, (component unit, power), ...]; components resolve recursively to SI base units.
# Values: BIPM SI brochure, 9th ed. (same definitions the reference cites at physicl/__init__.py:21-22).
_DERIVED = {
    "s": [1, ("s", 1)], "m": [1, ("m", 1)], "kg": [1, ("kg", 1)], "A": [1, ("A", 1)], "K": [1, ("K", 1)],
    "mol": [1, ("mol", 1)], "cd": [1, ("cd", 1)],
    "N": [1, ("kg", 1), ("m", 1), ("s", -2)],
    "Pa": [1, ("kg", 1), ("m", -1), ("s", -2)],
    "J": [1, ("N", 1), ("m", 1)],
    "W": [1, ("kg", 1), ("m", 2), ("s", -3)],
    "C": [1, ("A", 1), ("s", 1)],
    "V": [1, ("W", 1), ("A", -1)],
    "F": [1, ("C", 1), ("V", -1)],
    "Ohm": [1, ("V", 1), ("A", 1)],          # sic: the reference multiplies by A (physicl/__init__.py:65)
    "Wb": [1, ("V", 1), ("s", 1)],
    "T": [1, ("Wb", 1), ("m", -2)],
    "H": [1, ("Wb", 1), ("A", -1)],
    "lm": [1, ("cd", 1)],
    "Bq": [1, ("s", -1)],
    "Gy": [1, ("m", 2), ("s", -2)],
    "Sv": [1, ("m", 2), ("s", -2)],
    "kat": [1, ("mol", 1), ("s", -1)],
    "min": [60, ("s", 1)], "h": [3600, ("s", 1)], "d": [86400, ("s", 1)],
    "au": [149597870700, ("m", 1)],
    "ha": [10 ** 4, ("m", 2)],
    "L": [10 ** -3, ("m", 3)],
    "t": [10 ** 3, ("kg", 1)],
    "Da": [1.6605390666050e-27, ("kg", 1)],
    "eV": [1.602176634e-19, ("J", 1)],
}

_TOKEN = re.compile(r"([a-zA-Z]*)\s*(?:\*\*|\^)\s*(-?\d*)")   # "unit**power" / "unit ^ power"; a bare "m" is ignored

_SCALING = ("multiply", "divide", "true_divide", "floor_divide")
_INVERSE = ("divide", "true_divide", "floor_divide")


def _to_base(unit, power):
    """(factor, [(SI base unit, power), ...]) of ``unit**power``; components are listed, not merged."""
    entry = Measurement.unit_scale[unit]
    factor = entry[0] ** power
    parts = []
    for sub, p in entry[1:]:
        if sub in Measurement.code_scale:
            parts.append((sub, p * power))
        else:
            parts.extend(_to_base(sub, p * power)[1])   # factors of nested units are NOT folded in (reference)
    return factor, parts


class Measurement(np.ndarray):
    # base unit -> [code scale, (code dimension, 1)]; class-global, mutated by set_code_scale
    code_scale = {"s": [1, ("T", 1)], "m": [1, ("L", 1)], "kg": [1, ("M", 1)], "A": [1, ("I", 1)],
                  "K": [1, ("Th", 1)], "mol": [1, ("N", 1)], "cd": [1, ("J", 1)]}
    unit_scale = _DERIVED

    # ------------------------------------------------------------------ class-level configuration
    def set_code_scale(base_unit, new_scale):            # called on the class: Measurement.set_code_scale("m", 1e-3)
        Measurement.code_scale[base_unit][0] = new_scale

    def reset_code_scale(base_unit):
        Measurement.set_code_scale(base_unit, 1)

    # ------------------------------------------------------------------ construction
    def __new__(cls, raw_value, units):
        if isinstance(raw_value, list):
            raw_value = [x.__unscaled__() if isinstance(x, Measurement) else x for x in raw_value]
        obj = np.asarray(raw_value, dtype=np.double).view(cls)
        obj.__scale__(units)
        return obj

    def __scale__(self, units):
        scale = np.double(1)
        dims, named = {}, {}
        for name, power in _TOKEN.findall(units):
            power = int(power)
            factor, parts = _to_base(name, power)
            for base, p in parts:
                code = Measurement.code_scale[base]
                factor *= code[0] ** p
                dims[code[1][0]] = dims.get(code[1][0], 0) + code[1][1] * p
            scale *= factor
            named[name] = named.get(name, 0) + power
        self.scale, self.units, self.original_units = scale, dims, named
        flat = self.flat
        for i in range(self.size):
            flat[i] *= scale

    def __array_finalize__(self, obj):
        pass                                              # derived arrays carry no unit attributes

    def _adopt(self, scale, dims, named):
        self.scale, self.units, self.original_units = scale, dims, named
        return self

    # ------------------------------------------------------------------ views of the value
    def __unscaled__(self):
        out = np.copy(self).view(np.ndarray)
        try:
            flat = out.flat
            for i in range(out.size):
                flat[i] /= self.scale
        except Exception:                                 # phantom measurement: no scale
            print("Error: " + str(out))
        return out

    def value(self):
        return self.__unscaled__()

    def unitstr(self):
        try:
            return " ".join("%s**%s" % (k, v) for k, v in self.original_units.items())
        except AttributeError:
            return ""

    def fstr(self):
        return str(float(self))

    def valstr(self):
        return str(self.value())

    def __str__(self):
        return str(self.view(np.ndarray)).upper()         # pasted into kernel source by the reference (light.py:301)

    def __format__(self, spec):
        return super().__format__(spec).upper()

    def __repr__(self):
        return str(self.value()) + " " + self.unitstr()

    def __deepcopy__(self, memo):
        out = np.copy(self).view(Measurement)
        return out._adopt(self.scale, copy.deepcopy(self.units, memo), copy.deepcopy(self.original_units, memo))

    def rescale(self):
        pass

    # ------------------------------------------------------------------ arithmetic
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        ref = inputs[0] if isinstance(inputs[0], Measurement) else inputs[1]
        ops = [x if (isinstance(x, Measurement) and hasattr(x, "units")) else Measurement(x, ref.unitstr())
               for x in inputs]
        raw = [x.view(np.ndarray) for x in ops]
        if "out" in kwargs:
            kwargs["out"] = tuple(o.view(np.ndarray) for o in kwargs["out"])
        result = getattr(ufunc, method)(*raw, **kwargs)
        name = ufunc.__name__
        first = ops[0]
        if name in _SCALING:
            sign = -1 if name in _INVERSE else 1
            second = ops[1]
            dims = dict(first.units)
            for k, p in second.units.items():
                dims[k] = dims.get(k, 0) + p * sign if k in dims else p * sign
            named = dict(first.original_units)
            for k, p in second.original_units.items():
                if k not in dims:                         # sic: looked up among the code dimensions
                    named[k] = p * sign
                else:
                    named[k] += p * sign
            res = Measurement(np.asarray(result), "")._adopt(first.scale * second.scale ** sign, dims, named)
        elif name in ("power", "square", "sqrt"):
            power = raw[1] if name == "power" else (2 if name == "square" else 1 / 2)
            res = np.asarray(result).view(Measurement)
            dims, named = copy.deepcopy(first.units), copy.deepcopy(first.original_units)
            for k in self.units:                          # the ufunc's receiver, as in the reference
                dims[k] *= power
            for k in self.original_units:
                named[k] *= power
            res._adopt(first.scale ** power, dims, named)
        else:                                             # + - comparisons, reductions, everything else
            res = np.asarray(result).view(Measurement)
            res._adopt(first.scale, copy.deepcopy(first.units), copy.deepcopy(first.original_units))
        return res

    # internal: wrap numbers that are ALREADY in code units (values coming back from the device)
    @classmethod
    def _from_code(cls, values, like=None, units=""):
        out = np.array(values, dtype=np.double).view(cls)
        if like is not None and hasattr(like, "units"):
            return out._adopt(like.scale, copy.deepcopy(like.units), copy.deepcopy(like.original_units))
        probe = cls(np.double(0), units)
        return out._adopt(probe.scale, probe.units, probe.original_units)
