"""CPU: physicl_amd.units.Measurement against results recorded from the reference's Measurement
(tests/golden/g6_unit_ops.json, g6_units.npz; generator tests/golden/make_golden.py).  The comparisons the reference's
own unit tests make (test/test_units.py:25-78) are rows of that recorded table too: the reference evaluated them and the
build has to give the same answers (including the one comparison the reference itself gets wrong)."""
import json
import os

import numpy as np
import pytest

import physicl_amd as phys
import physicl_amd.light as light

M = phys.Measurement
HERE = os.path.dirname(os.path.abspath(__file__))
OPS = json.load(open(os.path.join(HERE, "golden", "g6_unit_ops.json")))


def describe(x):
    d = {"type": type(x).__name__}
    if isinstance(x, M):
        d["code"] = np.asarray(x.view(np.ndarray)).astype(np.float64).tolist()
        d["has_units"] = hasattr(x, "units")
        if hasattr(x, "units"):
            d["scale"] = float(np.asarray(x.scale))
            d["units"] = {k: float(np.asarray(v)) for k, v in x.units.items()}
            d["original_units"] = {k: float(np.asarray(v)) for k, v in x.original_units.items()}
            d["unitstr"] = x.unitstr()
    elif isinstance(x, np.ndarray):
        d["code"] = x.astype(np.float64).tolist()
    elif isinstance(x, (float, int, np.floating, np.integer, bool, np.bool_)):
        d["code"] = float(x)
    else:
        d["text"] = str(x)
    return d


def same(a, b):
    if isinstance(a, dict) and isinstance(b, dict):
        return a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
    if isinstance(a, list) and isinstance(b, list):
        return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
    if isinstance(a, float) and isinstance(b, float):
        return a == b or (np.isnan(a) and np.isnan(b))
    return a == b


@pytest.mark.parametrize("scale_key", ["default", "m_scale_1e-3"])
def test_every_recorded_operation_matches_the_reference(scale_key):
    if scale_key != "default":
        M.set_code_scale("m", 0.001)
    try:
        bad = []
        for rec in OPS[scale_key]:
            try:
                got = describe(eval(rec["expr"], {"M": M, "light": light, "np": np, "__import__": __import__}))
            except Exception as e:
                got = {"raises": type(e).__name__}
            if not same(got, rec["result"]):
                bad.append((rec["expr"], got, rec["result"]))
        assert not bad, "\n".join("%s\n   got  %s\n   want %s" % b for b in bad[:5])
    finally:
        M.reset_code_scale("m")


def test_recorded_constants_and_literals(golden):
    z = golden("g6_units")
    assert str(light.c) == str(z["str_c"]) and str(light.h).upper() == str(z["str_h_upper"])
    assert "{}".format(M(2.5e25, "m**-3")) == str(z["fmt_n0"])
    assert repr(M([1.5, -2.0, 3.25], "m**1 s**-1")) == str(z["repr_vec"])
    for name, m in (("c", light.c), ("h", light.h), ("kB", light.kB)):
        assert np.array_equal(np.asarray(m.view(np.ndarray)), z[name + "_code"]) and float(m.scale) == float(z[name + "_scale"])
    M.set_code_scale("m", 0.001)
    try:
        c2, h2 = M(np.double(299792458), "m**1 s**-1"), M(np.double(6.62607015e-34), "J**1 s**1")
        assert str(c2) == str(z["str_c_mscale"]) and str(h2).upper() == str(z["str_h_mscale_upper"])
        assert float(h2.scale) == float(z["h_mscale_scale"])
        E = (h2 * c2) / M(200e-9, "m**1")
        assert np.array_equal(np.asarray(E.view(np.ndarray)), z["E200_mscale_code"])
        nA = M(2.0e25, "m**-3") * M(5.1e-31, "m**2")
        assert np.array_equal(np.asarray((1 / nA).view(np.ndarray)), z["inv_nA_mscale_code"])
    finally:
        M.reset_code_scale("m")
    assert float(M(1, "m**1").scale) == 1.0            # the class-global scale is restored


def test_photon_constraints():
    with pytest.raises(Exception, match="valid speed"):
        light.PhotonObject(E=1.0, v=np.array([1.0, 0, 0]))
    with pytest.raises(Exception, match="valid energy"):
        light.PhotonObject(v=np.array([light.c, 0, 0], dtype=np.double))
    ps = light.generate_photons(5, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9))
    assert len(ps) == 5 and all(type(p) is light.PhotonObject for p in ps)
    qs = light.generate_photons_from_E([M(1e-19, "J**1"), M(2e-19, "J**1")])
    assert [float(q.E) for q in qs] == [1e-19, 2e-19] and np.array_equal(np.asarray(qs[0].v), [float(light.c), 0, 0])
