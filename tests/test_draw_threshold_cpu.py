"""The identity k_delete_ahead_live relies on (physicl_amd/csrc/physicl_hip.hip, ``ahead_draw``): the delete test of
light.py:243, ``pcoll >= rand`` with rand = m * 2^-53 (m the 53 random bits of two Philox words; fp32: 24 bits of one), is
the integer compare ``m <= floor(pcoll * 2^53)`` -- exact for every pcoll, including the values next to k * 2^-53, zero, the
negative zero, denormals, values of one and above (every draw removes), NaN and negative values (none does).  numpy's
float64 / float32 arithmetic is the device's; this runs without a GPU."""
import numpy as np


def threshold(pc, bits, ftype, itype):
    """ahead_draw<T>::threshold, restated."""
    pc = np.asarray(pc, dtype=ftype)
    scale = ftype(2.0) ** bits
    y = pc * scale
    out = np.full(pc.shape, -1, dtype=itype)
    ok = pc >= 0                                     # False for NaN
    big = ok & (y >= scale)
    mid = ok & ~big
    out[big] = itype(2) ** bits
    out[mid] = np.trunc(y[mid]).astype(itype)
    return out


def check(bits, ftype, itype, seed):
    rng = np.random.default_rng(seed)
    top = 2 ** bits
    m_edge = np.array([0, 1, 2, top // 2 - 1, top // 2, top // 2 + 1, top - 2, top - 1], dtype=np.int64)
    m = np.concatenate([m_edge, rng.integers(0, top, 4000)])
    u = m.astype(ftype) * ftype(2.0) ** -bits         # R::uniform: exact
    assert np.all(u.astype(np.float64) * 2.0 ** bits == m)
    eps = np.finfo(ftype).eps
    pcs = [u, np.nextafter(u, ftype(2)), np.nextafter(u, ftype(-1)), u * (1 + eps), rng.random(len(m)).astype(ftype)]
    special = np.array([0.0, -0.0, np.finfo(ftype).tiny, np.nextafter(ftype(0), ftype(1)), 1.0, np.nextafter(ftype(1), ftype(0)), 1.5,
                        np.finfo(ftype).max, np.inf, -np.inf, np.nan, -1e-300 if ftype is np.float64 else -1e-30, -0.25], dtype=ftype)
    with np.errstate(over="ignore", invalid="ignore"):
        for pc in pcs:
            pc = pc.astype(ftype)
            assert np.array_equal(pc >= u, m.astype(itype) <= threshold(pc, bits, ftype, itype))
        for pc in special:                               # every special value against every draw
            want = np.full(len(u), pc, dtype=ftype) >= u
            got = m.astype(itype) <= threshold(np.full(len(u), pc, dtype=ftype), bits, ftype, itype)
            assert np.array_equal(want, got), pc


def test_threshold_compare_is_the_reference_compare_fp64():
    check(53, np.float64, np.int64, 1)


def test_threshold_compare_is_the_reference_compare_fp32():
    check(24, np.float32, np.int32, 2)
