"""CPU: the C-ABI library builds, loads, and exports every symbol include/physicl_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "physicl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcl_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from physicl_amd import build, _hip
    build.build_lib()
    return ctypes.CDLL(_hip.LIB_PATH)


def test_header_declares_the_expected_surface():
    syms = header_symbols()
    assert len(syms) >= 39
    for must in ("pcl_step_newton", "pcl_step_scatter_isotropic", "pcl_step_scatter_delete",
                 "pcl_k_light_scatter_step_sphere", "pcl_k_light_scatter_step_del", "pcl_k_scatter_delete_test"):
        assert must in syms


def test_library_exports_every_declared_symbol(lib):
    missing = [s for s in header_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    from physicl_amd import _hip
    assert sorted(_hip.EXPORTS) == header_symbols()


def test_abi_version_and_error_string(lib):
    assert lib.pcl_abi_version() == 1
    lib.pcl_last_error.restype = ctypes.c_char_p
    assert lib.pcl_last_error() is not None


def test_expression_validator_runs_without_gpu():
    from physicl_amd import _hip
    ok = ["0.000000001 * exp(r0[gid] - 5)",
          "2.5E+25 * exp(-1 * (sqrt(pow(r0[gid], 2) + pow(r1[gid], 2) + pow(r2[gid], 2)) - 6371000.0)/(8600.0))",
          "2.5e25 * exp(r2[gid] / 8600.0)", "1.0", "fmax(0.0, 1e-3 * (10 - fabs(r1[ gid ])))", ".5e-3*E[gid]"]
    for e in ok:
        _hip.validate_expr(e)
    bad = ["", "r0[gid+1]", "r0[0]", "r0", "system(1)", "x", "1; while(1){}", "exp", "r0[gid]]", "(1", "1e", "2.0f",
           "a.hits[0]", "gid", "r0[gid] ? 1 : 0", "\"s\"", "__builtin_trap()", "1)+(2"]
    for e in bad:
        with pytest.raises(ValueError):
            _hip.validate_expr(e)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from physicl_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _hip.load()
