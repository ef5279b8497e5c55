"""Where the first delete launch of a fresh 1e8-photon store spends its time (allocation vs kernels)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicl_amd import _hip
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
d = _hip.Device(0)
def T(label, fn):
    d.sync(); t = time.perf_counter(); r = fn(); d.sync(); print("%-44s %.1f ms" % (label, (time.perf_counter() - t) * 1e3), flush=True); return r
T("store_alloc", lambda: d.store_alloc(N))
T("fill_photons (first touch)", lambda: d.fill_photons(N, 0, 299792458.0, 1.0, 1.0, 7))
T("fill_photons again", lambda: d.fill_photons(N, 0, 299792458.0, 1.0, 1.0, 7))
plane = [[1e6, np.nan, np.nan]]
T("fused_delete_multi K=16 (first: second slab)", lambda: d.step_fused_delete_multi(1e-3, 16, 1e-3, 1e-3, 7, 0, plane))
T("fused_delete_multi K=16 (second launch)", lambda: d.step_fused_delete_multi(1e-3, 16, 1e-3, 1e-3, 7, 16, plane))
T("fill_photons again", lambda: d.fill_photons(N, 0, 299792458.0, 1.0, 1.0, 7))
T("fused_delete_multi K=16 (warm)", lambda: d.step_fused_delete_multi(1e-3, 16, 1e-3, 1e-3, 7, 0, plane))
T("store_free", lambda: d.store_free())
T("store_alloc again", lambda: d.store_alloc(N))
d.close()
