"""The delete and mixed legs of bench.py alone (for tools/ab_lib.sh): python tools/ab_legs.py [delete|mixed|both]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
which = sys.argv[1] if len(sys.argv) > 1 else "both"
sys.argv = sys.argv[:1]
import bench
from physicl_amd import _hip as hip
dev = hip.Device(0)
if which in ("delete", "both"):
    d = bench.delete_leg(dev, hip, [100_000_000], 1234)
    for size, rec in d["sizes"].items():
        print("delete", size, {m: "%.4g" % rec[m]["value"] for m in ("per_step", "multi")},
              "ahead kernel ms", rec["per_step"]["kernels_total_ms"].get("k_delete_ahead"), flush=True)
if which in ("mixed", "both"):
    m = bench.mixed_leg(dev, 100_000_000)
    print("mixed f64 %.4g f32 %.4g  s_f64 %.4f s_f32 %.4f" % (m["value_f64"], m["value_f32"], m["seconds_f64"], m["seconds_f32"]), flush=True)
dev.close()
