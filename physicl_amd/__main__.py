"""``python -m physicl_amd``: what the library sees on this machine (devices, library build, hipRTC, memory settings)."""
import json
import os
import sys


def main():
    out = {"package": os.path.dirname(os.path.abspath(__file__))}
    try:
        from . import _hip
        lib = _hip.load()
        out["library"] = {"path": getattr(lib, "_name", None), "abi_version": lib.pcl_abi_version()}
        n = _hip.device_count()
        out["devices"] = []
        for i in range(n):
            with _hip.Device(i) as d:
                out["devices"].append(d.info())
        out["expression_check"] = "ok" if _hip.validate_expr("0.000000001 * exp(r0[gid] - 5)") is None else "?"
        out["pool_bytes_idle"] = _hip.pool_bytes()
    except Exception as e:                           # noqa: BLE001 -- this is a diagnostic: say what failed and go on
        out["error"] = "%s: %s" % (type(e).__name__, e)
    out["environment"] = {k: v for k, v in sorted(os.environ.items()) if k.startswith(("PCL_", "HIP_VISIBLE", "ROCR_VISIBLE", "HSA_ENABLE"))}
    json.dump(out, sys.stdout, indent=1, default=str)
    sys.stdout.write("\n")
    return 1 if "error" in out else 0


if __name__ == "__main__":
    sys.exit(main())
