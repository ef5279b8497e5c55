"""debug: does device memory come back after pcl_pool_trim (idle handles released) while the address ranges stay reserved?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from physicl_amd import _hip as hip
BIG = int(float(sys.argv[1])) if len(sys.argv) > 1 else 400_000_000
d = hip.Device(0)
gb = lambda x: round(x / 1e9, 2)
print("start free", gb(d.mem_info()[0]), flush=True)
d.store_alloc(BIG); d.fill_photons(BIG, 0, 299792458.0, 1.0, 2.0, 3); d.sync()
print("store up: free", gb(d.mem_info()[0]), "pool", gb(hip.pool_bytes()), flush=True)
d.store_free()
print("store freed: free", gb(d.mem_info()[0]), "pool", gb(hip.pool_bytes()), flush=True)
print("trim released", gb(hip.pool_trim()), flush=True)
for k in range(8):
    print("  t+%.1f free" % (0.25 * k), gb(d.mem_info()[0]), "pool", gb(hip.pool_bytes()), flush=True)
    time.sleep(0.25)
d.store_alloc(100_000_000)
print("new store: free", gb(d.mem_info()[0]), "pool", gb(hip.pool_bytes()), d.alloc_info(), flush=True)
