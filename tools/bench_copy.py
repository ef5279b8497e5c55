"""Upload/download rates of one field of the tiled store (dense host array <-> rows of 2048 in the slab)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from physicl_amd import _hip
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
d = _hip.Device(0)
d.store_alloc(N)
d.set_count(N, 0)
a = np.random.RandomState(0).uniform(size=N)
for rep in range(2):
    t0 = time.perf_counter(); d.upload(_hip.R0, a); t1 = time.perf_counter()
    b = d.download(_hip.R0, N); t2 = time.perf_counter()
    print("upload %.3f s (%.1f GB/s)  download %.3f s (%.1f GB/s)  equal=%s" % (t1 - t0, N * 8 / (t1 - t0) / 1e9, t2 - t1, N * 8 / (t2 - t1) / 1e9, np.array_equal(a, b)))
d.close()
