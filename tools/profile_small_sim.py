"""Where a small simulation's wall time goes (host side): test/test_light.py's scatter run, 1e4 photons x 100 passes,
under cProfile.  Usage: python tools/profile_small_sim.py [n] [steps_per_launch]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import physicl as phys
import physicl.light
import physicl.newton

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def build():
    sim = phys.Simulation(bounds=np.array([1000, 1000, 1000]), cl_on=True, rng="philox", seed=1, steps_per_launch=K,
                          exit=lambda cond: cond.t >= 0.100)
    for i in range(n):
        sim.add_obj(phys.light.PhotonObject(uid=i, v=np.array([phys.light.c, 0, 0], dtype=np.double), E=np.double(1)))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: np.double(0.001)))
    sim.add_step(1, phys.newton.NewtonianKinematicsStep())
    sim.add_step(2, phys.light.ScatterIsotropicStep(A=np.double(0.001), n=np.double(0.001)))
    sim.add_step(3, phys.light.ScatterSignMeasureStep(None, True))
    return sim


for rep in range(2):                       # rep 0 warms the device and the caches
    sim = build()
    t0 = time.perf_counter()
    if rep:
        pr = cProfile.Profile()
        pr.runcall(sim.run)                # the thread's body, in this thread, so that cProfile sees it
    else:
        sim.run()
    el = time.perf_counter() - t0
    print("rep %d: %d passes in %.4f s = %.1f us per pass" % (rep, len(sim.ts), el, el / len(sim.ts) * 1e6))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
