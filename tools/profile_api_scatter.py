"""Where the host time of BASELINE configs[2]'s run through the plugin API goes (bench.py's api.default: 1e8 photons, 500 passes):
cProfile of Simulation.run, after one unprofiled run."""
import cProfile, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import physicl_amd as phys
import physicl_amd.light as light
import physicl_amd.newton as newton

N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
steps = 500


def make():
    dt = np.double(0.005)
    sim = phys.Simulation(exit=lambda s: len(s.ts) >= steps, seed=1234)
    sim.add_objs(light.generate_photons_bulk(N, min=light.E_from_wavelength(700e-9), max=light.E_from_wavelength(200e-9), seed=1234))
    sim.add_step(0, phys.UpdateTimeStep(lambda s: dt))
    sim.add_step(1, newton.NewtonianKinematicsStep())
    sim.add_step(2, light.ScatterIsotropicStep(n=0.000000000000001, A=0.0000000000000000001, wavelength_dep_scattering=True,
                                               variable_n=True, variable_n_fn="0.000000001 * exp(r0[gid] - 5)"))
    sim.add_step(3, light.ScatterSignMeasureStep(None, True))
    sim._to_device()
    sim._dev.sync()
    return sim


sim = make(); sim.start(); sim.join(); print("threaded run_time %.1f ms" % (sim.run_time * 1e3), dict(sim.schedule)); sim.close(download=False)
pr = cProfile.Profile()
sim = make()
sim._dev.prof_enable(True)
t0 = time.perf_counter(); pr.enable(); sim.run(); pr.disable()
print("run() %.1f ms profiled; kernel time of its launches %.1f ms" % ((time.perf_counter() - t0) * 1e3, sim._dev.prof_read(6)["total_ms"]))
rows = sorted(pr.getstats(), key=lambda e: -e.totaltime)
print("microseconds: cumulative | own | calls | function")
for e in rows[:28]:
    c = e.code
    name = c if isinstance(c, str) else "%s:%d(%s)" % (os.path.basename(c.co_filename), c.co_firstlineno, c.co_name)
    print("%10.1f %10.1f %7d  %s" % (e.totaltime * 1e6, e.inlinetime * 1e6, e.callcount, name))
