"""physicl.newton equivalent: explicit-Euler kinematics on the device store."""
from .core import DeviceStep


class NewtonianKinematicsStep(DeviceStep):
    """Moves ALL objects: ``dr = v*dt`` (rounded and stored), ``r = r + dr``
    (physicl/newton.py:10-16).  Kernel k_newton, or part of the fused loop-body kernel when the next
    step is a ScatterIsotropicStep / counting measure step (bit-identical either way)."""
    _fuse_role = "newton"

    def __init__(self):
        pass

    def _device_run(self, sim):
        sim._dev.step_newton(sim._dt_code())
