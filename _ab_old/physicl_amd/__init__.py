"""physicl_amd -- MI355X-native implementation of PhysiCL's per-particle time-step hot path, behind
PhysiCL's own Simulation / Step plugin API and NumPy code-units system.

    import physicl_amd as phys          # or: import physicl as phys  (alias package at the repo root)
    import physicl_amd.newton, physicl_amd.light

Importing this package never touches the GPU; creating ``Simulation(cl_on=True)`` (the default) does,
through libphysicl_hip.so (``physicl_amd._hip``).  There is no CPU path for the device steps.
"""
from .units import Measurement, MeasurementError
from .core import (CLInput, CLOutput, CLProgram, DeviceStep, MeasureStep, Object, ObjectList, PhotonBatch,
                   Simulation, Step, UpdateTimeStep)

__version__ = "0.1.0"


def _hip_error():
    """The exception class raised for device / hipRTC failures (lazy: importing the package stays GPU-free)."""
    from ._hip import HipError
    return HipError
__all__ = ["Measurement", "MeasurementError", "Step", "DeviceStep", "UpdateTimeStep", "MeasureStep", "Object",
           "ObjectList", "PhotonBatch", "Simulation", "CLInput", "CLOutput", "CLProgram"]
