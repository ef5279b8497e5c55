"""physicl_amd -- MI355X-native implementation of PhysiCL's per-particle time-step hot path.

Importing this package never touches the GPU.  ``physicl_amd._hip`` binds libphysicl_hip.so.
"""
__version__ = "0.1.0"
