"""Alias package so that scripts written for PhysiCL (``import physicl as phys`` -- or the older
``import phys``) run unchanged on the MI355X build.  Everything lives in ``physicl_amd``.

Like the reference package, importing this does NOT import the ``light`` / ``newton`` submodules:
scripts call ``Measurement.set_code_scale`` first and import ``physicl.light`` afterwards so that
``c`` and ``h`` are created in the chosen code units (examples/code_unit_scale_test.ipynb:55)."""
from physicl_amd import *          # noqa: F401,F403
from physicl_amd import __all__, __version__, _hip_error  # noqa: F401
