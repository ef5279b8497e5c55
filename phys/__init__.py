"""Alias package so that scripts written for PhysiCL (``import physicl as phys`` -- or the older
``import phys``) run unchanged on the MI355X build.  Everything lives in ``physicl_amd``."""
from physicl_amd import *          # noqa: F401,F403
from physicl_amd import __all__, __version__  # noqa: F401
from . import light, newton        # noqa: F401,E402
