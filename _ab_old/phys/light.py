from physicl_amd.light import *    # noqa: F401,F403
from physicl_amd.light import c, h, kB  # noqa: F401
