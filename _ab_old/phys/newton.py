from physicl_amd.newton import *   # noqa: F401,F403
