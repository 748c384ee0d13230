from alphasurf_amd.shutils import *  # noqa: F401,F403
from alphasurf_amd.shutils import MAX_SH_BASIS, _get_c_extension  # noqa: F401
