"""Drop-in alias: `import svox2` resolves to the MI355X-native implementation in alphasurf_amd,
so the reference's training script (opt/opt.py) runs unchanged against the HIP hot path."""
from alphasurf_amd.grid import *  # noqa: F401,F403
from alphasurf_amd.grid import CONSTANTS as _CONSTANTS
from alphasurf_amd import __version__  # noqa: F401
from . import utils  # noqa: F401

globals().update(_CONSTANTS)
