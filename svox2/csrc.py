"""`svox2.csrc`: the operator module the reference binds with pybind11 (svox2/csrc/svox2.cpp),
served here by ctypes calls into libasurf_hip.so."""
from alphasurf_amd.csrc_api import *  # noqa: F401,F403
from alphasurf_amd.csrc_api import (SparseGridSpec, CameraSpec, RaysSpec, RayVoxIntersecSpec,  # noqa: F401
                                    RenderOptions, GridOutputGrads)
