"""pyamg_amd: the AMG solve phase (multilevel_solver.solve and the amg_core
relaxation kernels of PyAMG) on AMD MI355X through hand-written HIP.

    from pyamg_amd import multilevel_solver, change_smoothers, amg_core, relaxation
"""
from . import _lib
from . import amg_core, relaxation, smoothing, util
from .multilevel import coarse_grid_solver, multilevel_solver
from .smoothing import change_smoothers

__all__ = ["multilevel_solver", "coarse_grid_solver", "change_smoothers", "amg_core", "relaxation",
           "smoothing", "util", "device_count"]


def device_count():
    return _lib.device_count()
