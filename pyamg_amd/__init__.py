"""pyamg_amd: the AMG solve phase (multilevel_solver.solve and the amg_core
relaxation kernels of PyAMG) on AMD MI355X through hand-written HIP.

    from pyamg_amd import ruge_stuben_solver, smoothed_aggregation_solver, solve, multilevel_solver
"""
from . import _lib
from . import amg_core, relaxation, smoothing, util
from .multilevel import coarse_grid_solver, multilevel_solver
from .smoothing import change_smoothers
from .classical import ruge_stuben_solver
from .aggregation import smoothed_aggregation_solver
from .blackbox import solve, solver, solver_configuration

# the package-level names of the reference (pyamg/__init__.py:61-65) that live on this path
__all__ = ["multilevel_solver", "coarse_grid_solver", "change_smoothers", "ruge_stuben_solver",
           "smoothed_aggregation_solver", "solve", "solver", "solver_configuration", "amg_core", "relaxation",
           "smoothing", "util", "device_count"]


def device_count():
    return _lib.device_count()
