"""
vgpa_amd: MI355X-native (gfx950) implementation of VGPA's forward-backward variational smoothing sweep,
behind the reference's own Python plugin surface.  All arithmetic of the hot path runs in hand-written HIP
kernels (libvgpa_hip.so, C ABI in include/vgpa_hip.h); there is no CPU fallback.
"""
from ._lib import Context, device_count, load, LIB_PATH                    # noqa: F401
from .numerics import (OdeSolver, Euler, Heun, RungeKutta2, RungeKutta4,   # noqa: F401
                       num_integration, FwdOde, BwdOde)
from .dynamics import (StochasticProcess, OrnsteinUhlenbeck, DoubleWell,   # noqa: F401
                       Lorenz63, Lorenz96, dynamical_systems)
from .likelihood import Likelihood, GaussianLikelihood, PriorKL0           # noqa: F401
from .variational import VarGP                                             # noqa: F401
from .scg import SCG, DeviceSCG                                            # noqa: F401
from .h5io import save_h5, load_h5, result_dict, save_results, load_results   # noqa: F401

__all__ = ["Context", "device_count", "load", "OdeSolver", "Euler", "Heun", "RungeKutta2", "RungeKutta4",
           "num_integration", "FwdOde", "BwdOde", "StochasticProcess", "OrnsteinUhlenbeck", "DoubleWell",
           "Lorenz63", "Lorenz96", "dynamical_systems", "Likelihood", "GaussianLikelihood", "PriorKL0",
           "VarGP", "SCG", "DeviceSCG", "result_dict", "save_results", "load_results", "save_h5", "load_h5"]
