"""MI355X-native Matrix-Operator RT core (drop-in for vSmartMOM.jl's CoreRT hot path).

The directory name contains a dot, so import it through the root-level shim:
    import rtamd            # -> this package, registered as module "rtamd"
"""
from . import _lib  # noqa: F401
from ._lib import Handle, MomError, voigt_xsec, load  # noqa: F401
from . import corert, scenes, absorption, sharding  # noqa: F401,E402
from .corert import (MI355X, rt_run, rt_run_dual, ScenePartial, rt_run_test_ms, rt_run_operators, model_from_parameters, prepare_scene,  # noqa: F401,E402
                     vSmartMOM_Parameters, vSmartMOM_Model, Stokes_I, Stokes_IQU, Stokes_IQUV)
