"""cgx_hip: host-side mirror of the reference's KNP-EMI interface over libknpemi_hip (HIP, gfx950)."""
from .fem import Constant, Function  # noqa: F401
from .ionic_models import (ATPPump, GlialCotransporters, HodgkinHuxley, IonicModel, KirNaKPumpModel,  # noqa: F401
                           NeuronalCotransporters, PassiveModel)
from .problem import MixedDimensionalProblem, ProblemKNPEMI  # noqa: F401
from .solver import SolverKNPEMI  # noqa: F401

__all__ = ["ProblemKNPEMI", "SolverKNPEMI", "MixedDimensionalProblem", "IonicModel", "PassiveModel", "HodgkinHuxley",
           "ATPPump", "NeuronalCotransporters", "GlialCotransporters", "KirNaKPumpModel", "Constant", "Function"]
