"""Same import path as the reference's src/CGx/KNPEMI/KNPEMIx_solver.py."""
from cgx_hip.solver import SolverKNPEMI  # noqa: F401
