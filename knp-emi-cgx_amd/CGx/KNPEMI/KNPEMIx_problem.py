"""Same import path as the reference's src/CGx/KNPEMI/KNPEMIx_problem.py."""
from cgx_hip.problem import ProblemKNPEMI  # noqa: F401
