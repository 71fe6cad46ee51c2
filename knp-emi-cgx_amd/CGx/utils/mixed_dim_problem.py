"""Same import path as the reference's src/CGx/utils/mixed_dim_problem.py."""
from cgx_hip.problem import MixedDimensionalProblem  # noqa: F401
