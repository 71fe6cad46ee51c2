"""Subset of the reference's src/CGx/utils/misc.py that the hot path needs."""
from cgx_hip.mesh import mark_subdomains_box as mark_subdomains  # noqa: F401
from cgx_hip.problem import flatten_list, range_constructor  # noqa: F401
