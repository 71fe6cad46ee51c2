import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "knp-emi-cgx_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Build the HIP library (hipcc cross-compiles gfx950 without a GPU) unless it is already the product of the current
    sources (content stamp); a stale or absent library must never make the product fall back to anything else."""
    import __graft_entry__ as g
    g.build()
