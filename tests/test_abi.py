"""The C-ABI library loads and exports every symbol include/knpemi_hip.h declares (no compute calls:
this runs without a GPU), and the product path refuses to run without a HIP device."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "knpemi_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(knp_[a-z0-9_]+)\s*\(", txt)) - {"knp_halo_fn", "knp_allreduce_fn"})


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from cgx_hip import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in knpemi_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in cgx_hip/_lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_null_ctx_is_an_error_not_a_crash():
    from cgx_hip import _lib
    lib = _lib.load()
    assert lib.knp_set_nullspace(None, 1) < 0
    assert lib.knp_destroy(None) == 0
    assert lib.knp_last_error(None) == b"null ctx"


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cgx_hip._lib import KnpError
    from parity_utils import ci_config, make_problem
    p = make_problem(ci_config(N=8, steps=1))
    with pytest.raises(KnpError):
        p.create_backend()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "knp-emi-cgx_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "knpemi_oracle" not in src and "oracle/" not in src.replace("the oracle", ""), os.path.join(dirpath, f)


def test_host_graph_builder_under_address_sanitizer():
    """tools/asan/run.sh: the host-side graph builder (csrc/knp_setup.cpp) compiled with -fsanitize=address,undefined and run on a
    small 2D and a small 3D mesh (GPU sanitizers are not available on the pool: CPU build only).  Skipped without libasan."""
    import shutil
    import subprocess
    import pytest
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    probe = subprocess.run("echo 'int main(){return 0;}' | g++ -x c++ -fsanitize=address,undefined - -o /dev/null", shell=True, capture_output=True)
    if probe.returncode != 0:
        pytest.skip("libasan / libubsan not installed")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan", "run.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert r.stdout.count("rc=0") == 2, r.stdout


def test_host_threads_follow_the_cpu_share_of_the_process():
    """The library's one-off host passes (graph build, hierarchy hand-over) are OpenMP loops sized to what the process may use: a GPU box
    shows every logical CPU of its host (256) inside a 16-core cgroup quota, and 256 threads there made the hand-over 10x slower.
    ``knp_host_thread_count``: KNP_HOST_THREADS overrides; otherwise at most 32, at most the CPUs of the affinity mask / cgroup quota,
    divided among the ranks of a node (LOCAL_WORLD_SIZE)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); from cgx_hip import _lib; print(_lib.load().knp_host_thread_count())"
            % os.path.join(ROOT, "knp-emi-cgx_amd"))

    def count(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("KNP_HOST_THREADS", "LOCAL_WORLD_SIZE")}
        e.update(env)
        return int(subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout.split()[-1])
    base = count()
    assert 1 <= base <= min(32, len(os.sched_getaffinity(0)))
    assert count(KNP_HOST_THREADS="3") == 3
    assert 1 <= count(LOCAL_WORLD_SIZE="2") <= base and count(LOCAL_WORLD_SIZE="4096") == 1

