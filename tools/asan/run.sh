#!/bin/bash
# AddressSanitizer / UBSan run of the host-side graph builder (csrc/knp_setup.cpp) on small 2D and 3D meshes.
# CPU build only (GPU sanitizers are not available on the pool).  usage: bash tools/asan/run.sh
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
W=$(mktemp -d)
cd "$W"
python - <<PY
import sys; sys.path[:0]=['$ROOT/knp-emi-cgx_amd']
import numpy as np
from cgx_hip import mesh as M
for name,(gen,N) in {"sq":(M.create_unit_square,12),"cu":(M.create_unit_cube,6)}.items():
    coords,cells=gen(N); tags=M.mark_subdomains_box(coords,cells)
    gamma,gt,fv=M.gamma_integration_entities(cells,tags,(1,),(2,))
    side=np.where(tags==1,0,1).astype(np.uint8)
    qp,qw=M.facet_quadrature(coords.shape[1],10)
    with open(f"{name}.bin","wb") as f:
        np.array([coords.shape[1],coords.shape[0],cells.shape[0],gamma.shape[0],qw.shape[0]],dtype=np.int32).tofile(f)
        cells.astype(np.int32).tofile(f); coords.astype(np.float64).tofile(f); side.tofile(f); gamma.astype(np.int32).tofile(f)
        np.zeros(gamma.shape[0],dtype=np.int32).tofile(f); qp.astype(np.float64).tofile(f); qw.astype(np.float64).tofile(f)
PY
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fopenmp -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
    -I"$ROOT/include" -I"$ROOT/knp-emi-cgx_amd/csrc" "$ROOT/tools/asan/graph_driver.cpp" "$ROOT/knp-emi-cgx_amd/csrc/knp_setup.cpp" -o drv
ASAN_OPTIONS=detect_leaks=0 OMP_NUM_THREADS=4 ./drv sq.bin cu.bin
