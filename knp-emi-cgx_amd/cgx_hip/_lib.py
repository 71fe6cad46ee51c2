"""ctypes binding of libknpemi_hip.so (see include/knpemi_hip.h for the ABI contract).

The product path has no CPU fallback: if the shared library is missing this module raises
at first use, and every compute entry point fails loudly when no HIP device is present.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libknpemi_hip.so")

KNP_MAX_IONS = 3
KNP_MAX_AUX = 8
KNP_MAX_PROG_REGS = 48
KNP_SZ_COUNT = 16
(SZ_N_NODES, SZ_N_NODES_OWNED, SZ_N_DOF_LOCAL, SZ_N_DOF_OWNED, SZ_NNZ, SZ_N_PAIRS, SZ_N_CONTRIB,
 SZ_N_GAMMA_VERTS, SZ_N_GAMMA_PAIRS, SZ_NNZ_P, SZ_N_PHI_OWNED, SZ_NNZ_P_PHI) = range(12)

PC_NONE, PC_VBJACOBI, PC_AMG, PC_AMG_BT, PC_AMG_LT = 0, 1, 2, 3, 4

OPS = dict(CONST=0, KI=1, KE=2, PHIM=3, AUX=4, X=5, ADD=6, SUB=7, MUL=8, DIV=9, NEG=10, POW=11, LN=12,
           EXP=13, SQRT=14, MAX=15, MIN=16, ABS=17, LT=18, GT=19, LE=20, GE=21, EQ=22, AND=23, OR=24,
           NOT=25, SEL=26, OUT=27, MOV=28, POWI=29)

REASONS = {2: "CONVERGED_RTOL", 3: "CONVERGED_ATOL", -3: "DIVERGED_ITS", -4: "DIVERGED_DTOL",
           -9: "DIVERGED_NANORINF", 0: "ITERATING"}

i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)
f64p = C.POINTER(C.c_double)
u8p = C.POINTER(C.c_uint8)
vp = C.c_void_p


class MeshDesc(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_vertices", C.c_int32), ("n_vertices_owned", C.c_int32),
                ("n_cells", C.c_int32), ("n_cells_owned", C.c_int32), ("cells", i32p), ("coords", f64p),
                ("cell_side", u8p), ("n_gamma", C.c_int32), ("gamma", i32p), ("gamma_prog", i32p),
                ("n_q", C.c_int32), ("q_pts", f64p), ("q_w", f64p)]


class Fields(C.Structure):
    _fields_ = [("k_i", vp * KNP_MAX_IONS), ("k_e", vp * KNP_MAX_IONS), ("phi_m", vp), ("aux", vp * KNP_MAX_AUX)]


class FieldsOut(C.Structure):
    _fields_ = [("k_i", vp * KNP_MAX_IONS), ("k_e", vp * KNP_MAX_IONS), ("phi_i", vp), ("phi_e", vp), ("phi_m", vp)]


HALO_FN = C.CFUNCTYPE(C.c_int, vp, vp)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, vp, vp, C.c_int32)
LEVEL_COMM_FN = C.CFUNCTYPE(C.c_int, vp, C.c_int32, C.c_int32, C.c_int32, vp)

# every symbol include/knpemi_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "knp_create": (C.c_int, [C.POINTER(vp), C.POINTER(MeshDesc)]),
    "knp_destroy": (C.c_int, [vp]),
    "knp_last_error": (C.c_char_p, [vp]),
    "knp_set_stream": (C.c_int, [vp, vp]),
    "knp_set_comm": (C.c_int, [vp, HALO_FN, ALLREDUCE_FN, vp]),
    "knp_set_level_comm": (C.c_int, [vp, LEVEL_COMM_FN]),
    "knp_get_sizes": (C.c_int, [vp, i64p]),
    "knp_get_layout": (C.c_int, [vp, i32p, i32p]),
    "knp_get_csr_pattern": (C.c_int, [vp, i32p, i32p]),
    "knp_get_csr_values": (C.c_int, [vp, f64p]),
    "knp_get_precond_csr": (C.c_int, [vp, i32p, i32p, f64p]),
    "knp_matrix_max_abs": (C.c_int, [vp, f64p]),
    "knp_set_params": (C.c_int, [vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, f64p, f64p, f64p]),
    "knp_set_program": (C.c_int, [vp, C.c_int32, C.c_int32, i32p, C.c_int32, f64p]),
    "knp_set_program_constants": (C.c_int, [vp, C.c_int32, C.c_int32, f64p]),
    "knp_set_dirichlet": (C.c_int, [vp, C.c_int32, i32p]),
    "knp_set_sources": (C.c_int, [vp, C.POINTER(vp), C.POINTER(vp)]),
    "knp_assemble_matrix": (C.c_int, [vp, C.POINTER(Fields)]),
    "knp_assemble_matrix_async": (C.c_int, [vp, C.POINTER(Fields)]),
    "knp_assemble_rhs": (C.c_int, [vp, C.POINTER(Fields), vp]),
    "knp_assemble_precond": (C.c_int, [vp, C.POINTER(Fields)]),
    "knp_pc_set_coupled_potential": (C.c_int, [vp, C.c_int32]),
    "knp_get_precond_phi_csr": (C.c_int, [vp, i32p, i32p, f64p]),
    "knp_spmv": (C.c_int, [vp, vp, vp]),
    "knp_set_nullspace": (C.c_int, [vp, C.c_int32]),
    "knp_project_nullspace": (C.c_int, [vp, vp]),
    "knp_nullspace_test": (C.c_int, [vp, f64p]),
    "knp_pc_setup": (C.c_int, [vp, C.c_int32]),
    "knp_pc_apply": (C.c_int, [vp, vp, vp]),
    "knp_set_deflation": (C.c_int, [vp, C.c_int32, i32p, f64p]),
    "knp_amg_reset": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "knp_amg_set_level": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, i32p, i32p, f64p, f64p, C.c_double,
                                    C.c_int32, i32p, i32p, f64p, i32p, i32p, f64p]),
    "knp_amg_set_level_mode": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "knp_amg_set_level_prolongator": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                               C.POINTER(C.c_double)]),
    "knp_amg_set_coarse": (C.c_int, [vp, C.c_int32, C.c_int32, f64p]),
    "knp_amg_set_level_smoothed": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, i32p, i32p, f64p]),
    "knp_amg_set_level_coarse_fused": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, i32p, i32p, f64p, C.c_int32, i32p, i32p, f64p]),
    "knp_amg_set_node_fields": (C.c_int, [vp, C.c_int32, C.c_int32]),
    "knp_amg_use_native_level0": (C.c_int, [vp, C.c_int32, C.c_int32]),
    "knp_amg_set_precision": (C.c_int, [vp, C.c_int32]),
    "knp_jit_status": (C.c_char_p, [vp]),
    "knp_gmres_prepare": (C.c_int, [vp, vp]),
    "knp_jit_compile_check": (C.c_int, [C.POINTER(C.c_int32), C.c_int32, C.c_char_p, C.c_char_p, C.c_int32]),
    "knp_host_thread_count": (C.c_int, []),
    "knp_p2p_init": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_double]),
    "knp_p2p_shutdown": (C.c_int, [vp]),
    "knp_p2p_plan_create": (C.c_int, [vp, C.c_int32, C.c_int64, C.c_int64, C.POINTER(C.c_int32), vp]),
    "knp_p2p_plan_connect": (C.c_int, [vp, C.c_int32, vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                      C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                                      C.POINTER(C.c_int64)]),
    "knp_p2p_attach": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "knp_p2p_test_halo": (C.c_int, [vp, C.c_int32, vp, C.c_int32]),
    "knp_p2p_test_allreduce": (C.c_int, [vp, C.c_int32, vp, C.c_int32]),
    "knp_gmres_solve": (C.c_int, [vp, vp, vp, C.c_double, C.c_double, C.c_int32, C.c_int32, i32p, f64p, i32p]),
    "knp_pack": (C.c_int, [vp, C.POINTER(FieldsOut), vp]),
    "knp_unpack": (C.c_int, [vp, vp, C.POINTER(FieldsOut)]),
    "knp_hh_update": (C.c_int, [vp, vp, vp, vp, vp, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32]),
    "knp_l2_norms": (C.c_int, [vp, vp, vp, f64p]),
    "knp_timer_mark": (C.c_int, [vp, C.c_int32]),
    "knp_timer_read": (C.c_int, [vp, C.c_int32, f64p, C.POINTER(C.c_int32)]),
    "knp_timer_pending": (C.c_int, [vp]),
    "knp_profile_enable": (C.c_int, [vp, C.c_int32]),
    "knp_profile_get": (C.c_int, [vp, C.c_int32, f64p, i64p]),
    "knp_profile_reset": (C.c_int, [vp]),
    "knp_get_stats": (C.c_int, [vp, f64p]),
    "knp_get_traffic_model": (C.c_int, [vp, f64p]),
}

_lib = None


class KnpError(RuntimeError):
    pass


def load():
    """Load libknpemi_hip.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KnpError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback for the KNP-EMI hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx, rc):
    if rc != 0:
        msg = load().knp_last_error(ctx)
        raise KnpError(f"libknpemi_hip error {rc}: {msg.decode() if msg else ''}")
