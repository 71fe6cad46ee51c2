/*
 * knpemi_hip.h -- C ABI of libknpemi_hip.so: the MI355X-native (gfx950) replacement for the
 * third-party native code that the reference (hherlyng/knp-emi-cgx) calls on its per-timestep
 * KNP-EMI assemble-and-solve path.  Plain pointers and sizes only; no torch / DOLFINx / PETSc types.
 *
 * What each entry point replaces (file:line relative to the reference repository):
 *
 *   knp_create / knp_get_layout / knp_get_csr_pattern
 *        multiphenicsx.fem.DofMapRestriction + create_matrix_block / create_vector_block
 *        (src/CGx/KNPEMI/KNPEMIx_problem.py:75-94, src/CGx/KNPEMI/KNPEMIx_solver.py:157-161)
 *   knp_set_params                   constants of the forms (KNPEMIx_problem.py:459-462, 909-981)
 *   knp_set_program / knp_set_program_constants
 *        the FFCx-JIT-compiled membrane integrands built from IonicModel._eval
 *        (KNPEMIx_problem.py:504-555, 609-610, 641-642; KNPEMIx_ionic_model.py:_eval methods)
 *   knp_assemble_matrix              assemble_matrix_block(A, a)      (KNPEMIx_solver.py:110-115)
 *   knp_assemble_matrix_async        the same, next to the right-hand side chain (own stream)
 *   knp_assemble_rhs                 assemble_vector_block(b, L, a)   (KNPEMIx_solver.py:116)
 *   knp_assemble_precond             assemble_matrix_block(P)         (KNPEMIx_solver.py:118-127,
 *                                    form KNPEMIx_problem.py:657-744)
 *   knp_set_nullspace / knp_project_nullspace
 *        MatNullSpace create/test/remove                              (KNPEMIx_solver.py:297-335)
 *   knp_set_deflation                (multi-GPU only) restores what BoomerAMG's global coarse levels give the reference
 *   knp_pc_setup / knp_amg_*         PC setup: ksp.setUp() with pc_type hypre (KNPEMIx_solver.py:211-214,
 *                                    269-273, 386-389) -> vertex-block Jacobi / aggregation AMG
 *   knp_gmres_solve                  ksp.solve(b, x): GMRES(30), left PC, CGS, preconditioned norm
 *                                    (KNPEMIx_solver.py:435; options :276-280)
 *   knp_pack / knp_unpack            BlockVecSubVectorWrapper copies + phi_m = phi_i - phi_e
 *                                    (KNPEMIx_solver.py:177-209, 452-468)
 *   knp_hh_update                    HodgkinHuxley.update_gating_variables
 *                                    (KNPEMIx_ionic_model.py:605-671)
 *   knp_l2_norms                     assemble_scalar(inner(phi,phi)*dx(tag)) (src/CGx/KNPEMI/main.py:70-84)
 *   knp_set_comm                     the MPI calls hidden in PETSc/DOLFINx (ghost updates
 *                                    KNPEMIx_solver.py:439,459,468; Allreduce inside KSPSolve)
 *
 * Conventions
 *   - every function returns 0 on success, a negative KNP_E_* otherwise; knp_last_error() gives text.
 *     Nothing throws or exits across the ABI.  Non-convergence is a *reason code*, not an error.
 *   - a ctx is bound to the HIP device current at knp_create and to one stream (knp_set_stream);
 *     it is not thread-safe; one ctx per GPU.
 *   - the one-off host-side passes (graph build in knp_create, hierarchy hand-over in knp_amg_set_level*) are OpenMP loops sized to
 *     the CPU share of the process: cgroup quota or affinity mask, divided by LOCAL_WORLD_SIZE, at most 32; KNP_HOST_THREADS=<n>
 *     overrides.  Nothing on the per-step path uses host threads.
 *   - "host" pointers are read/written by the CPU during the call; "device" pointers are HBM
 *     addresses owned by the caller (e.g. torch tensors) unless stated otherwise.
 *   - unknown numbering: node = (vertex, side); DoF = 4*node + f, f = 0..2 ions, f = 3 potential.
 *     Nodes follow vertex order; a membrane vertex contributes its intra node then its extra node.
 *     Owned vertices come first (multi-GPU): rows exist for owned nodes only, columns may refer
 *     to ghost nodes; vectors have n_dof_local entries (owned part first).
 */
#ifndef KNPEMI_HIP_H
#define KNPEMI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KNP_MAX_IONS 3
#define KNP_MAX_AUX 8
#define KNP_MAX_PROG_REGS 48
#define KNP_MAX_AMG_LEVELS 16

#define KNP_OK 0
#define KNP_E_ARG (-1)
#define KNP_E_HIP (-2)
#define KNP_E_STATE (-3)
#define KNP_E_ALLOC (-4)
#define KNP_E_MESH (-5)

/* converged reasons (mirroring PETSc's KSPConvergedReason values) */
#define KNP_CONVERGED_RTOL 2
#define KNP_CONVERGED_ATOL 3
#define KNP_DIVERGED_ITS (-3)
#define KNP_DIVERGED_DTOL (-4)
#define KNP_DIVERGED_NANORINF (-9)

/* preconditioner kinds */
#define KNP_PC_NONE 0
#define KNP_PC_VBJACOBI 1 /* per-vertex 4x4 / 8x8 blocks of A, refreshed with A */
#define KNP_PC_AMG 2      /* multilevel V-cycle on P, hierarchy 0 = all fields (the reference's block-Jacobi form) */
#define KNP_PC_AMG_BT 3   /* block lower-triangular: hierarchy 0 = ion fields, then the potential (hierarchy 1) on
                             r_phi - A_{phi,k} z_k with a Cahouet-Chabard Schur term psi/(sum z^2 k)/M_lumped added */
#define KNP_PC_AMG_LT 4   /* the reference's P with use_block_jacobi=False (KNPEMIx_problem.py:720-722): P keeps the (phi,k) blocks
                             dt z_j D_j K (= those of A); applied as the block forward substitution z_k = V_k r_k,
                             z_phi = V_phi (r_phi - P_{phi,k} z_k) with the same two hierarchies, no Schur term */

/* membrane-program opcodes: instruction = {op, dst, a, b}, registers are doubles */
enum {
    KNP_OP_CONST = 0, /* dst = consts[a] */
    KNP_OP_KI = 1,    /* dst = k_i^a at the quadrature point */
    KNP_OP_KE = 2,    /* dst = k_e^a */
    KNP_OP_PHIM = 3,  /* dst = phi_m */
    KNP_OP_AUX = 4,   /* dst = aux field a (gating variables, ...) */
    KNP_OP_X = 5,     /* dst = coordinate a */
    KNP_OP_ADD = 6, KNP_OP_SUB = 7, KNP_OP_MUL = 8, KNP_OP_DIV = 9, KNP_OP_NEG = 10,
    KNP_OP_POW = 11, KNP_OP_LN = 12, KNP_OP_EXP = 13, KNP_OP_SQRT = 14,
    KNP_OP_MAX = 15, KNP_OP_MIN = 16, KNP_OP_ABS = 17,
    KNP_OP_LT = 18, KNP_OP_GT = 19, KNP_OP_LE = 20, KNP_OP_GE = 21, KNP_OP_EQ = 22,
    KNP_OP_AND = 23, KNP_OP_OR = 24, KNP_OP_NOT = 25,
    KNP_OP_SEL = 26,   /* dst = (reg[a] != 0) ? reg[b] : reg[dst]   (dst preloaded with the else-value) */
    KNP_OP_OUT = 27,   /* I_ch^a += reg[b] */
    KNP_OP_MOV = 28,
    KNP_OP_POWI = 29   /* dst = reg[a]^b with integer literal b */
};

typedef struct knp_ctx knp_ctx;

typedef struct {
    int32_t dim;              /* 2 (triangles) or 3 (tetrahedra) */
    int32_t n_vertices;       /* local vertices, owned first */
    int32_t n_vertices_owned; /* == n_vertices on one GPU */
    int32_t n_cells;
    int32_t n_cells_owned;    /* owned cells come first; used by knp_l2_norms only */
    const int32_t* cells;     /* host [n_cells*(dim+1)] local vertex ids */
    const double* coords;     /* host [n_vertices*dim], metres */
    const uint8_t* cell_side; /* host [n_cells] 0 = intracellular, 1 = extracellular */
    int32_t n_gamma;
    const int32_t* gamma;     /* host [n_gamma*4] (cell+, lf+, cell-, lf-), '+' = intracellular */
    const int32_t* gamma_prog;/* host [n_gamma] membrane-program id of the facet */
    int32_t n_q;              /* facet quadrature */
    const double* q_pts;      /* host [n_q*dim] barycentric coordinates on the facet */
    const double* q_w;        /* host [n_q] weights, sum 1 */
} knp_mesh_desc;

/* nodal fields (device pointers, each n_vertices doubles) */
typedef struct {
    const double* k_i[KNP_MAX_IONS];
    const double* k_e[KNP_MAX_IONS];
    const double* phi_m;
    const double* aux[KNP_MAX_AUX];
} knp_fields;

typedef struct {
    double* k_i[KNP_MAX_IONS];
    double* k_e[KNP_MAX_IONS];
    double* phi_i;
    double* phi_e;
    double* phi_m;
} knp_fields_out;

/* indices into the array filled by knp_get_sizes */
enum {
    KNP_SZ_N_NODES = 0, KNP_SZ_N_NODES_OWNED = 1, KNP_SZ_N_DOF_LOCAL = 2, KNP_SZ_N_DOF_OWNED = 3,
    KNP_SZ_NNZ = 4, KNP_SZ_N_PAIRS = 5, KNP_SZ_N_CONTRIB = 6, KNP_SZ_N_GAMMA_VERTS = 7,
    KNP_SZ_N_GAMMA_PAIRS = 8, KNP_SZ_NNZ_P = 9, KNP_SZ_N_PHI_OWNED = 10,
    KNP_SZ_NNZ_P_PHI = 11 /* entries of knp_get_precond_phi_csr */, KNP_SZ_COUNT = 16
};

/* communication hooks (multi-GPU); both receive DEVICE pointers */
typedef int (*knp_halo_fn)(void* user, double* x_local);              /* fill ghost part of x */
typedef int (*knp_allreduce_fn)(void* user, double* buf, int32_t n);  /* in-place SUM over ranks */
/* per-level exchange of a distributed AMG hierarchy; op 0: forward halo (fill the ghost entries of a level vector),
 * op 1: reverse halo (add ghost entries to their owners), op 2: all-reduce SUM of a replicated coarse vector */
typedef int (*knp_level_comm_fn)(void* user, int32_t hier, int32_t level, int32_t op, double* vec);

/* ---- lifetime ---- */
int knp_create(knp_ctx** out, const knp_mesh_desc* mesh);
int knp_destroy(knp_ctx* ctx);
const char* knp_last_error(const knp_ctx* ctx);
int knp_set_stream(knp_ctx* ctx, void* hip_stream);
int knp_set_comm(knp_ctx* ctx, knp_halo_fn halo, knp_allreduce_fn allreduce, void* user);
int knp_set_level_comm(knp_ctx* ctx, knp_level_comm_fn fn);

/* ---- native peer-to-peer exchange (multi-GPU, optional; replaces the hooks above once attached) ----
 * One kernel per exchange: it writes this rank's ghost values / reduction partials straight into the neighbours' mailboxes
 * (uncached device memory mapped over hipIpc: xGMI stores between GPUs), announces them with a sequence flag, waits for the
 * neighbours' flags (bounded) and consumes its own mailbox.  No host round trip, no library call per exchange.  The rendezvous (who are my peers, their IPC handles, where
 * my data lands in their mailbox) is done by the caller with whatever it has (torch.distributed, MPI):
 *   knp_p2p_init            once per ctx
 *   knp_p2p_plan_create     allocates this rank's mailbox of one plan, returns its 64-byte IPC handle
 *   knp_p2p_plan_connect    maps the peers' mailboxes; halo plans also get the index lists and the peers' offsets
 *   knp_p2p_attach          binds a connected plan to the fine halo, a level halo, a level's replicated all-reduce
 *                           or the reduction slots
 * All four are collective in the sense that every rank must make the same sequence of calls.  A wait that exceeds
 * the timeout sets an error that the next knp_gmres_solve / knp_p2p_test_* reports (KNP_E_STATE). */
#define KNP_P2P_HALO 0
#define KNP_P2P_ALLREDUCE 1
#define KNP_P2P_ATTACH_FINE_HALO 0
#define KNP_P2P_ATTACH_LEVEL_HALO 1
#define KNP_P2P_ATTACH_LEVEL_REPL 2
#define KNP_P2P_ATTACH_SLOTS 3
int knp_p2p_init(knp_ctx* ctx, int32_t rank, int32_t size, double timeout_seconds);
int knp_p2p_shutdown(knp_ctx* ctx);   /* drop every plan and return to the hooks (all ranks together) */
int knp_p2p_plan_create(knp_ctx* ctx, int32_t kind, int64_t n_fwd /* halo: ghost count | all-reduce: max length */,
                        int64_t n_rev /* halo: total send count */, int32_t* plan_out, void* ipc_handle_out /* host [64] */);
int knp_p2p_plan_connect(knp_ctx* ctx, int32_t plan, const void* handles /* host [size*64], one per rank */,
                         int32_t n_peers, const int32_t* peer_rank /* host, ascending */,
                         const int64_t* send_ptr /* host [n_peers+1] */, const int32_t* send_idx /* host: owned entries each peer needs */,
                         const int64_t* remote_fwd_off /* host [2*n_peers]: element offset of my values in the peer's mailbox data
                                                          area, for sequence parity 0 and 1 (the mailboxes are double buffered) */,
                         const int64_t* recv_ptr /* host [n_peers+1] */, const int32_t* recv_idx /* host: my ghost entries per peer */,
                         const int64_t* remote_rev_off /* host [2*n_peers]: the same for the reverse (ghost -> owner) direction */);
int knp_p2p_attach(knp_ctx* ctx, int32_t what, int32_t hier, int32_t level, int32_t plan /* -1 detaches */);
int knp_p2p_test_halo(knp_ctx* ctx, int32_t plan, double* x /* device */, int32_t reverse);
int knp_p2p_test_allreduce(knp_ctx* ctx, int32_t plan, double* v /* device */, int32_t n);

/* ---- description ---- */
int knp_get_sizes(const knp_ctx* ctx, int64_t* sizes /* host [KNP_SZ_COUNT] */);
int knp_get_layout(const knp_ctx* ctx, int32_t* node_i, int32_t* node_e /* host [n_vertices] each */);
int knp_get_csr_pattern(const knp_ctx* ctx, int32_t* rowptr, int32_t* colind /* host */);
/* A is held pair-major on the device; these two export it as standard CSR for checkers: rows = owned DoFs,
 * row (n,j<3) = [(nb,j),(nb,phi) for nb in the node's pairs] ++ [(cross,phi)], row (n,phi) = [(nb,0..3) ...] ++ [(cross,phi)] */
int knp_get_csr_values(const knp_ctx* ctx, double* vals /* host [nnz] */);
int knp_get_precond_csr(const knp_ctx* ctx, int32_t* rowptr, int32_t* colind, double* vals /* host */);
/* max |A_ij| over the locally stored entries (what the reference's null-space check scales by; device reduction) */
int knp_matrix_max_abs(knp_ctx* ctx, double* out /* host */);

/* ---- problem data ---- */
int knp_set_params(knp_ctx* ctx, double dt, double F, double C_M, double psi, int32_t n_ions,
                   const double* z, const double* Di, const double* De);
int knp_set_program(knp_ctx* ctx, int32_t prog_id, int32_t n_instr, const int32_t* code /* host [n_instr*4] */,
                    int32_t n_consts, const double* consts /* host */);
/* "native" when the membrane programs were compiled with hiprtc for this device (knp_jit.cpp), otherwise the reason why the
 * bytecode interpreter runs (KNP_JIT=0, no libhiprtc, compile log).  Both run on the GPU and give the same values. */
const char* knp_jit_status(knp_ctx* ctx);
/* test hook, needs neither a device nor a context: generate the HIP source for one program and compile it with hiprtc
 * for `arch` (e.g. "gfx950"); 0 on success, `log` receives the compiler log or a one-line summary */
int knp_jit_compile_check(const int32_t* code, int32_t n_instr, const char* arch, char* log, int32_t log_cap);
/* test hook, needs neither a device nor a context: the number of host threads the one-off OpenMP passes of the library use in this
 * process (KNP_HOST_THREADS, else cgroup CPU quota / affinity mask divided by LOCAL_WORLD_SIZE, at most 32; evaluated once) */
int knp_host_thread_count(void);
int knp_set_program_constants(knp_ctx* ctx, int32_t prog_id, int32_t n_consts, const double* consts);
/* Dirichlet conditions (reference: dfx.fem.dirichletbc + bcs= of assemble_*_block, KNPEMIx_problem.py:106-134,
 * KNPEMIx_solver.py:114-116): rows of the listed owned DoFs become identity rows in A and P at every assembly;
 * the caller sets b[dof] = g.  (DOLFINx also eliminates the columns; same solution.) n = 0 clears. host array. */
int knp_set_dirichlet(knp_ctx* ctx, int32_t n, const int32_t* dofs);
/* volumetric source terms dt*f (KNPEMIx_problem.py:613-614); NULL pointers mean zero. nodal, device. */
int knp_set_sources(knp_ctx* ctx, const double* const* f_i, const double* const* f_e);

/* ---- per-timestep assembly ---- */
int knp_assemble_matrix(knp_ctx* ctx, const knp_fields* fields);
/* the same assembly (assemble_matrix_block(A, a), KNPEMIx_solver.py:110-115) on the library's own stream: A depends on the previous
 * solution only, so the right-hand side chain of the step (gating update, knp_assemble_rhs, knp_gmres_prepare) can be enqueued
 * while it runs.  The fields must not be modified until the next call that needs A (knp_gmres_solve, knp_spmv, exports), which
 * joins it.  In line (== knp_assemble_matrix) on the first assembly, with vertex-block Jacobi or when kernel classes are timed. */
int knp_assemble_matrix_async(knp_ctx* ctx, const knp_fields* fields);
int knp_assemble_rhs(knp_ctx* ctx, const knp_fields* fields, double* b /* device [n_dof_local] */);
int knp_assemble_precond(knp_ctx* ctx, const knp_fields* fields);
/* form of the potential block of P: 0 the reference's (- (C_M/F) M_Gamma per side, sides uncoupled, KNPEMIx_problem.py:735-738),
 * 1 the potential block of A at assembly time (+ (C_M/F) M_Gamma and the phi_i-phi_e coupling, :637-638).  Before knp_assemble_precond. */
int knp_pc_set_coupled_potential(knp_ctx* ctx, int32_t on);
/* the potential block of P on node-indexed rows / columns: CSR [n_nodes_owned + 1], [KNP_SZ_NNZ_P_PHI] x 2, columns unsorted */
int knp_get_precond_phi_csr(const knp_ctx* ctx, int32_t* rowptr, int32_t* colind, double* vals);

/* ---- linear algebra ---- */
int knp_spmv(knp_ctx* ctx, const double* x, double* y); /* y(owned) = A x ; calls the halo hook first */
int knp_set_nullspace(knp_ctx* ctx, int32_t on);
int knp_project_nullspace(knp_ctx* ctx, double* v);
int knp_nullspace_test(knp_ctx* ctx, double* out_norm /* host: ||A ns||_2 */);
int knp_pc_setup(knp_ctx* ctx, int32_t kind);
int knp_pc_apply(knp_ctx* ctx, const double* r, double* z); /* r, z: [n_dof_local]; distributed contexts overwrite the ghost entries of r (see knp_gmres_solve) */
/* Additive coarse correction for near-null modes the per-GPU preconditioner blocks cannot see (constants of a
 * potential block on a connected component that the partition cuts): z += Z Einv Z^T r with Z the indicator
 * vectors of the potential DoFs of each mode. node_mode: host [n_nodes_owned], -1 = not deflated.
 * Einv: host [n_modes^2] (pseudo-)inverse of Z^T A Z, identical on every rank. n_modes = 0 disables. */
int knp_set_deflation(knp_ctx* ctx, int32_t n_modes, const int32_t* node_mode, const double* einv);
/* AMG hierarchies (hier 0 or 1) supplied level by level (level 0 = finest = P itself, possibly restricted to a
 * field class by zero rows). All arrays HOST; copied. */
int knp_amg_reset(knp_ctx* ctx, int32_t hier, int32_t n_levels, int32_t pre_sweeps, int32_t post_sweeps, int32_t cheby_degree);
/* Optional, between knp_amg_reset and the levels: the hierarchy has nf (3 or 4) decoupled fields per node that share one
 * sparsity pattern on every level (node-synchronised aggregation: level-0 unknowns 4*node + field, coarse unknowns
 * nf*aggregate + field, rows sorted by column).  With fp32 storage the library then keeps node-blocked copies of the
 * restrictors, the coarse level operators and S (one column index per node entry, nf values behind it) and the fused cycle
 * runs on them.  The pattern is verified level by level; a level that does not have it keeps the whole hierarchy on the
 * scalar kernels (KNP_ST_BLOCKED tells).  What PCGAMG/hypre reach with a block size of the matrix (MatSetBlockSize) for the
 * reference's preconditioner matrix (KNPEMI/KNPEMIx_solver.py:78-101). */
int knp_amg_set_node_fields(knp_ctx* ctx, int32_t hier, int32_t nf);
int knp_amg_set_level(knp_ctx* ctx, int32_t hier, int32_t level, int32_t n_rows, int32_t n_cols_halo,
                      const int32_t* A_rowptr, const int32_t* A_colind, const double* A_vals,
                      const double* inv_diag, double lambda_max,
                      int32_t n_coarse,
                      const int32_t* P_rowptr, const int32_t* P_colind, const double* P_vals,
                      const int32_t* R_rowptr, const int32_t* R_colind, const double* R_vals);
/* distributed levels: replace the prolongator by one that also has rows for the ghost entries (n_rows_P = local size).  The
 * ghost part of the iterate then stays current through the coarse correction and the halo before the first post-smoothing
 * step is skipped. */
int knp_amg_set_level_prolongator(knp_ctx* ctx, int32_t hier, int32_t level, int32_t n_rows_P, const int32_t* P_rp,
                                  const int32_t* P_ci, const double* P_v);
/* distributed hierarchy: `distributed` != 0 -> the level operator has ghost columns (n_cols_halo local columns);
 * repl_n > 0 -> the next level is replicated on all ranks with repl_n unknowns */
int knp_amg_set_level_mode(knp_ctx* ctx, int32_t hier, int32_t level, int32_t distributed, int32_t repl_n);
int knp_amg_set_coarse(knp_ctx* ctx, int32_t hier, int32_t n, const double* dense_inverse /* host [n*n] row-major */);
/* optional, per level with a coarser level below it: S = (I - c2 Dinv A) Pprol with c2 = 1 / (0.6 lambda_max), CSR with n_rows rows
 * (HOST arrays, copied).  With S on every level, V(1,1) / Chebyshev degree 1 and level 0 on the library's own P
 * (knp_amg_use_native_level0) the cycle runs in its fused form: pre-smoothing + residual as one gather, prolongation +
 * post-smoothing as one gather per level (same operator as the unfused cycle; KNP_FUSED=0 selects the latter). */
int knp_amg_set_level_smoothed(knp_ctx* ctx, int32_t hier, int32_t level, int32_t n_rows, const int32_t* S_rowptr, const int32_t* S_colind,
                        const double* S_vals);
/* optional, intermediate levels (1 .. n_levels-2) of a single-GPU hierarchy whose cycle runs fused: both legs as plain products,
 * Rt = R (I - c A Dinv) [n_coarse x n] and U = [c Dinv (2I - c A Dinv) | S] [n x (n + n_coarse)] (cgx_hip/amg.py
 * coarse_fused_operators); replaces restriction + residual and the three-vector up-leg of those levels */
int knp_amg_set_level_coarse_fused(knp_ctx* ctx, int32_t hier, int32_t level, int32_t Rt_rows, const int32_t* Rt_rowptr, const int32_t* Rt_colind,
                                   const double* Rt_vals, int32_t U_rows, const int32_t* U_rowptr, const int32_t* U_colind, const double* U_vals);
/* level 0 == the library's own P (owned block): use its pair-major storage and node kernels instead of the uploaded
 * CSR. mode: 0 off, 1 all four fields, 2 ion fields only, 3 potential only, 4 potential only on node-indexed vectors with the
 * UPLOADED level-0 operator (one GPU; the operator may couple the two sides of the membrane, knp_pc_set_coupled_potential) */
int knp_amg_use_native_level0(knp_ctx* ctx, int32_t hier, int32_t mode);
/* mixed-precision preconditioner: level operators, transfer operators and P on level 0 are STORED in fp32
 * (converted on load); every vector, the dense coarse inverse, the system matrix and the Krylov process stay fp64.
 * Call before uploading a hierarchy. */
int knp_amg_set_precision(knp_ctx* ctx, int32_t fp32_storage);
/* optional: start ||B b|| of the next knp_gmres_solve(ctx, b, ...) on a side stream now (b complete, not to be modified until
 * the solve); it then overlaps whatever the caller enqueues next, typically knp_assemble_matrix.  No-op before the first solve
 * and on distributed contexts whose exchanges go through the hooks (with the native peer-to-peer plans attached everywhere it
 * runs: the exchange kernels are then ordered on the side stream on every rank).  Any other call that touches the
 * preconditioner or b joins / discards it. */
int knp_gmres_prepare(knp_ctx* ctx, const double* b);
/* GMRES(restart), 1 <= restart <= 55 (the per-iteration reduction carries restart + 2 values in the library's 57 Gram-Schmidt
 * slots; anything larger is KNP_E_ARG).  b and x are [n_dof_local] device vectors.  On distributed contexts the GHOST entries
 * [n_dof_owned, n_dof_local) of b -- and of r in knp_pc_apply, b in knp_gmres_prepare -- are OVERWRITTEN by the halo exchange of
 * the preconditioner's fused level-0 leg although the arguments are const-qualified (the owned entries are never written);
 * on one GPU n_dof_local == n_dof_owned and nothing is written. */
int knp_gmres_solve(knp_ctx* ctx, const double* b, double* x, double rtol, double atol, int32_t max_it,
                    int32_t restart, int32_t* its, double* rnorm, int32_t* reason);

/* ---- state transfer ---- */
int knp_pack(knp_ctx* ctx, const knp_fields_out* fields, double* x);
int knp_unpack(knp_ctx* ctx, const double* x, const knp_fields_out* fields);
int knp_hh_update(knp_ctx* ctx, const double* phi_m, double* n, double* m, double* h, int32_t count,
                  double dt, double phi_rest, int32_t rush_larsen, int32_t substeps);
int knp_l2_norms(knp_ctx* ctx, const double* phi_i, const double* phi_e, double* out /* host [2]: squared, owned cells */);

/* ---- instrumentation ---- */
/* elapsed ms and launch count of a kernel class since the last reset (HIP events on the ctx stream).
 * classes: 0 spmv, 1 orthogonalisation, 2 pc, 3 assembly, 4 other */
/* step timers of the host loop (reference: perf_counter + allreduce(MAX) around assembly and solve, KNPEMIx_solver.py:402-413,
 * 434-449): knp_timer_mark records one event on the stream (join_assembly != 0: after joining knp_assemble_matrix_async);
 * knp_timer_read synchronises once and returns the seconds between consecutive marks (n marks -> n - 1 intervals), then
 * forgets them; knp_timer_pending = marks recorded and not yet read. */
int knp_timer_mark(knp_ctx* ctx, int32_t join_assembly);
int knp_timer_read(knp_ctx* ctx, int32_t capacity, double* seconds /* host [capacity] */, int32_t* n_intervals);
int knp_timer_pending(const knp_ctx* ctx);
int knp_profile_enable(knp_ctx* ctx, int32_t class_mask); /* bit k enables class k; 0 disables */
int knp_profile_get(knp_ctx* ctx, int32_t cls, double* ms, int64_t* launches);
int knp_profile_reset(knp_ctx* ctx);
/* counters of the linear solves since the last knp_profile_reset: what PETSc's -log_view reports for the KSPSolve stage
 * (VecMDot/VecNorm reductions, VecScatter halos) behind KNPEMIx_solver.py:435 */
enum { KNP_ST_BNORM = 0 /* ||B b|| of the last solve */, KNP_ST_ALLREDUCE = 1 /* reductions over ranks */,
       KNP_ST_HALO = 2 /* fine-level halo exchanges */, KNP_ST_READBACK = 3 /* host waits on a reduced value */,
       KNP_ST_FUSED = 4 /* bit h set: hierarchy h runs the fused V(1,1) cycle; bit 2+h: its level 0 runs fused inside the
                           level-by-level cycle (distributed hierarchies) */,
       KNP_ST_NORM_FALLBACK = 5 /* GMRES iterations whose norm needed a second reduction (cancellation guard) */,
       KNP_ST_BLOCKED = 6 /* bit h set: the fused cycle of hierarchy h runs on node-blocked operators */,
       KNP_ST_FUSED_LEVELS = 7 /* levels >= 1 that run in fused form inside the level-by-level cycle, all hierarchies */, KNP_ST_COUNT = 8 };
int knp_get_stats(const knp_ctx* ctx, double* out /* host [KNP_ST_COUNT] */);
/* bytes the kernels of one application must move, from the sizes of the arrays they read and write (per-class roofline): [0] SpMV on A,
 * [1] one preconditioner application, [2] matrix assembly of one step, [3] right-hand side assembly, [4] one owned vector */
int knp_get_traffic_model(const knp_ctx* ctx, double* out /* host [5] */);

#ifdef __cplusplus
}
#endif
#endif /* KNPEMI_HIP_H */
