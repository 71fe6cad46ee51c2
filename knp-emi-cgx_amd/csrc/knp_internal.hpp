// Internal structures of libknpemi_hip.so (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "knpemi_hip.h"

#define KNP_NPROF 5

// ---- host-side result of the graph builder (knp_setup.cpp) -------------------------------
struct KnpHostGraph {
    int dim = 0, nv1 = 0;
    int n_v = 0, n_v_owned = 0, n_c = 0, n_c_owned = 0, n_g = 0, n_q = 0;
    int n_nodes = 0, n_nodes_owned = 0;
    std::vector<int32_t> node_i, node_e;          // [n_v]
    std::vector<int32_t> node_vertex;             // [n_nodes]
    std::vector<uint8_t> node_side;               // [n_nodes]
    // same-side node graph (owned rows)
    std::vector<int32_t> pair_ptr;                // [n_nodes_owned+1]
    std::vector<int32_t> pair_col;                // [n_pairs] node ids (sorted per row)
    std::vector<int32_t> pair_row;                // [n_pairs]
    std::vector<double> pair_M, pair_K;           // [n_pairs]
    std::vector<int32_t> contrib_ptr;             // [n_pairs+1]
    std::vector<int32_t> contrib_cell;            // [n_contrib]
    std::vector<double> contrib_k;                // [n_contrib]
    // same-side cells of every owned node and, per contribution, the position of its cell in the list of the pair's row node:
    // the assembly stages the node's cell means in LDS once instead of gathering them per contribution
    std::vector<int32_t> node_cell_ptr;           // [n_nodes_owned+1]
    std::vector<int32_t> node_cell;               // [sum]
    std::vector<uint8_t> contrib_slot;            // [n_contrib]
    int max_node_cells = 0;
    // membrane graph
    std::vector<int32_t> fv;                      // [n_g*dim] facet vertices
    std::vector<double> fmeas;                    // [n_g]
    int n_gv = 0;                                 // owned membrane vertices
    std::vector<int32_t> gv_vertex, gv_node_i, gv_node_e;  // [n_gv]
    std::vector<int32_t> node_gv;                 // [n_nodes_owned] index of the membrane vertex or -1
    std::vector<int32_t> gptr;                    // [n_gv+1]
    std::vector<int32_t> gcol;                    // [n_gp] membrane-vertex neighbours as VERTEX ids
    std::vector<int32_t> grow;                    // [n_gp] owning membrane-vertex index
    std::vector<int32_t> gq_i, gq_e;              // [n_gp] slot of the neighbour in the same-side rows
    std::vector<int32_t> gx_i, gx_e;              // [n_gp] node ids of the cross columns (extra node for the intra row, ...)
    std::vector<int32_t> gdiag;                   // [n_gv] slot with gcol == own vertex
    std::vector<int32_t> gcptr;                   // [n_gp+1]
    std::vector<int32_t> gc_facet;                // [n_gc]
    std::vector<int32_t> gc_lab;                  // [n_gc] la*4+lb
    // CSR pattern of A: export only (knp_get_csr_*), built on demand by knp_build_csr_pattern -- the device never reads it
    int64_t nnz = 0;
    std::vector<int32_t> rowptr;                  // [4*n_nodes_owned+1]
    std::vector<int32_t> colind;                  // [nnz]
    std::string error;
};

int knp_build_graph(const knp_mesh_desc* m, KnpHostGraph& g);
int knp_build_csr_pattern(KnpHostGraph& g);
struct knp_ctx;
void knp_jit_build(knp_ctx* ctx);
void knp_jit_release(knp_ctx* ctx);

// ---- device-side program ------------------------------------------------------------------
struct KnpProgram {
    int n_instr = 0, n_consts = 0, n_regs = 0;
    int32_t* d_code = nullptr;
    double* d_consts = nullptr;
    double* h_consts = nullptr;   // pinned staging copy of the constants (per-step refresh without a synchronisation)
    std::vector<int32_t> h_code;  // host copy of the bytecode (source of the run-time compiled kernel, knp_jit.cpp)
};

struct knp_ctx;
// ---- native peer-to-peer communication (knp_p2p.hip) ----
#define KNP_P2P_MAXPEERS 16
struct KnpP2PPlan {
    int kind = 0;                       // 0 halo, 1 all-reduce
    int64_t n_fwd = 0, n_rev = 0;       // halo: ghost count / send total; all-reduce: n_max / unused
    char* box = nullptr;                // own mailbox: uncached device memory exported over IPC
    size_t box_bytes = 0, hdr_bytes = 0;
    std::vector<char*> peer_box;        // [size] mapped mailboxes of the other ranks (nullptr: not a peer; own rank: box)
    int n_peers = 0;
    int peer_rank[KNP_P2P_MAXPEERS];
    int64_t send_ptr[KNP_P2P_MAXPEERS + 1], recv_ptr[KNP_P2P_MAXPEERS + 1];
    std::vector<int64_t> remote_fwd_off, remote_rev_off;   // [2*n_peers]: element offset in the peer's data area per parity
    int32_t *d_send_idx = nullptr, *d_recv_idx = nullptr;
    int64_t n_rev_dst = 0;              // reverse halo: distinct owned entries, their mailbox positions in peer order
    int32_t *d_rev_dst = nullptr, *d_rev_ptr = nullptr, *d_rev_pos = nullptr;
    unsigned int* d_counter = nullptr;  // "last block done" counters of the pack kernels
    int64_t seq_fwd = 0, seq_rev = 0;   // message sequence numbers (identical on all ranks: calls are collective)
    bool connected = false;
};
struct KnpP2P {
    int rank = 0, size = 1;
    int64_t timeout_ticks = 0;          // wall_clock64 ticks a wait may take before it is reported
    int* h_err = nullptr;               // pinned, device-visible: bit 0 = a wait timed out
    int* h_err_dev = nullptr;
    int* d_err = nullptr;               // device twin: later waits give up at once after the first timeout
    std::vector<KnpP2PPlan> plans;
};
int knp_p2p_halo_forward(knp_ctx* ctx, int plan, double* x);
int knp_p2p_halo_reverse(knp_ctx* ctx, int plan, double* x);
int knp_p2p_allreduce(knp_ctx* ctx, int plan, double* v, int n, double* mirror, int64_t* seq_dev, int64_t seq_val);
int knp_p2p_check(knp_ctx* ctx);
void knp_p2p_free(knp_ctx* ctx);

// Node-blocked CSR of a hierarchy whose nf fields per node are decoupled and share one sparsity pattern (node-synchronised
// aggregation, cgx_hip/amg.py build_hierarchy(node_fields=...)): one column NODE per entry, nf values behind it.  16 B per entry
// for nf == 3 ({v0, v1, v2, column} in one float4) instead of 3 x (4 B column + 4 B value), and one gather of nf consecutive
// unknowns instead of nf separate ones.  fp32 values only (mixed-precision preconditioner storage).
struct KnpBlockedCsr {
    int n_rows = 0;          // node rows
    int32_t* rp = nullptr;   // [n_rows + 1]
    float4* ev = nullptr;    // nf == 3: {v0, v1, v2, bit pattern of the column node}; nf == 4: {v0, v1, v2, v3}
    int32_t* ci = nullptr;   // nf == 4 only
    int lanes = 4;
    int64_t nnz = 0;         // node entries
};

struct KnpAmgLevel {
    int n = 0, n_coarse = 0;
    int n_loc = 0;   // local columns = owned + ghost (== n on one GPU)
    int dist = 0;    // operator has ghost columns: forward halo before every application
    int repl_n = 0;  // > 0: the next level is replicated on all ranks with this many unknowns (all-reduce after restriction)
    int32_t *A_rp = nullptr, *A_ci = nullptr;
    double* A_v = nullptr;
    float *A_vf = nullptr, *P_vf = nullptr, *R_vf = nullptr;   // fp32 copies (mixed-precision preconditioner storage)
    int p2p_halo = -1, p2p_repl = -1;   // native exchange plans of this level (-1: hook)
    int P_rows = 0;                     // rows of the prolongator: n, or n_loc when it also prolongates the ghost entries
    int P_n_act = 0;                    // > 0: compact list of the non-empty prolongator rows (field-restricted hierarchies)
    int32_t *P_act_rows = nullptr, *P_act_rp = nullptr;
    double* inv_diag = nullptr;
    double lambda_max = 1.0;
    int32_t *P_rp = nullptr, *P_ci = nullptr;
    double* P_v = nullptr;
    int32_t *R_rp = nullptr, *R_ci = nullptr;
    double* R_v = nullptr;
    double *x = nullptr, *b = nullptr, *r = nullptr, *d = nullptr, *r2 = nullptr;  // work vectors (x / r2 ping-pong)
    double *xs = nullptr, *bs = nullptr, *rs = nullptr, *ds = nullptr, *r2s = nullptr;   // second set: the side-stream cycle of knp_gmres_prepare
    int A_lanes = 8, P_lanes = 4, R_lanes = 8;
    // fused cycle: S = (I - c2 Dinv A) Pprol, rows = this level's rows (compact list of the non-empty ones when S_n_act > 0)
    int32_t *S_rp = nullptr, *S_ci = nullptr;
    double* S_v = nullptr;
    float* S_vf = nullptr;
    int S_rows = 0, S_n_act = 0, S_lanes = 4;
    int32_t *S_act_rows = nullptr, *S_act_rp = nullptr;
    // level 0 of a potential-only hierarchy on compact vectors [n_nodes]: node indices instead of 4*node+3
    int32_t *R_ci_c = nullptr, *S_act_rows_c = nullptr;
    double* dinv_c = nullptr;
    // levels 1 .. n-2 of the fused cycle as two plain sparse products (knp_amg_set_level_coarse_fused; cgx_hip/amg.py
    // coarse_fused_operators): b_{l+1} = Rt b_l on the way down, x_l = U [b_l ; x_{l+1}] on the way up.  `cat` holds b_l followed by
    // x_{l+1} (n + n_coarse entries) so that the up-leg is an ordinary product with one input vector; `cats` = side-stream twin
    int32_t *Rt_rp = nullptr, *Rt_ci = nullptr, *U_rp = nullptr, *U_ci = nullptr;
    double *Rt_v = nullptr, *U_v = nullptr;
    float *Rt_vf = nullptr, *U_vf = nullptr;
    int Rt_lanes = 8, U_lanes = 8;
    KnpBlockedCsr bRt, bU;
    double *cat = nullptr, *cats = nullptr;
    KnpBlockedCsr bA, bR, bS;   // node-blocked copies (hierarchies with node_nf > 0, fp32 storage): level operator, restrictor, S
    // levels >= 1 in fused form inside the level-by-level cycle (distributed hierarchies and their replicated tails):
    // At = c A Dinv on the pattern of A (ghost columns scaled with the ghost inverse diagonal), knp_pc_setup builds it
    int64_t A_nnz = 0, R_nnz = 0, S_nnz = 0, Rt_nnz = 0, U_nnz = 0;
    int lfused = 0;
    double* At_v = nullptr;
    float* At_vf = nullptr;
};

#define KNP_MAX_HIER 2
struct KnpAmgHier {
    int levels = 0, pre = 1, post = 1, cheby = 2;
    int native0 = 0;   // level 0 runs on the pair-major P with node kernels: 0 no, 1 all fields, 2 ions only, 3 potential only
    KnpAmgLevel lv[KNP_MAX_AMG_LEVELS];
    int nc = 0;
    double* cinv = nullptr;
    float* cinv_f = nullptr;
    // fused V(1,1) cycle (knp_pc_setup decides): Pt = P Dinv of level 0, fp64 or fp32, 4 fields per pair; potential part compact
    int fused = 0;
    int l0_fused = 0;   // level 0 in fused form inside the level-by-level cycle (distributed hierarchies)
    int node_nf = 0;    // > 0: fields per node with identical patterns on every level (knp_amg_set_node_fields)
    int blocked = 0;    // the fused cycle runs on the node-blocked copies (knp_pc_setup decides)
    int cfused = 0;     // ... and its intermediate levels as two plain products each (Rt, U)
    // potential hierarchy whose level 0 is the UPLOADED operator (it may couple the two sides of the membrane, which the library's
    // pair-major P cannot): compact CSR of c A Dinv on node-indexed vectors for the down-leg of the fused cycle
    int l0_upload = 0;
    int32_t *at0_rp = nullptr, *at0_ci = nullptr;
    double* at0_v = nullptr;
    float* at0_vf = nullptr;
    int at0_lanes = 8;
    double *pt = nullptr, *pt_phi = nullptr;
    float *pt_f = nullptr, *pt_phi_f = nullptr;
};

struct knp_ctx {
    std::string err;
    hipStream_t stream = nullptr;
    int device = 0;
    KnpHostGraph g;  // host copy kept (pattern export, diagnostics)
    // params
    double dt = 0, F = 1, C_M = 1, psi = 1;
    int n_ions = 3;
    double z[KNP_MAX_IONS] = {1, 1, -1}, Di[KNP_MAX_IONS] = {1, 1, 1}, De[KNP_MAX_IONS] = {1, 1, 1};
    // sizes
    int64_t nnz = 0, n_pairs = 0, n_contrib = 0, n_gp = 0, n_gc = 0;
    int n_dof_owned = 0, n_dof_local = 0;
    // device mesh / graph
    int32_t* d_cells = nullptr;
    uint8_t* d_cell_side = nullptr;
    double* d_coords = nullptr;
    int32_t* d_node_vertex = nullptr;
    uint8_t* d_node_side = nullptr;
    int32_t *d_node_i = nullptr, *d_node_e = nullptr;
    int32_t *d_pair_ptr = nullptr, *d_pair_col = nullptr, *d_pair_row = nullptr;
    double *d_pair_M = nullptr, *d_pair_K = nullptr;
    double2* d_pair_MK = nullptr;   // {M, K} per pair (SpMV with matrix-free time-invariant entries)
    int32_t *d_contrib_ptr = nullptr, *d_contrib_cell = nullptr;
    double* d_contrib_k = nullptr;
    int32_t *d_node_cell_ptr = nullptr, *d_node_cell = nullptr;
    uint8_t* d_contrib_slot = nullptr;
    int64_t* d_tc_meta = nullptr;   // transposed contribution lists (k_assemble_nodes_tr): per node base | trips << 48 | self index << 56
    double* d_tc_k = nullptr;
    uint8_t* d_tc_slot = nullptr;
    int64_t n_tc = 0;
    int asm_stage = 0;    // cells per node the staged assembly reserves LDS for (0: gather per contribution)
    // cell means computed inside the assembly's LDS staging (no k_cell_means pass): the node's neighbours' concentrations are read as
    // 32-byte records from a node-indexed copy (d_knod, k_nodal_conc) and averaged per cell through d_ncv, the local neighbour
    // index of every vertex of every cell of the node's list (cell vertex order: the same sums, bit for bit, as k_cell_means)
    int asm_dmax = 0;     // > 0: fused form, LDS for this many neighbours per node
    int64_t n_node_cells = 0;   // total length of the per-node cell lists
    uint8_t* d_ncv = nullptr;
    double* d_knod = nullptr;
    int32_t* d_fv = nullptr;
    double* d_fmeas = nullptr;
    int32_t* d_gamma_prog = nullptr;
    double *d_qp = nullptr, *d_qw = nullptr;
    int32_t *d_gv_vertex = nullptr, *d_gv_node_i = nullptr, *d_gv_node_e = nullptr, *d_node_gv = nullptr;
    int32_t *d_gptr = nullptr, *d_gcol = nullptr, *d_grow = nullptr, *d_gq_i = nullptr, *d_gq_e = nullptr,
            *d_gdiag = nullptr;
    int32_t *d_gcptr = nullptr, *d_gc_facet = nullptr, *d_gc_lab = nullptr;
    // A, pair-major: per same-side node pair p = (n, nb) the 10 entries of the 4x4 block, split into
    //   a_t[4p + {0,1,2}] = A[(n,j),(nb,phi)]   a_t[4p+3] = A[(n,phi),(nb,phi)]        (depend on the previous solution)
    //   a_c[6p + j]       = A[(n,j),(nb,j)]     a_c[6p+3+j] = A[(n,phi),(nb,j)]        (time invariant)
    // and per membrane vertex pair s the coupling to the other side's potential, a_x[8s + 4*side + f] = A[(n_side,f),(cross,phi)]
    // P pair-major (p_vals[4*pair + field])
    double *d_at = nullptr, *d_ac = nullptr, *d_ax = nullptr;
    int32_t *d_gx_i = nullptr, *d_gx_e = nullptr;
    double* d_p_vals = nullptr;
    double* d_px = nullptr;          // [n_gp] coupled-potential form of P: the phi_i-phi_e cross entry of every membrane vertex pair
    int pc_coupled_phi = 0;
    bool have_A = false, have_P = false, have_cc = false;
    // work arrays
    double* d_cbar = nullptr;   // [3*n_c]
    double* d_fmat = nullptr;   // [n_g][npk][6]: facet-major 48-byte records
    double* d_fvec = nullptr;   // [7*dim*n_g]
    // programs
    std::vector<KnpProgram> progs;
    int32_t** d_prog_code = nullptr;   // device table of pointers
    double** d_prog_consts = nullptr;
    int32_t* d_prog_len = nullptr;
    int32_t* d_prog_nconsts = nullptr;
    int prog_regs = 0, prog_len_cap = 0, prog_consts_cap = 0;   // maxima over the programs (LDS sizing of k_gamma_facets)
    size_t gamma_lds_set = 0;
    // run-time compiled membrane kernel (knp_jit.cpp); null: the interpreter runs
    void* jit_module = nullptr;
    void* jit_fn[3] = {nullptr, nullptr, nullptr};   // [0] 2D, [1] 3D right-hand-side facet kernel, [2] 3D with 4 lanes x 9 points per facet
    std::string jit_msg;
    bool progs_dirty = true;
    int max_prog = -1;
    // sources
    const double* src_i[KNP_MAX_IONS] = {nullptr, nullptr, nullptr};
    const double* src_e[KNP_MAX_IONS] = {nullptr, nullptr, nullptr};
    bool have_sources = false;
    // preconditioner
    int pc_kind = KNP_PC_NONE;
    double* d_vbj = nullptr;  // [n_nodes_owned*16] compact vertex blocks
    KnpAmgHier hier[KNP_MAX_HIER];   // 0: all fields (block-Jacobi form) or ion fields; 1: potential
    int amg_fp32 = 0;                // store the preconditioner's operators in fp32 (vectors and A stay fp64)
    float* d_p_vals_f = nullptr;     // fp32 shadow of the pair-major P
    double* d_ML = nullptr;          // [n_nodes_owned] lumped mass of each node
    double* d_cc = nullptr;          // [n_nodes_owned] diagonal Schur term psi / (sum_j z_j^2 k_j) / ML
    double *d_t2 = nullptr, *d_w2 = nullptr;  // work vectors of the block-triangular preconditioner
    // null space
    int ns_on = 0;
    int spmv_group = 8;   // lanes per node of the node-structured SpMV (0 = generic CSR kernel)
    int pc_group = 8;     // lanes per node of the level-0 preconditioner kernels (k_pnode, k_phi_rhs)
    int asm_group = 8;    // lanes per node of the volume assembly (one lane per node pair)
    // GMRES workspace
    int gm_restart = 0;
    double* d_V = nullptr;       // [(restart+1)*n_dof_local]
    double *d_w = nullptr, *d_t = nullptr;  // [n_dof_local]
    double* d_partial = nullptr; // reduction scratch
    double* d_red = nullptr;     // [64] reduced values
    double* h_red = nullptr;     // pinned host mirror
    double* h_red_dev = nullptr; // device-visible address of h_red (zero-copy read-back)
    // reductions publish straight to pinned host memory, except on the hook path (its all-reduce result is fetched by a copy)
    double* mirror() const { return (allreduce && p2p_red < 0) ? nullptr : h_red_dev; }
    int64_t phi_count_cached = -1;
    int64_t* h_seq = nullptr;        // pinned sequence word published by the last reduction kernel of a read-back
    int64_t* h_seq_dev = nullptr;
    int64_t seq_counter = 0;
    int asm_full = 0;                // KNP_ASM_FULL=1: rewrite the time-invariant blocks at every assembly
    double asm_dt = -1.0;
    // Dirichlet rows
    int n_bc = 0;
    int32_t* d_bc_dofs = nullptr;
    // deflation
    int defl_m = 0;
    int32_t* d_defl_mode = nullptr;
    double* d_defl_einv = nullptr;
    double* d_y = nullptr;       // [restart+1]
    double* d_gm = nullptr;      // GMRES bookkeeping on the device (GmLayout: Hessenberg columns, rotations, g, y, state)
    int gm_cap = 0;
    int64_t n_norm_fallback = 0; // iterations that needed the explicit norm (second reduction)
    int n_red_blocks = 0;
    // comm
    knp_halo_fn halo = nullptr;
    knp_allreduce_fn allreduce = nullptr;
    knp_level_comm_fn level_comm = nullptr;
    void* comm_user = nullptr;
    // native peer-to-peer exchange (knp_p2p.hip); plan indices, -1 = use the hooks above
    KnpP2P* p2p = nullptr;
    int p2p_fine = -1;   // halo of the fine DoF vector
    int p2p_red = -1;    // all-reduce of the reduction slots
    bool hook_allreduce() const { return allreduce && p2p_red < 0; }
    int comm_rc = 0;     // first failure of a level exchange inside a preconditioner application
    // interior / boundary split of the SpMV on A (multi-GPU): nodes without / with a ghost column; the native forward halo runs
    // on stream3 next to the interior rows, the boundary rows follow the join
    int n_int = 0, n_bnd = 0;
    int32_t *d_nodes_int = nullptr, *d_nodes_bnd = nullptr;
    hipStream_t stream3 = nullptr;
    hipEvent_t ev_x = nullptr, ev_halo = nullptr;
    // side stream for ||B b|| of the next solve (knp_gmres_prepare)
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    const double* prep_b = nullptr;
    int prep_fused = 0;   // the side-stream ||B b|| used the one-reduction projected norm (flag in slot 61)
    // Concurrent form (one GPU, fused cycle, null space on): the side-stream cycle works on its OWN vectors, partial sums and
    // reduction slots, so it may still be running while the solve computes its first preconditioned residual on the main
    // stream; the solve only joins it when it needs ||B b||, after its first read-back.
    int prep_conc = 0;
    bool prep_deferred = false;   // concurrent form: the side-stream cycle is enqueued by the solve, behind its own first residual chain
    bool side_ws = false;
    double *d_t2_s = nullptr, *d_w2_s = nullptr, *d_wb = nullptr, *d_partial_s = nullptr;
    // matrix assembly on its own stream (knp_assemble_matrix_async): independent of the right-hand side chain of the same step
    hipStream_t stream_asm = nullptr;
    hipEvent_t ev_fork_asm = nullptr, ev_asm = nullptr;
    bool asm_pending = false;
    // statistics of the solves (knp_get_stats): ||B b|| of the last solve, exchanges and host read-backs since the last reset
    double last_bnorm = 0.0;
    int64_t n_allreduce = 0, n_halo = 0, n_readback = 0;
    // step timers (knp_timer_mark / knp_timer_read): timing events recorded on the main stream, read back in one go
    std::vector<hipEvent_t> tm_events;
    size_t tm_used = 0;
    // profiling
    int prof_on = 0;
    std::vector<hipEvent_t> prof_pool;   // recycled timing events
    struct ProfRec { hipEvent_t a, b; int cls; };
    std::vector<ProfRec> prof_recs;
    double prof_ms[KNP_NPROF] = {0, 0, 0, 0, 0};
    int64_t prof_n[KNP_NPROF] = {0, 0, 0, 0, 0};
};

// Threads of the host-side OpenMP loops: the CPU share of the process, not the machine.  A GPU box shows all logical CPUs of its host
// (256) but grants a job a fraction of them (16 per GPU): 256 threads on a 16-core share spend their time descheduled inside barriers --
// the hierarchy hand-over took 5-11 s instead of 1.  KNP_HOST_THREADS=<n> overrides; else the cgroup CPU quota, else the affinity mask,
// divided by LOCAL_WORLD_SIZE (one process per GPU), at most 32.
int knp_host_threads();
