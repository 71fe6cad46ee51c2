// libknpemi_hip -- HIP kernels (gfx950, wave64, fp64) and the C-ABI for the KNP-EMI
// assemble-and-solve hot path.  Everything here is HBM-bandwidth bound: no MFMA.
//
//   K0  k_cell_means            per-cell means of the previous concentrations (feeds K[c] blocks)
//   K1  k_assemble_nodes        gather-assembly of all volume blocks of A (or P) into the pair-major arrays, one lane
//                               per same-side node pair, atomic-free and deterministic
//   K2  k_gamma_facets          membrane facet quadrature (rational alpha weights, mechanism programs)
//   K2b k_gamma_pairs           gather of facet matrices into the CSR coupling entries
//   K3  k_rhs                   mass * k_prev + membrane vectors -> b
//   K4  k_spmv                  CSR SpMV, sub-wave per row, shuffle reduction
//   K5  k_multi_dot / k_update_scale  classical Gram-Schmidt (fused multi-dot; update + normalisation in one pass)
//   K6  k_scale, k_lincomb
//   K7  k_vbj_extract / k_vbj_apply   per-vertex 4x4 / 8x8 block-Jacobi (Schur form, no stored inverse)
//   K8  k_phi_sum / k_phi_sub   null-space (gauge) projection
//   K9  k_hh_update             Hodgkin-Huxley gating (Rush-Larsen collapsed analytically / forward Euler)
//   K10 k_pack / k_unpack       solution vector <-> nodal fields, phi_m = phi_i - phi_e
//   AMG k_cheby_step, k_dense_matvec  V-cycle pieces (level SpMVs reuse K4)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "knp_internal.hpp"

#define HIPCHK(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                  \
            return KNP_E_HIP;                                                              \
        }                                                                                  \
    } while (0)
#define KCHK(expr)                         \
    do {                                   \
        int rc_ = (expr);                  \
        if (rc_ != KNP_OK) return rc_;     \
    } while (0)
#define CHECK_CTX(ctx) \
    if (!(ctx)) return KNP_E_ARG

static constexpr int NT = 256;          // threads per block everywhere
static constexpr int RED_BLOCKS = 1024; // reduction partial blocks (4 per CU on 256 CUs)
static constexpr int RED_SLOTS = 128;   // reduced-value slots: 0..56 Gram-Schmidt, 58..63 misc, 64..95 deflation sums
static constexpr int DEFL_SLOT0 = 64;
static constexpr int DEFL_MAX = 32;
static constexpr int SIDE_SLOT = 120;  // 120..123: {norm, flag, phi sum, w.w} of the side-stream ||B b|| (concurrent knp_gmres_prepare)

struct DevParams {
    double dt, F, C_M, psi;
    double z[3], Di[3], De[3];
    // derived on the host (make_params) so that they arrive as kernel arguments, i.e. in scalar registers: the facet kernels multiply by
    // these at every quadrature point; computed per thread they would occupy 26 vector registers
    double dz2i[3], dz2e[3];   // D_i^k z_k^2, D_e^k z_k^2
    double rFz[3], cmFz[3];    // 1 / (F z_k), C_M / (F z_k)
    double rF;                 // 1 / F
};
struct FieldPtrs {
    const double* ki[3];
    const double* ke[3];
    const double* phim;
    const double* aux[KNP_MAX_AUX];
};

static void side_discard(knp_ctx* ctx);
static DevParams make_params(const knp_ctx* ctx);
static int64_t phi_block_nnz(const knp_ctx* ctx);
static int join_asm(knp_ctx* ctx);
static void free_hier(KnpAmgHier& H);
static inline int nblocks(int64_t n, int per = NT) { return (int)std::max<int64_t>(1, (n + per - 1) / per); }

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the block; result valid in thread 0
__device__ __forceinline__ double block_sum(double v, double* sm) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) sm[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x < (NT >> 6)) r = sm[threadIdx.x];
    if (w == 0) r = wave_sum(r);
    return r;
}

// The six time-invariant entries of a same-side node pair, (k_j,k_j) = M + dt D_j K and (phi,k_j) = dt z_j D_j K (SURVEY 3.2:
// KNPEMIx_problem.py:586-591), from the pair's mass and stiffness constants.  ONE definition for the assembly kernels (which store
// them in a_c for exports, vertex-block Jacobi and Dirichlet rows) and for the SpMV, which recomputes them from 16 bytes instead of
// reading 48: the explicit fused multiply-adds make both produce the same bits.
struct AcCoef { double a0, a1, a2, g0, g1, g2; };
__host__ __device__ __forceinline__ AcCoef ac_coef(const DevParams& P, int side) {
    const double D0 = side ? P.De[0] : P.Di[0], D1 = side ? P.De[1] : P.Di[1], D2 = side ? P.De[2] : P.Di[2];
    return AcCoef{P.dt * D0, P.dt * D1, P.dt * D2, P.dt * P.z[0] * D0, P.dt * P.z[1] * D1, P.dt * P.z[2] * D2};
}
__device__ __forceinline__ void ac_entries(const AcCoef& C, double M, double K, double2& c0, double2& c1, double2& c2) {
    c0 = make_double2(__fma_rn(C.a0, K, M), __fma_rn(C.a1, K, M));
    c1 = make_double2(__fma_rn(C.a2, K, M), C.g0 * K);
    c2 = make_double2(C.g1 * K, C.g2 * K);
}

// Part of a CSR row dot product, L lanes per row, four independent gathers in flight per lane.  These rows are short (3-40
// entries): with one entry per lane per trip the dependent chain rowptr -> (col, val) -> x[col] leaves ~8 B per lane in flight
// and the kernel runs at HBM *latency*; issuing four predicated trips at once keeps the memory system busy instead
// (clamped indices re-read the row's last entry, the products of the surplus trips are dropped).
template <int L, typename VT>
__device__ __forceinline__ double row_dot4(int k0, int e, int lane, const int32_t* __restrict__ ci, const VT* __restrict__ v,
                                           const double* __restrict__ x) {
    double s = 0.0;
    for (int k = k0 + lane; k < e; k += 4 * L) {
        const int kb = min(k + L, e - 1), kc = min(k + 2 * L, e - 1), kd = min(k + 3 * L, e - 1);
        const int ca = ci[k], cb = ci[kb], cc = ci[kc], cd = ci[kd];
        const double va = (double)v[k], vb = (double)v[kb], vc = (double)v[kc], vd = (double)v[kd];
        const double xa = x[ca], xb = x[cb], xc = x[cc], xd = x[cd];
        s += va * xa;
        if (k + L < e) s += vb * xb;
        if (k + 2 * L < e) s += vc * xc;
        if (k + 3 * L < e) s += vd * xd;
    }
    return s;
}

// ------------------------------------------------------------------------------------------
// K0: cell means
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT) k_cell_means(int n_c, int nv1, const int32_t* __restrict__ cells,
                                                   const uint8_t* __restrict__ side, FieldPtrs f,
                                                   double* __restrict__ cbar) {
    int c = blockIdx.x * NT + threadIdx.x;
    if (c >= n_c) return;
    const int s = side[c];
    const double inv = 1.0 / nv1;
    double m[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double* k = s ? f.ke[j] : f.ki[j];
        double acc = 0.0;
        for (int a = 0; a < nv1; ++a) acc += k[cells[(size_t)c * nv1 + a]];
        m[j] = acc * inv;
    }
    // 32-B record per cell, written whole (two 16-B stores): one gather per contribution in K1
    double2* out = reinterpret_cast<double2*>(cbar + (size_t)4 * c);
    out[0] = make_double2(m[0], m[1]);
    out[1] = make_double2(m[2], 0.0);
}

// ------------------------------------------------------------------------------------------
// K1: volume blocks.  One G-lane group per owned node, one lane per node pair (n, nb): the lane sums the pair's cell
// contributions K_ab(T) * cbar_T (gather, atomic-free, deterministic) and writes the pair's entries as whole 16-byte pieces
// of the pair-major arrays -- consecutive lanes write consecutive addresses.  The self pair has no contribution list:
// sum_b K_ab(T) = 0 on every simplex, so its weighted stiffness is minus the sum over the node's other pairs (a shuffle
// reduction over the group) -- a quarter of the gathers and no 24-contribution straggler lane per node.
// ------------------------------------------------------------------------------------------
template <bool PRECOND, bool TD_ONLY, int G>   // TD_ONLY: write only the entries that depend on the previous solution
__global__ void __launch_bounds__(NT)
k_assemble_nodes(int n_nodes, DevParams P, const int32_t* __restrict__ pair_ptr, const int32_t* __restrict__ pair_col,
                 const uint8_t* __restrict__ node_side, const double* __restrict__ pair_M, const double* __restrict__ pair_K,
                 const int32_t* __restrict__ contrib_ptr, const int32_t* __restrict__ contrib_cell,
                 const double* __restrict__ contrib_k, const double* __restrict__ cbar,
                 double* __restrict__ at /* P: p_vals */, double* __restrict__ ac) {
    const int node_raw = (blockIdx.x * NT + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    const bool live = node_raw < n_nodes;
    const int node = live ? node_raw : n_nodes - 1;    // surplus groups shadow the last node (uniform shuffles), they do not write
    const int p0 = pair_ptr[node];
    const int deg = pair_ptr[node + 1] - p0;
    const int side = node_side[node];
    const double D0 = side ? P.De[0] : P.Di[0], D1 = side ? P.De[1] : P.Di[1], D2 = side ? P.De[2] : P.Di[2];
    const double f0 = P.dt * D0 * P.z[0] / P.psi, f1 = P.dt * D1 * P.z[1] / P.psi, f2 = P.dt * D2 * P.z[2] / P.psi;
    auto emit = [&](int p, double S0, double S1, double S2) {
        const double phiphi = f0 * P.z[0] * S0 + f1 * P.z[1] * S1 + f2 * P.z[2] * S2;
        if (!PRECOND) {
            *reinterpret_cast<double2*>(at + 4 * (size_t)p) = make_double2(f0 * S0, f1 * S1);
            *reinterpret_cast<double2*>(at + 4 * (size_t)p + 2) = make_double2(f2 * S2, phiphi);
            if (!TD_ONLY) {
                const double M = pair_M[p], K = pair_K[p];
                double2 c0, c1, c2;
                ac_entries(ac_coef(P, side), M, K, c0, c1, c2);
                *reinterpret_cast<double2*>(ac + 6 * (size_t)p) = c0;          // (k,k)
                *reinterpret_cast<double2*>(ac + 6 * (size_t)p + 2) = c1;      // (k,k) | (phi,k)
                *reinterpret_cast<double2*>(ac + 6 * (size_t)p + 4) = c2;      // (phi,k)
            }
        } else {
            const double M = pair_M[p], K = pair_K[p];
            double2 c0, c1, c2;
            ac_entries(ac_coef(P, side), M, K, c0, c1, c2);
            *reinterpret_cast<double2*>(at + 4 * (size_t)p) = c0;
            *reinterpret_cast<double2*>(at + 4 * (size_t)p + 2) = make_double2(c1.x, phiphi);
        }
    };
    double T0 = 0.0, T1 = 0.0, T2 = 0.0;   // this lane's share of the sum over the node's off-diagonal pairs
    int self_p = -1;
    for (int q = lane; q < deg; q += G) {
        const int p = p0 + q;
        if (pair_col[p] == node) { self_p = p; continue; }
        double S0 = 0.0, S1 = 0.0, S2 = 0.0;
        const int c1 = contrib_ptr[p + 1];
        for (int c = contrib_ptr[p]; c < c1; c += 2) {   // two cells in flight (an edge has 2 cells in 2D, ~5 in 3D)
            const int cb = min(c + 1, c1 - 1);
            const int cell = contrib_cell[c], cell2 = contrib_cell[cb];
            const double k = contrib_k[c], k2 = contrib_k[cb];
            const double2 c01 = *reinterpret_cast<const double2*>(cbar + (size_t)4 * cell);
            const double c2 = cbar[(size_t)4 * cell + 2];
            const double2 d01 = *reinterpret_cast<const double2*>(cbar + (size_t)4 * cell2);
            const double d2 = cbar[(size_t)4 * cell2 + 2];
            S0 += k * c01.x;
            S1 += k * c01.y;
            S2 += k * c2;
            if (c + 1 < c1) {
                S0 += k2 * d01.x;
                S1 += k2 * d01.y;
                S2 += k2 * d2;
            }
        }
        if (live) emit(p, S0, S1, S2);
        T0 += S0; T1 += S1; T2 += S2;
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) {
        T0 += __shfl_xor(T0, o, G);
        T1 += __shfl_xor(T1, o, G);
        T2 += __shfl_xor(T2, o, G);
    }
    if (live && self_p >= 0) emit(self_p, -T0, -T1, -T2);
}
// The same with the cell means of the node's cells staged in LDS: a node's ~24 (3D) same-side cells are each gathered ONCE by
// the group (24-byte records from the cell-mean array) instead of once per contribution (~72 per node: every cell feeds the
// three off-diagonal pairs of the node it belongs to); a contribution is then 9 streamed bytes (K_ab(T), 1-byte slot) and three
// LDS reads.  `cmax` = LDS slots reserved per group (>= the largest cell count of any owned node).
// node-indexed copy of the concentrations of the node's own side: {k0, k1, k2, -} per node, one 32-byte record per gather
__global__ void __launch_bounds__(NT) k_nodal_conc(int n_nodes, const int32_t* __restrict__ node_vertex, const uint8_t* __restrict__ node_side,
                                                   FieldPtrs f, double* __restrict__ knod) {
    const int n = blockIdx.x * NT + threadIdx.x;
    if (n >= n_nodes) return;
    const int v = node_vertex[n];
    const int s = node_side[n];
    const double a = (s ? f.ke[0] : f.ki[0])[v], b = (s ? f.ke[1] : f.ki[1])[v], c = (s ? f.ke[2] : f.ki[2])[v];
    *reinterpret_cast<double2*>(knod + 4 * (size_t)n) = make_double2(a, b);
    *reinterpret_cast<double2*>(knod + 4 * (size_t)n + 2) = make_double2(c, 0.0);
}
// Cell means of the node's same-side cells into LDS (`mine`, 3 doubles per cell of the node's list).  FUSED: from the neighbours'
// nodal values (staged in `nbv` first: one 32-byte read per neighbour instead of a gathered 32-byte record per CELL plus the separate
// k_cell_means pass that produced it -- 15 neighbours against 24 cells per node on the tetrahedral cubes), summed in the cell's
// vertex order exactly like k_cell_means; otherwise from the precomputed records `cbar`.
template <int G, bool FUSED>
__device__ __forceinline__ void stage_cell_means(double* mine, double* nbv, int lane, int node, int p0, int deg, const int32_t* __restrict__ pair_col,
                                                 const int32_t* __restrict__ node_cell_ptr, const int32_t* __restrict__ node_cell,
                                                 const double* __restrict__ cbar, const uint8_t* __restrict__ ncv, const double* __restrict__ knod, int nv1) {
    const int c0 = node_cell_ptr[node], c1 = node_cell_ptr[node + 1];
    if (FUSED) {
        for (int q = lane; q < deg; q += G) {
            const size_t nb = (size_t)pair_col[p0 + q];
            const double2 a = *reinterpret_cast<const double2*>(knod + 4 * nb);
            const double c = knod[4 * nb + 2];
            double* d = nbv + 3 * q;
            d[0] = a.x; d[1] = a.y; d[2] = c;
        }
        __syncthreads();
        const double inv = 1.0 / nv1;
        for (int i = c0 + lane; i < c1; i += G) {
            const uint8_t* __restrict__ lv = ncv + (size_t)nv1 * i;
            double m0 = 0.0, m1 = 0.0, m2 = 0.0;
            for (int a = 0; a < nv1; ++a) {
                const double* u = nbv + 3 * (int)lv[a];
                m0 += u[0]; m1 += u[1]; m2 += u[2];
            }
            double* d = mine + 3 * (i - c0);
            d[0] = m0 * inv; d[1] = m1 * inv; d[2] = m2 * inv;
        }
    } else {
        for (int i = c0 + lane; i < c1; i += G) {
            const int cell = node_cell[i];
            const double2 c01 = *reinterpret_cast<const double2*>(cbar + (size_t)4 * cell);
            const double c2 = cbar[(size_t)4 * cell + 2];
            double* d = mine + 3 * (i - c0);
            d[0] = c01.x; d[1] = c01.y; d[2] = c2;
        }
    }
    __syncthreads();
}

template <bool PRECOND, bool TD_ONLY, int G, bool FUSED>
__global__ void __launch_bounds__(NT)
k_assemble_nodes_staged(int n_nodes, DevParams P, int cmax, const int32_t* __restrict__ pair_ptr, const int32_t* __restrict__ pair_col,
                        const uint8_t* __restrict__ node_side, const double* __restrict__ pair_M, const double* __restrict__ pair_K,
                        const int32_t* __restrict__ contrib_ptr, const uint8_t* __restrict__ contrib_slot,
                        const double* __restrict__ contrib_k, const int32_t* __restrict__ node_cell_ptr,
                        const int32_t* __restrict__ node_cell, const double* __restrict__ cbar,
                        double* __restrict__ at, double* __restrict__ ac,
                        const uint8_t* __restrict__ ncv, const double* __restrict__ knod, int dmax, int nv1) {
    extern __shared__ double lds[];   // [NT / G][cmax + dmax][3]
    const int node_raw = (blockIdx.x * NT + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    const bool live = node_raw < n_nodes;
    const int node = live ? node_raw : n_nodes - 1;
    double* mine = lds + (size_t)(threadIdx.x / G) * (cmax + dmax) * 3;
    const int p0 = pair_ptr[node];
    const int deg = pair_ptr[node + 1] - p0;
    stage_cell_means<G, FUSED>(mine, mine + 3 * cmax, lane, node, p0, deg, pair_col, node_cell_ptr, node_cell, cbar, ncv, knod, nv1);
    const int side = node_side[node];
    const double D0 = side ? P.De[0] : P.Di[0], D1 = side ? P.De[1] : P.Di[1], D2 = side ? P.De[2] : P.Di[2];
    const double f0 = P.dt * D0 * P.z[0] / P.psi, f1 = P.dt * D1 * P.z[1] / P.psi, f2 = P.dt * D2 * P.z[2] / P.psi;
    auto emit = [&](int p, double S0, double S1, double S2) {
        const double phiphi = f0 * P.z[0] * S0 + f1 * P.z[1] * S1 + f2 * P.z[2] * S2;
        if (!PRECOND) {
            *reinterpret_cast<double2*>(at + 4 * (size_t)p) = make_double2(f0 * S0, f1 * S1);
            *reinterpret_cast<double2*>(at + 4 * (size_t)p + 2) = make_double2(f2 * S2, phiphi);
            if (!TD_ONLY) {
                const double M = pair_M[p], K = pair_K[p];
                double2 c0, c1, c2;
                ac_entries(ac_coef(P, side), M, K, c0, c1, c2);
                *reinterpret_cast<double2*>(ac + 6 * (size_t)p) = c0;          // (k,k)
                *reinterpret_cast<double2*>(ac + 6 * (size_t)p + 2) = c1;      // (k,k) | (phi,k)
                *reinterpret_cast<double2*>(ac + 6 * (size_t)p + 4) = c2;      // (phi,k)
            }
        } else {
            const double M = pair_M[p], K = pair_K[p];
            double2 c0, c1, c2;
            ac_entries(ac_coef(P, side), M, K, c0, c1, c2);
            *reinterpret_cast<double2*>(at + 4 * (size_t)p) = c0;
            *reinterpret_cast<double2*>(at + 4 * (size_t)p + 2) = make_double2(c1.x, phiphi);
        }
    };
    double T0 = 0.0, T1 = 0.0, T2 = 0.0;
    int self_p = -1;
    for (int q = lane; q < deg; q += G) {
        const int p = p0 + q;
        if (pair_col[p] == node) { self_p = p; continue; }
        double S0 = 0.0, S1 = 0.0, S2 = 0.0;
        const int c1 = contrib_ptr[p + 1];
        for (int c = contrib_ptr[p]; c < c1; c += 2) {
            const int cb = min(c + 1, c1 - 1);
            const double k = contrib_k[c], k2 = contrib_k[cb];
            const double* u = mine + 3 * (int)contrib_slot[c];
            const double* w = mine + 3 * (int)contrib_slot[cb];
            S0 += k * u[0];
            S1 += k * u[1];
            S2 += k * u[2];
            if (c + 1 < c1) {
                S0 += k2 * w[0];
                S1 += k2 * w[1];
                S2 += k2 * w[2];
            }
        }
        if (live) emit(p, S0, S1, S2);
        T0 += S0; T1 += S1; T2 += S2;
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) {
        T0 += __shfl_xor(T0, o, G);
        T1 += __shfl_xor(T1, o, G);
        T2 += __shfl_xor(T2, o, G);
    }
    if (live && self_p >= 0) emit(self_p, -T0, -T1, -T2);
}

// The staged kernel with the node's contributions stored TRANSPOSED: contribution t of the node's off-diagonal pair j sits at
// base(node) + t * (number of off-diagonal pairs) + j, lists padded with zeros to the longest one of the node (an even number).
// Consecutive lanes then stream consecutive K_ab(T) values and slots (one 8-byte and one 1-byte load per lane and trip, each a
// contiguous ~112-byte piece per node) where the pair-major lists put ~40 bytes between neighbouring lanes and made every load
// walk ~20 cache lines per wave.  `meta[node]` = base | trips << 48 | (index of the self pair among the node's pairs, 255: none) << 56.
template <bool PRECOND, bool TD_ONLY, int G, bool FUSED>
__global__ void __launch_bounds__(NT)
k_assemble_nodes_tr(int n_nodes, DevParams P, int cmax, const int32_t* __restrict__ pair_ptr, const uint8_t* __restrict__ node_side,
                    const double* __restrict__ pair_M, const double* __restrict__ pair_K, const int64_t* __restrict__ meta,
                    const uint8_t* __restrict__ tslot, const double* __restrict__ tk, const int32_t* __restrict__ node_cell_ptr,
                    const int32_t* __restrict__ node_cell, const double* __restrict__ cbar, double* __restrict__ at, double* __restrict__ ac,
                    const int32_t* __restrict__ pair_col, const uint8_t* __restrict__ ncv, const double* __restrict__ knod, int dmax, int nv1) {
    extern __shared__ double lds[];   // [NT / G][cmax + dmax][3]
    const int node_raw = (blockIdx.x * NT + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    const bool live = node_raw < n_nodes;
    const int node = live ? node_raw : n_nodes - 1;
    double* mine = lds + (size_t)(threadIdx.x / G) * (cmax + dmax) * 3;
    const int64_t mt = meta[node];
    const int p0 = pair_ptr[node];
    const int deg = pair_ptr[node + 1] - p0;
    stage_cell_means<G, FUSED>(mine, mine + 3 * cmax, lane, node, p0, deg, pair_col, node_cell_ptr, node_cell, cbar, ncv, knod, nv1);
    const int64_t base = mt & 0xffffffffffffLL;
    const int trips = (int)((mt >> 48) & 0xff);
    const int selfq = (int)((mt >> 56) & 0xff);
    const int dod = deg - (selfq < 255 ? 1 : 0);
    const int side = node_side[node];
    const double D0 = side ? P.De[0] : P.Di[0], D1 = side ? P.De[1] : P.Di[1], D2 = side ? P.De[2] : P.Di[2];
    const double f0 = P.dt * D0 * P.z[0] / P.psi, f1 = P.dt * D1 * P.z[1] / P.psi, f2 = P.dt * D2 * P.z[2] / P.psi;
    auto emit = [&](int p, double S0, double S1, double S2) {
        const double phiphi = f0 * P.z[0] * S0 + f1 * P.z[1] * S1 + f2 * P.z[2] * S2;
        if (!PRECOND) {
            *reinterpret_cast<double2*>(at + 4 * (size_t)p) = make_double2(f0 * S0, f1 * S1);
            *reinterpret_cast<double2*>(at + 4 * (size_t)p + 2) = make_double2(f2 * S2, phiphi);
            if (!TD_ONLY) {
                const double M = pair_M[p], K = pair_K[p];
                double2 c0, c1, c2;
                ac_entries(ac_coef(P, side), M, K, c0, c1, c2);
                *reinterpret_cast<double2*>(ac + 6 * (size_t)p) = c0;          // (k,k)
                *reinterpret_cast<double2*>(ac + 6 * (size_t)p + 2) = c1;      // (k,k) | (phi,k)
                *reinterpret_cast<double2*>(ac + 6 * (size_t)p + 4) = c2;      // (phi,k)
            }
        } else {
            const double M = pair_M[p], K = pair_K[p];
            double2 c0, c1, c2;
            ac_entries(ac_coef(P, side), M, K, c0, c1, c2);
            *reinterpret_cast<double2*>(at + 4 * (size_t)p) = c0;
            *reinterpret_cast<double2*>(at + 4 * (size_t)p + 2) = make_double2(c1.x, phiphi);
        }
    };
    double T0 = 0.0, T1 = 0.0, T2 = 0.0;
    for (int q = lane; q < deg; q += G) {
        if (q == selfq) continue;
        const int j = q - (q > selfq ? 1 : 0);
        const double* __restrict__ kk = tk + base + j;
        const uint8_t* __restrict__ ss = tslot + base + j;
        double S0 = 0.0, S1 = 0.0, S2 = 0.0;
        // two contributions in flight per lane (four or six, or more resident waves, measured no faster on MI355X);
        // trips is even; padded entries are K = 0 on slot 0
        for (int t = 0; t < trips; t += 2) {
            const double k = kk[(size_t)t * dod], k2 = kk[(size_t)(t + 1) * dod];
            const double* u = mine + 3 * (int)ss[(size_t)t * dod];
            const double* w = mine + 3 * (int)ss[(size_t)(t + 1) * dod];
            S0 += k * u[0];
            S1 += k * u[1];
            S2 += k * u[2];
            S0 += k2 * w[0];
            S1 += k2 * w[1];
            S2 += k2 * w[2];
        }
        if (live) emit(p0 + q, S0, S1, S2);
        T0 += S0; T1 += S1; T2 += S2;
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) {
        T0 += __shfl_xor(T0, o, G);
        T1 += __shfl_xor(T1, o, G);
        T2 += __shfl_xor(T2, o, G);
    }
    if (live && selfq < 255 && (selfq & (G - 1)) == lane) emit(p0 + selfq, -T0, -T1, -T2);
}

// what the volume assembly reads the previous concentrations from: the node-indexed copy (fused cell means) or the per-cell means
static void launch_cell_means(knp_ctx* ctx, const FieldPtrs& f) {
    const KnpHostGraph& g = ctx->g;
    if (ctx->asm_dmax > 0)
        hipLaunchKernelGGL(k_nodal_conc, dim3(nblocks(g.n_nodes)), dim3(NT), 0, ctx->stream, g.n_nodes, ctx->d_node_vertex, ctx->d_node_side, f, ctx->d_knod);
    else
        hipLaunchKernelGGL(k_cell_means, dim3(nblocks(g.n_c)), dim3(NT), 0, ctx->stream, g.n_c, g.nv1, ctx->d_cells, ctx->d_cell_side, f, ctx->d_cbar);
}
template <bool PRECOND, bool TD_ONLY>
static void launch_assemble_nodes(knp_ctx* ctx, const DevParams& P, double* at, double* ac) {
    const KnpHostGraph& g = ctx->g;
    const int n = g.n_nodes_owned;
    if (n <= 0) return;
    if (ctx->asm_stage > 0 && ctx->d_tc_meta) {   // staged + transposed contribution lists (default)
        const int cmax = ctx->asm_stage;
        const int dmax = ctx->asm_dmax, nv1 = g.nv1;
#define KNP_ASMT2(GG, FF) hipLaunchKernelGGL((k_assemble_nodes_tr<PRECOND, TD_ONLY, GG, FF>), dim3(nblocks((int64_t)n * GG)), dim3(NT),                  \
                                        (size_t)(NT / GG) * (cmax + dmax) * 3 * sizeof(double), ctx->stream, n, P, cmax, ctx->d_pair_ptr, ctx->d_node_side, \
                                        ctx->d_pair_M, ctx->d_pair_K, ctx->d_tc_meta, ctx->d_tc_slot, ctx->d_tc_k, ctx->d_node_cell_ptr,              \
                                        ctx->d_node_cell, ctx->d_cbar, at, ac, ctx->d_pair_col, ctx->d_ncv, ctx->d_knod, dmax, nv1)
#define KNP_ASMT(GG) do { if (dmax > 0) KNP_ASMT2(GG, true); else KNP_ASMT2(GG, false); } while (0)
        switch (ctx->asm_group) {
            case 4: KNP_ASMT(4); break;
            case 8: KNP_ASMT(8); break;
            case 16: KNP_ASMT(16); break;
            default: KNP_ASMT(32); break;
        }
#undef KNP_ASMT
#undef KNP_ASMT2
        return;
    }
    if (ctx->asm_stage > 0) {   // cell means staged in LDS (default)
        const int cmax = ctx->asm_stage;
        const int dmax = ctx->asm_dmax, nv1 = g.nv1;
#define KNP_ASMS2(GG, FF) hipLaunchKernelGGL((k_assemble_nodes_staged<PRECOND, TD_ONLY, GG, FF>), dim3(nblocks((int64_t)n * GG)), dim3(NT),              \
                                        (size_t)(NT / GG) * (cmax + dmax) * 3 * sizeof(double), ctx->stream, n, P, cmax, ctx->d_pair_ptr, ctx->d_pair_col,  \
                                        ctx->d_node_side, ctx->d_pair_M, ctx->d_pair_K, ctx->d_contrib_ptr, ctx->d_contrib_slot, ctx->d_contrib_k, \
                                        ctx->d_node_cell_ptr, ctx->d_node_cell, ctx->d_cbar, at, ac, ctx->d_ncv, ctx->d_knod, dmax, nv1)
#define KNP_ASMS(GG) do { if (dmax > 0) KNP_ASMS2(GG, true); else KNP_ASMS2(GG, false); } while (0)
        switch (ctx->asm_group) {
            case 4: KNP_ASMS(4); break;
            case 8: KNP_ASMS(8); break;
            case 16: KNP_ASMS(16); break;
            default: KNP_ASMS(32); break;
        }
#undef KNP_ASMS
#undef KNP_ASMS2
        return;
    }
#define KNP_ASM(GG) hipLaunchKernelGGL((k_assemble_nodes<PRECOND, TD_ONLY, GG>), dim3(nblocks((int64_t)n * GG)), dim3(NT), 0, ctx->stream, n, P, ctx->d_pair_ptr, \
                                       ctx->d_pair_col, ctx->d_node_side, ctx->d_pair_M, ctx->d_pair_K, ctx->d_contrib_ptr, ctx->d_contrib_cell,       \
                                       ctx->d_contrib_k, ctx->d_cbar, at, ac)
    switch (ctx->asm_group) {
        case 4: KNP_ASM(4); break;
        case 8: KNP_ASM(8); break;
        case 16: KNP_ASM(16); break;
        default: KNP_ASM(32); break;
    }
#undef KNP_ASM
}

// ------------------------------------------------------------------------------------------
// K2: membrane facet quadrature + the membrane-program interpreter (shared with the run-time compiled variant)
// ------------------------------------------------------------------------------------------
#include "knp_gamma_facets.inc"
// 3D: 4 lanes x 9 points per facet instead of 16 x 3 -- the 36 points of the degree-10 rule exactly, where 16 x 3 evaluates 48
// slots and shuffles over four times the lanes.  Measured on MI355X: 1.5 M facets (tissue surrogate) mechanism currents 4.67 -> 3.36 ms,
// matrix part 1.82 -> 0.70 ms; 55 k facets (cube 136^3) 162 -> 148 / 79 -> 36 us; 12 k facets (cube 64^3) 45.7 -> 44.2 / 24.3 -> 16.9 us.
// KNP_GAMMA_MANY=<least number of facets> moves the switch, 0 keeps the 16 x 3 kernels.
static bool gamma_many(const KnpHostGraph& g) {
    static const int many_min = getenv("KNP_GAMMA_MANY") ? atoi(getenv("KNP_GAMMA_MANY")) : 1;
    return g.dim == 3 && g.n_q == 36 && many_min > 0 && g.n_g >= many_min;
}

// ------------------------------------------------------------------------------------------
// K2b: membrane coupling entries, one thread per membrane vertex pair
// ------------------------------------------------------------------------------------------
template <bool PRECOND>
__global__ void __launch_bounds__(NT)
k_gamma_pairs(int64_t n_gp, int n_g, int dim, DevParams P, const int32_t* __restrict__ grow,
              const int32_t* __restrict__ gptr, const int32_t* __restrict__ gv_node_i,
              const int32_t* __restrict__ gv_node_e, const int32_t* __restrict__ gq_i,
              const int32_t* __restrict__ gq_e, const int32_t* __restrict__ gcptr,
              const int32_t* __restrict__ gc_facet, const int32_t* __restrict__ gc_lab,
              const double* __restrict__ fmeas, const double* __restrict__ fmat,
              const int32_t* __restrict__ pair_ptr, double* __restrict__ at /* P: p_vals */, double* __restrict__ ax,
              double* __restrict__ px = nullptr) {
    int64_t s = (int64_t)blockIdx.x * NT + threadIdx.x;
    if (s >= n_gp) return;
    const int A = grow[s];
    const int ni = gv_node_i[A], ne = gv_node_e[A];
    const int npk = dim * (dim + 1) / 2;
    double m[6] = {0, 0, 0, 0, 0, 0};
    double m0 = 0.0;
    const double mref = 1.0 / (dim * (dim + 1.0));
    for (int c = gcptr[s]; c < gcptr[s + 1]; ++c) {
        const int fct = gc_facet[c];
        int la = gc_lab[c] >> 2, lb = gc_lab[c] & 3;
        m0 += fmeas[fct] * mref * (la == lb ? 2.0 : 1.0);
        if (!PRECOND) {
            if (la > lb) { int t = la; la = lb; lb = t; }
            const int idx = la * dim - la * (la - 1) / 2 + (lb - la);
            const double2* rec = reinterpret_cast<const double2*>(fmat + ((size_t)fct * npk + idx) * 6);   // 48-byte record of the facet's pair
            const double2 r0 = rec[0], r1 = rec[1], r2 = rec[2];
            m[0] += r0.x; m[1] += r0.y; m[2] += r1.x; m[3] += r1.y; m[4] += r2.x; m[5] += r2.y;
        }
    }
    m0 *= P.C_M / P.F;
    const size_t pi = (size_t)pair_ptr[ni] + gq_i[s], pe = (size_t)pair_ptr[ne] + gq_e[s];
    if (!PRECOND) {
        // same-side slots: + M_Gamma[C^k] on (k,phi), + (C_M/F) M_Gamma on (phi,phi); cross slots: the negatives
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            at[4 * pi + j] += m[j];
            at[4 * pe + j] += m[3 + j];
        }
        at[4 * pi + 3] += m0;
        at[4 * pe + 3] += m0;
        *reinterpret_cast<double2*>(ax + 8 * (size_t)s) = make_double2(-m[0], -m[1]);
        *reinterpret_cast<double2*>(ax + 8 * (size_t)s + 2) = make_double2(-m[2], -m0);
        *reinterpret_cast<double2*>(ax + 8 * (size_t)s + 4) = make_double2(-m[3], -m[4]);
        *reinterpret_cast<double2*>(ax + 8 * (size_t)s + 6) = make_double2(-m[5], -m0);
    } else if (px) {
        // coupled-potential form (knp_pc_set_coupled_potential): the potential block of P is the potential block of A at the time of the
        // preconditioner assembly -- + (C_M/F) M_Gamma on both sides and the phi_i-phi_e coupling - (C_M/F) M_Gamma (KNPEMIx_problem.py:637-638)
        at[4 * pi + 3] += m0;
        at[4 * pe + 3] += m0;
        px[s] = -m0;
    } else {
        at[4 * pi + 3] -= m0;   // KNPEMIx_problem.py:737
        at[4 * pe + 3] -= m0;   // KNPEMIx_problem.py:738
    }
}

// ------------------------------------------------------------------------------------------
// K3: right-hand side, one G-lane group per owned node (lanes split the node's pairs and its membrane facets)
// ------------------------------------------------------------------------------------------
template <int G>
__global__ void __launch_bounds__(NT)
k_rhs(int n_nodes_owned, int n_g, int dim, double dt, const int32_t* __restrict__ node_vertex,
      const uint8_t* __restrict__ node_side, const int32_t* __restrict__ pair_ptr,
      const int32_t* __restrict__ pair_col, const double* __restrict__ pair_M, FieldPtrs f, FieldPtrs src,
      int have_src, const int32_t* __restrict__ node_gv, const int32_t* __restrict__ gdiag,
      const int32_t* __restrict__ gcptr, const int32_t* __restrict__ gc_facet,
      const int32_t* __restrict__ gc_lab, const double* __restrict__ fvec, double* __restrict__ b) {
    const int n = (blockIdx.x * NT + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, ap = 0.0;
    if (n < n_nodes_owned) {
        const int side = node_side[n];
        const double* k0 = side ? f.ke[0] : f.ki[0];
        const double* k1 = side ? f.ke[1] : f.ki[1];
        const double* k2 = side ? f.ke[2] : f.ki[2];
        const int pe = pair_ptr[n + 1];
        for (int p = pair_ptr[n] + lane; p < pe; p += G) {
            const int vb = node_vertex[pair_col[p]];
            const double M = pair_M[p];
            a0 += M * k0[vb];
            a1 += M * k1[vb];
            a2 += M * k2[vb];
            if (have_src) {
                const double* s0 = side ? src.ke[0] : src.ki[0];
                const double* s1 = side ? src.ke[1] : src.ki[1];
                const double* s2 = side ? src.ke[2] : src.ki[2];
                if (s0) a0 += dt * M * s0[vb];
                if (s1) a1 += dt * M * s1[vb];
                if (s2) a2 += dt * M * s2[vb];
            }
        }
        const int A = node_gv[n];
        if (A >= 0) {
            const int s = gdiag[A];
            const double sg = side ? 1.0 : -1.0;
            const int off = side ? 3 : 0;
            for (int c = gcptr[s] + lane; c < gcptr[s + 1]; c += G) {
                const int fct = gc_facet[c];
                const int la = gc_lab[c] >> 2;
                a0 += sg * fvec[((size_t)(off + 0) * dim + la) * n_g + fct];
                a1 += sg * fvec[((size_t)(off + 1) * dim + la) * n_g + fct];
                a2 += sg * fvec[((size_t)(off + 2) * dim + la) * n_g + fct];
                ap += sg * fvec[((size_t)6 * dim + la) * n_g + fct];
            }
        }
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) {
        a0 += __shfl_xor(a0, o, G);
        a1 += __shfl_xor(a1, o, G);
        a2 += __shfl_xor(a2, o, G);
        ap += __shfl_xor(ap, o, G);
    }
    if (lane == 0 && n < n_nodes_owned) {
        *reinterpret_cast<double2*>(b + (size_t)4 * n) = make_double2(a0, a1);
        *reinterpret_cast<double2*>(b + (size_t)4 * n + 2) = make_double2(a2, ap);
    }
}

// ------------------------------------------------------------------------------------------
// K4: CSR SpMV, L lanes per row.  mode 0: y = A x ; mode 1: y = b - A x ; mode 2: y += A x
// ------------------------------------------------------------------------------------------
template <int L, int MODE, int TAG, typename VT = double>   // TAG 1 = the system matrix A, 0 = AMG level operators (VT float: fp32-stored)
__global__ void __launch_bounds__(NT)
k_spmv(int n_rows, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci, const VT* __restrict__ v,
       const double* __restrict__ x, const double* __restrict__ b, double* __restrict__ y) {
    const int gid = blockIdx.x * NT + threadIdx.x;
    const int row = gid / L;
    const int lane = threadIdx.x & (L - 1);
    double s = 0.0;
    if (row < n_rows) s = row_dot4<L, VT>(rp[row], rp[row + 1], lane, ci, v, x);
#pragma unroll
    for (int o = L >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, L);
    if (lane == 0 && row < n_rows) {
        if (MODE == 0) y[row] = s;
        else if (MODE == 1) y[row] = b[row] - s;
        else y[row] += s;
    }
}


// ------------------------------------------------------------------------------------------
// K4n: SpMV on the system matrix, one G-lane group per NODE (its 4 rows).  Per node pair the lane reads the pair's 10
// entries as five 16-byte loads from two contiguous streams (a_c: 48 B, a_t: 32 B per pair), the neighbour index (4 B) and
// the neighbour's 4 unknowns (two 16-byte loads): no column index per entry, no row pointer.  Membrane nodes add the
// coupling to the other side's potentials (a_x, 32 B + a 4-byte column per membrane neighbour).
// `nodes` (optional) lists the nodes to process: interior / boundary split of the multi-GPU path.
// ------------------------------------------------------------------------------------------
// MF ("matrix-free constants", the default without Dirichlet rows): the six time-invariant entries of a pair are not read (48 B) but
// recomputed from its mass and stiffness constants (16 B, `mk`) with ac_entries -- bit-identical to the stored a_c, 35 % less
// traffic per pair (52 instead of 84 bytes with the neighbour index).
template <int G, int MODE, bool MF, int U>
__global__ void __launch_bounds__(NT)
k_spmv_node(int n_list, const int32_t* __restrict__ nodes, const int32_t* __restrict__ pair_ptr, const int32_t* __restrict__ pair_col,
            const double* __restrict__ ac, const double2* __restrict__ mk, AcCoef coef_i, AcCoef coef_e,
            const double* __restrict__ at, const int32_t* __restrict__ node_gv,
            const uint8_t* __restrict__ node_side, const int32_t* __restrict__ gptr, const int32_t* __restrict__ gx_i,
            const int32_t* __restrict__ gx_e, const double* __restrict__ ax,
            const double* __restrict__ x, const double* __restrict__ b, double* __restrict__ y) {
    const int i = (blockIdx.x * NT + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    double y0 = 0.0, y1 = 0.0, y2 = 0.0, y3 = 0.0;
    int node = 0;
    if (i < n_list) {
        node = nodes ? nodes[i] : i;
        const int p0 = pair_ptr[node];
        const int p1 = pair_ptr[node + 1];
        const AcCoef C = (MF && node_side[node]) ? coef_e : coef_i;      // pairs join nodes of the same side
        // U pairs in flight per lane (predicated: the surplus trips re-read the node's last pair and are dropped): the gathers of x
        // are latency bound, two independent chains per lane hide more of it than twice the lanes with one chain each
        for (int pb = p0 + lane; pb < p1; pb += U * G) {
            int nb[U];
            double2 xa[U], xb[U], mv[U], t0[U], t1[U], c0[U], c1[U], c2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) nb[u] = pair_col[min(pb + u * G, p1 - 1)];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const size_t p = (size_t)min(pb + u * G, p1 - 1);
                if (MF) {
                    mv[u] = mk[p];
                } else {
                    c0[u] = *reinterpret_cast<const double2*>(ac + 6 * p);                      // kk0 kk1
                    c1[u] = *reinterpret_cast<const double2*>(ac + 6 * p + 2);                  // kk2 phik0
                    c2[u] = *reinterpret_cast<const double2*>(ac + 6 * p + 4);                  // phik1 phik2
                }
                t0[u] = *reinterpret_cast<const double2*>(at + 4 * p);                          // kphi0 kphi1
                t1[u] = *reinterpret_cast<const double2*>(at + 4 * p + 2);                      // kphi2 phiphi
                xa[u] = *reinterpret_cast<const double2*>(x + 4 * (size_t)nb[u]);
                xb[u] = *reinterpret_cast<const double2*>(x + 4 * (size_t)nb[u] + 2);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (u > 0 && pb + u * G >= p1) break;
                if (MF) ac_entries(C, mv[u].x, mv[u].y, c0[u], c1[u], c2[u]);
                y0 += c0[u].x * xa[u].x + t0[u].x * xb[u].y;
                y1 += c0[u].y * xa[u].y + t0[u].y * xb[u].y;
                y2 += c1[u].x * xb[u].x + t1[u].x * xb[u].y;
                y3 += c1[u].y * xa[u].x + c2[u].x * xa[u].y + c2[u].y * xb[u].x + t1[u].y * xb[u].y;
            }
        }
        const int A = node_gv[node];
        if (A >= 0) {
            const int sd = node_side[node];
            const int32_t* __restrict__ gx = sd ? gx_e : gx_i;
            const int s1 = gptr[A + 1];
            for (int s = gptr[A] + lane; s < s1; s += G) {
                const double xv = x[4 * (size_t)gx[s] + 3];
                const double2 v0 = *reinterpret_cast<const double2*>(ax + 8 * (size_t)s + 4 * sd);
                const double2 v1 = *reinterpret_cast<const double2*>(ax + 8 * (size_t)s + 4 * sd + 2);
                y0 += v0.x * xv;
                y1 += v0.y * xv;
                y2 += v1.x * xv;
                y3 += v1.y * xv;
            }
        }
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) {
        y0 += __shfl_xor(y0, o, G);
        y1 += __shfl_xor(y1, o, G);
        y2 += __shfl_xor(y2, o, G);
        y3 += __shfl_xor(y3, o, G);
    }
    if (lane == 0 && i < n_list) {
        double2 o0, o1;
        if (MODE) {
            const double2 b0 = *reinterpret_cast<const double2*>(b + 4 * (size_t)node);
            const double2 b1 = *reinterpret_cast<const double2*>(b + 4 * (size_t)node + 2);
            o0 = make_double2(b0.x - y0, b0.y - y1);
            o1 = make_double2(b1.x - y2, b1.y - y3);
        } else {
            o0 = make_double2(y0, y1);
            o1 = make_double2(y2, y3);
        }
        *reinterpret_cast<double2*>(y + 4 * (size_t)node) = o0;
        *reinterpret_cast<double2*>(y + 4 * (size_t)node + 2) = o1;
    }
}
// ev_a / ev_b (optional): events bound to the kernel's own begin / end (hipExtLaunchKernelGGL), so that their elapsed time
// is the kernel duration a profiler reports, without the gap between an event record and the launch
template <int MODE>
static void launch_spmv_node(knp_ctx* ctx, int n_list, const int32_t* nodes, const double* x, const double* b, double* y,
                             hipEvent_t ev_a = nullptr, hipEvent_t ev_b = nullptr) {
    if (n_list <= 0) return;
    // Dirichlet rows rewrite stored entries of a_c (identity rows): those contexts read the stored values
    static const bool mf_off = getenv("KNP_SPMV_MF") && atoi(getenv("KNP_SPMV_MF")) == 0;
    const bool mf = !mf_off && ctx->n_bc == 0 && ctx->d_pair_MK != nullptr;
    const DevParams P = make_params(ctx);
    const AcCoef ci = ac_coef(P, 0), ce = ac_coef(P, 1);
    static const int unroll = getenv("KNP_SPMV_UNROLL") ? atoi(getenv("KNP_SPMV_UNROLL")) : 2;
#define KNP_SPMV_NODE3(GG, MFF, UU)                                                                                                    \
    hipExtLaunchKernelGGL((k_spmv_node<GG, MODE, MFF, UU>), dim3(nblocks((int64_t)n_list * GG)), dim3(NT), 0, ctx->stream, ev_a, ev_b, 0, n_list, nodes, \
                          ctx->d_pair_ptr, ctx->d_pair_col, ctx->d_ac, ctx->d_pair_MK, ci, ce, ctx->d_at, ctx->d_node_gv, ctx->d_node_side, ctx->d_gptr, ctx->d_gx_i, \
                          ctx->d_gx_e, ctx->d_ax, x, b, y)
#define KNP_SPMV_NODE2(GG, MFF) do { if (unroll >= 2) KNP_SPMV_NODE3(GG, MFF, 2); else KNP_SPMV_NODE3(GG, MFF, 1); } while (0)
#define KNP_SPMV_NODE(GG) do { if (mf) KNP_SPMV_NODE2(GG, true); else KNP_SPMV_NODE2(GG, false); } while (0)
    switch (ctx->spmv_group) {
        case 4: KNP_SPMV_NODE(4); break;
        case 8: KNP_SPMV_NODE(8); break;
        case 16: KNP_SPMV_NODE(16); break;
        default: KNP_SPMV_NODE(32); break;
    }
#undef KNP_SPMV_NODE
#undef KNP_SPMV_NODE2
#undef KNP_SPMV_NODE3
}

// ------------------------------------------------------------------------------------------
// Level 0 of the AMG cycle works on P itself, which has the node-graph structure with 4 decoupled fields:
// one G-lane group per node, 32-byte loads of the pair's 4 entries and of the neighbour's 4 unknowns.
// Columns outside the owned block (ghost nodes) are skipped: the per-GPU block of P.
//   MODE 0: Chebyshev step  d = c1*d + c2*Dinv*(b - P xin), xout = xin + d
//   MODE 1: residual        xout = b - P xin
// ------------------------------------------------------------------------------------------
template <int G, int MODE, int FM, typename VT>   // FM field mask: 0 all four fields, 1 ion fields only, 2 potential only
__global__ void __launch_bounds__(NT)
k_pnode(int n_nodes, int n_col_nodes, const int32_t* __restrict__ pair_ptr, const int32_t* __restrict__ pair_col,
        const VT* __restrict__ pv, const double* __restrict__ dinv, const double* __restrict__ b,
        const double* __restrict__ xin, double c1, double c2, double* __restrict__ d, double* __restrict__ xout) {
    const int node = (blockIdx.x * NT + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (node < n_nodes) {
        const int p0 = pair_ptr[node];
        const int deg = pair_ptr[node + 1] - p0;
        for (int q = lane; q < deg; q += G) {
            const int nb = pair_col[p0 + q];
            if (nb < n_col_nodes) {
                const double2 xb = *reinterpret_cast<const double2*>(xin + 4 * (size_t)nb + 2);
                double a0, a1, c0, c1v;
                if (sizeof(VT) == 4) {
                    const float4 pq = *reinterpret_cast<const float4*>(pv + 4 * (size_t)(p0 + q));
                    a0 = pq.x; a1 = pq.y; c0 = pq.z; c1v = pq.w;
                } else {
                    const double2 a = *reinterpret_cast<const double2*>(pv + 4 * (size_t)(p0 + q));
                    const double2 c = *reinterpret_cast<const double2*>(pv + 4 * (size_t)(p0 + q) + 2);
                    a0 = a.x; a1 = a.y; c0 = c.x; c1v = c.y;
                }
                if (FM != 2) {
                    const double2 xa = *reinterpret_cast<const double2*>(xin + 4 * (size_t)nb);
                    s0 += a0 * xa.x;
                    s1 += a1 * xa.y;
                    s2 += c0 * xb.x;
                }
                if (FM != 1) s3 += c1v * xb.y;
            }
        }
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) {
        s0 += __shfl_xor(s0, o, G);
        s1 += __shfl_xor(s1, o, G);
        s2 += __shfl_xor(s2, o, G);
        s3 += __shfl_xor(s3, o, G);
    }
    if (lane == 0 && node < n_nodes) {
        const size_t r = 4 * (size_t)node;
        const double2 b0 = *reinterpret_cast<const double2*>(b + r);
        const double2 b1 = *reinterpret_cast<const double2*>(b + r + 2);
        if (MODE == 1) {
            *reinterpret_cast<double2*>(xout + r) = make_double2(b0.x - s0, b0.y - s1);
            *reinterpret_cast<double2*>(xout + r + 2) = make_double2(b1.x - s2, b1.y - s3);
        } else {
            const double2 i0 = *reinterpret_cast<const double2*>(dinv + r);
            const double2 i1 = *reinterpret_cast<const double2*>(dinv + r + 2);
            const double2 x0 = *reinterpret_cast<const double2*>(xin + r);
            const double2 x1 = *reinterpret_cast<const double2*>(xin + r + 2);
            double2 d0 = make_double2(0.0, 0.0), d1 = make_double2(0.0, 0.0);
            if (c1 != 0.0) {
                d0 = *reinterpret_cast<const double2*>(d + r);
                d1 = *reinterpret_cast<const double2*>(d + r + 2);
            }
            // masked-out fields keep d = 0 and x unchanged (their Dinv is 0 in a single-field-class hierarchy)
            d0.x = (FM == 2) ? 0.0 : c1 * d0.x + c2 * i0.x * (b0.x - s0);
            d0.y = (FM == 2) ? 0.0 : c1 * d0.y + c2 * i0.y * (b0.y - s1);
            d1.x = (FM == 2) ? 0.0 : c1 * d1.x + c2 * i1.x * (b1.x - s2);
            d1.y = (FM == 1) ? 0.0 : c1 * d1.y + c2 * i1.y * (b1.y - s3);
            *reinterpret_cast<double2*>(d + r) = d0;
            *reinterpret_cast<double2*>(d + r + 2) = d1;
            *reinterpret_cast<double2*>(xout + r) = make_double2(x0.x + d0.x, x0.y + d0.y);
            *reinterpret_cast<double2*>(xout + r + 2) = make_double2(x1.x + d1.x, x1.y + d1.y);
        }
    }
}
template <int MODE, int FM, typename VT>
static void launch_pnode_fm(hipStream_t st, int G, int n_nodes, int ncn, const int32_t* pp, const int32_t* pc, const VT* pv,
                            const double* dinv, const double* b, const double* xin, double c1, double c2, double* d, double* xout) {
    switch (G) {
        case 4: hipLaunchKernelGGL((k_pnode<4, MODE, FM, VT>), dim3(nblocks((int64_t)n_nodes * 4)), dim3(NT), 0, st, n_nodes, ncn, pp, pc, pv, dinv, b, xin, c1, c2, d, xout); break;
        case 8: hipLaunchKernelGGL((k_pnode<8, MODE, FM, VT>), dim3(nblocks((int64_t)n_nodes * 8)), dim3(NT), 0, st, n_nodes, ncn, pp, pc, pv, dinv, b, xin, c1, c2, d, xout); break;
        case 16: hipLaunchKernelGGL((k_pnode<16, MODE, FM, VT>), dim3(nblocks((int64_t)n_nodes * 16)), dim3(NT), 0, st, n_nodes, ncn, pp, pc, pv, dinv, b, xin, c1, c2, d, xout); break;
        default: hipLaunchKernelGGL((k_pnode<32, MODE, FM, VT>), dim3(nblocks((int64_t)n_nodes * 32)), dim3(NT), 0, st, n_nodes, ncn, pp, pc, pv, dinv, b, xin, c1, c2, d, xout); break;
    }
}
template <int MODE, typename VT>
static void launch_pnode_t(hipStream_t st, int fm, int G, int n_nodes, int ncn, const int32_t* pp, const int32_t* pc, const VT* pv,
                           const double* dinv, const double* b, const double* xin, double c1, double c2, double* d, double* xout) {
    if (n_nodes <= 0) return;
    if (fm == 1) launch_pnode_fm<MODE, 1, VT>(st, G, n_nodes, ncn, pp, pc, pv, dinv, b, xin, c1, c2, d, xout);
    else if (fm == 2) launch_pnode_fm<MODE, 2, VT>(st, G, n_nodes, ncn, pp, pc, pv, dinv, b, xin, c1, c2, d, xout);
    else launch_pnode_fm<MODE, 0, VT>(st, G, n_nodes, ncn, pp, pc, pv, dinv, b, xin, c1, c2, d, xout);
}
template <int MODE>
static void launch_pnode(hipStream_t st, int fm, int G, int n_nodes, int ncn, const int32_t* pp, const int32_t* pc, const double* pv,
                         const float* pvf, const double* dinv, const double* b, const double* xin, double c1, double c2, double* d, double* xout) {
    if (pvf) launch_pnode_t<MODE, float>(st, fm, G, n_nodes, ncn, pp, pc, pvf, dinv, b, xin, c1, c2, d, xout);
    else launch_pnode_t<MODE, double>(st, fm, G, n_nodes, ncn, pp, pc, pv, dinv, b, xin, c1, c2, d, xout);
}

__global__ void __launch_bounds__(NT) k_to_float(int64_t n, const double* __restrict__ in, float* __restrict__ out) {
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT) out[e] = (float)in[e];
}

// ------------------------------------------------------------------------------------------
// Block-triangular preconditioner pieces (ion blocks first, then the potential with a Cahouet-Chabard
// approximation of its Schur complement):
//   k_phi_rhs:   t = r ; t_phi -= A_{phi,k} z_k          (the phi rows of A, node-structured)
//   k_schur_fin: z_phi = w_phi + cc * t_phi
//   k_schur_diag: cc = psi / (sum_j z_j^2 k_j) / M_lumped  at every owned node
// ------------------------------------------------------------------------------------------
template <int G, bool COMPACT>   // COMPACT: t is the potential right-hand side on node-indexed vectors [n_nodes] (fused cycle)
__global__ void __launch_bounds__(NT)
k_phi_rhs(int n_nodes, double z0, double z1, double z2, const int32_t* __restrict__ pair_ptr,
          const int32_t* __restrict__ pair_col, const double* __restrict__ pair_M, const double* __restrict__ r,
          const double* __restrict__ z, const double* __restrict__ sc, double* __restrict__ t) {
    // t_phi = r_phi - sum_j z_j r_kj + M (sum_j z_j z_kj): equals r_phi - A_{phi,k} z_k for exact ion solves since
    // A_{phi,kj} = z_j (A_{kj,kj} - M) (KNPEMIx_problem.py:586-591,598,603,633-634), without the cancellation
    // that would amplify the V-cycle's error by 1/(1 - rho) ~ 10^2.
    const int node = (blockIdx.x * NT + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    double s = 0.0;
    if (node < n_nodes) {
        const int p0 = pair_ptr[node];
        const int deg = pair_ptr[node + 1] - p0;
        if (COMPACT) {   // sc[nb] = sum_j z_j z_kj[nb] was formed once per node (k_ion_charge): one 8-byte gather per pair
            for (int q = lane; q < deg; q += 4 * G) {
                const int qb = min(q + G, deg - 1), qc = min(q + 2 * G, deg - 1), qd = min(q + 3 * G, deg - 1);
                const int na = pair_col[p0 + q], nb = pair_col[p0 + qb], nc = pair_col[p0 + qc], nd = pair_col[p0 + qd];
                const double ma = pair_M[p0 + q], mb = pair_M[p0 + qb], mc = pair_M[p0 + qc], md = pair_M[p0 + qd];
                const double sa = sc[na], sb = sc[nb], scc = sc[nc], sd = sc[nd];
                s += ma * sa;
                if (q + G < deg) s += mb * sb;
                if (q + 2 * G < deg) s += mc * scc;
                if (q + 3 * G < deg) s += md * sd;
            }
        } else
        for (int q = lane; q < deg; q += 2 * G) {   // two predicated trips in flight per lane
            const int q2 = min(q + G, deg - 1);
            const int nb = pair_col[p0 + q], nb2 = pair_col[p0 + q2];
            const double m1 = pair_M[p0 + q], m2 = pair_M[p0 + q2];
            const double2 za = *reinterpret_cast<const double2*>(z + 4 * (size_t)nb);
            const double zc = z[4 * (size_t)nb + 2];
            const double2 ya = *reinterpret_cast<const double2*>(z + 4 * (size_t)nb2);
            const double yc = z[4 * (size_t)nb2 + 2];
            s += m1 * (z0 * za.x + z1 * za.y + z2 * zc);
            if (q + G < deg) s += m2 * (z0 * ya.x + z1 * ya.y + z2 * yc);
        }
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, G);
    if (lane == 0 && node < n_nodes) {
        const size_t i = 4 * (size_t)node;
        const double2 ra = *reinterpret_cast<const double2*>(r + i);
        const double2 rb = *reinterpret_cast<const double2*>(r + i + 2);
        const double tphi = rb.y - (z0 * ra.x + z1 * ra.y + z2 * rb.x) + s;
        if (COMPACT) {
            t[node] = tphi;
        } else {
            *reinterpret_cast<double2*>(t + i) = make_double2(0.0, 0.0);
            *reinterpret_cast<double2*>(t + i + 2) = make_double2(0.0, tphi);
        }
    }
}
// sc[n] = sum_j z_j z_kj[n]: the ion charge of the ion correction at every owned node (node-indexed)
__global__ void __launch_bounds__(NT) k_ion_charge(int n_nodes, double z0, double z1, double z2, const double* __restrict__ z, double* __restrict__ sc) {
    for (int n = blockIdx.x * NT + threadIdx.x; n < n_nodes; n += gridDim.x * NT) {
        const double2 a = *reinterpret_cast<const double2*>(z + 4 * (size_t)n);
        sc[n] = z0 * a.x + z1 * a.y + z2 * z[4 * (size_t)n + 2];
    }
}
template <bool COMPACT>
static void launch_phi_rhs(knp_ctx* ctx, const double* r, const double* z, double* t) {
    const int n = ctx->g.n_nodes_owned;
    const int G = COMPACT ? std::max(2, ctx->pc_group / 2) : ctx->pc_group;
    if (n <= 0) return;
    const double* sc = nullptr;
    if (COMPACT) {
        hipLaunchKernelGGL(k_ion_charge, dim3(std::min(nblocks(n), 4096)), dim3(NT), 0, ctx->stream, n, ctx->z[0], ctx->z[1], ctx->z[2], z, ctx->d_w2);
        sc = ctx->d_w2;
    }
#define KNP_PR(GG) hipLaunchKernelGGL((k_phi_rhs<GG, COMPACT>), dim3(nblocks((int64_t)n * GG)), dim3(NT), 0, ctx->stream, n, ctx->z[0], ctx->z[1], ctx->z[2], ctx->d_pair_ptr, ctx->d_pair_col, ctx->d_pair_M, r, z, sc, t)
    switch (G) {
        case 2: KNP_PR(2); break;
        case 4: KNP_PR(4); break;
        case 8: KNP_PR(8); break;
        case 16: KNP_PR(16); break;
        default: KNP_PR(32); break;
    }
#undef KNP_PR
}
// Literal block lower-triangular form (the reference's P with use_block_jacobi=False, KNPEMIx_problem.py:720-722: the phi rows
// of P keep their -D grad k flux, i.e. P_{phi,kj} = dt z_j D_j K = the (phi,k) block of A): t_phi = r_phi - P_{phi,k} z_k
template <int G, bool COMPACT>
__global__ void __launch_bounds__(NT)
k_phi_rhs_literal(int n_nodes, const int32_t* __restrict__ pair_ptr, const int32_t* __restrict__ pair_col, const double* __restrict__ ac,
                  const double* __restrict__ r, const double* __restrict__ z, double* __restrict__ t) {
    const int node = (blockIdx.x * NT + threadIdx.x) / G;
    const int lane = threadIdx.x & (G - 1);
    double s = 0.0;
    if (node < n_nodes) {
        const int p0 = pair_ptr[node];
        const int deg = pair_ptr[node + 1] - p0;
        for (int q = lane; q < deg; q += G) {
            const size_t p = (size_t)p0 + q;
            const int nb = pair_col[p];
            const double a0 = ac[6 * p + 3];
            const double2 a12 = *reinterpret_cast<const double2*>(ac + 6 * p + 4);
            const double2 za = *reinterpret_cast<const double2*>(z + 4 * (size_t)nb);
            const double zc = z[4 * (size_t)nb + 2];
            s += a0 * za.x + a12.x * za.y + a12.y * zc;
        }
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, G);
    if (lane == 0 && node < n_nodes) {
        const size_t i = 4 * (size_t)node;
        const double tphi = r[i + 3] - s;
        if (COMPACT) {
            t[node] = tphi;
        } else {
            *reinterpret_cast<double2*>(t + i) = make_double2(0.0, 0.0);
            *reinterpret_cast<double2*>(t + i + 2) = make_double2(0.0, tphi);
        }
    }
}
template <bool COMPACT>
static void launch_phi_rhs_literal(knp_ctx* ctx, const double* r, const double* z, double* t) {
    const int n = ctx->g.n_nodes_owned, G = ctx->pc_group;
    if (n <= 0) return;
#define KNP_PL(GG) hipLaunchKernelGGL((k_phi_rhs_literal<GG, COMPACT>), dim3(nblocks((int64_t)n * GG)), dim3(NT), 0, ctx->stream, n, ctx->d_pair_ptr, ctx->d_pair_col, ctx->d_ac, r, z, t)
    switch (G) {
        case 4: KNP_PL(4); break;
        case 8: KNP_PL(8); break;
        case 16: KNP_PL(16); break;
        default: KNP_PL(32); break;
    }
#undef KNP_PL
}
__global__ void __launch_bounds__(NT) k_schur_fin(int n_nodes, const double* __restrict__ cc, const double* __restrict__ t,
                                                  const double* __restrict__ w, double* __restrict__ z) {
    const int n = blockIdx.x * NT + threadIdx.x;
    if (n >= n_nodes) return;
    z[(size_t)4 * n + 3] = w[(size_t)4 * n + 3] + (cc ? cc[n] * t[(size_t)4 * n + 3] : 0.0);
}
__global__ void __launch_bounds__(NT)
k_schur_diag(int n_nodes, double psi, double z0, double z1, double z2, const int32_t* __restrict__ node_vertex,
             const uint8_t* __restrict__ node_side, FieldPtrs f, const double* __restrict__ ML, double* __restrict__ cc) {
    const int n = blockIdx.x * NT + threadIdx.x;
    if (n >= n_nodes) return;
    const int v = node_vertex[n];
    const int sd = node_side[n];
    const double s = z0 * z0 * (sd ? f.ke[0] : f.ki[0])[v] + z1 * z1 * (sd ? f.ke[1] : f.ki[1])[v] +
                     z2 * z2 * (sd ? f.ke[2] : f.ki[2])[v];
    cc[n] = psi / (s * ML[n]);
}

template <int MODE, int TAG, typename VT>
static void launch_spmv_t(hipStream_t st, int lanes, int n_rows, const int32_t* rp, const int32_t* ci,
                          const VT* v, const double* x, const double* b, double* y) {
    if (n_rows <= 0) return;
    switch (lanes) {
        case 2: hipLaunchKernelGGL((k_spmv<2, MODE, TAG, VT>), dim3(nblocks((int64_t)n_rows * 2)), dim3(NT), 0, st, n_rows, rp, ci, v, x, b, y); break;
        case 4: hipLaunchKernelGGL((k_spmv<4, MODE, TAG, VT>), dim3(nblocks((int64_t)n_rows * 4)), dim3(NT), 0, st, n_rows, rp, ci, v, x, b, y); break;
        case 8: hipLaunchKernelGGL((k_spmv<8, MODE, TAG, VT>), dim3(nblocks((int64_t)n_rows * 8)), dim3(NT), 0, st, n_rows, rp, ci, v, x, b, y); break;
        case 16: hipLaunchKernelGGL((k_spmv<16, MODE, TAG, VT>), dim3(nblocks((int64_t)n_rows * 16)), dim3(NT), 0, st, n_rows, rp, ci, v, x, b, y); break;
        case 32: hipLaunchKernelGGL((k_spmv<32, MODE, TAG, VT>), dim3(nblocks((int64_t)n_rows * 32)), dim3(NT), 0, st, n_rows, rp, ci, v, x, b, y); break;
        default: hipLaunchKernelGGL((k_spmv<64, MODE, TAG, VT>), dim3(nblocks((int64_t)n_rows * 64)), dim3(NT), 0, st, n_rows, rp, ci, v, x, b, y); break;
    }
}
template <int MODE, int TAG = 0>
static void launch_spmv(hipStream_t st, int lanes, int n_rows, const int32_t* rp, const int32_t* ci,
                        const double* v, const double* x, const double* b, double* y) {
    launch_spmv_t<MODE, TAG, double>(st, lanes, n_rows, rp, ci, v, x, b, y);
}
// AMG level operator stored in fp64 (v) or fp32 (vf != nullptr)
template <int MODE>
static void launch_spmv_mp(hipStream_t st, int lanes, int n_rows, const int32_t* rp, const int32_t* ci,
                           const double* v, const float* vf, const double* x, const double* b, double* y) {
    if (vf) launch_spmv_t<MODE, 0, float>(st, lanes, n_rows, rp, ci, vf, x, b, y);
    else launch_spmv_t<MODE, 0, double>(st, lanes, n_rows, rp, ci, v, x, b, y);
}

// Prolongation with a compact row list: a hierarchy restricted to one field class has a prolongator whose rows for the
// other fields are empty (3/4 of the rows for the potential hierarchy); only the non-empty rows are visited.
//   y[rows[i]] += sum_k v[k] x[ci[k]],  k in [rp[i], rp[i+1])
template <int L, typename VT>
__global__ void __launch_bounds__(NT)
k_prolong_rows(int n_act, const int32_t* __restrict__ rows, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
               const VT* __restrict__ v, const double* __restrict__ x, double* __restrict__ y) {
    const int gid = blockIdx.x * NT + threadIdx.x;
    const int i = gid / L;
    const int lane = threadIdx.x & (L - 1);
    double s = 0.0;
    if (i < n_act) s = row_dot4<L, VT>(rp[i], rp[i + 1], lane, ci, v, x);
#pragma unroll
    for (int o = L >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, L);
    if (lane == 0 && i < n_act) y[rows[i]] += s;
}
template <typename VT>
static void launch_prolong_rows_t(hipStream_t st, int lanes, int n_act, const int32_t* rows, const int32_t* rp, const int32_t* ci,
                                  const VT* v, const double* x, double* y) {
    if (n_act <= 0) return;
    switch (lanes) {
        case 2: hipLaunchKernelGGL((k_prolong_rows<2, VT>), dim3(nblocks((int64_t)n_act * 2)), dim3(NT), 0, st, n_act, rows, rp, ci, v, x, y); break;
        case 4: hipLaunchKernelGGL((k_prolong_rows<4, VT>), dim3(nblocks((int64_t)n_act * 4)), dim3(NT), 0, st, n_act, rows, rp, ci, v, x, y); break;
        case 8: hipLaunchKernelGGL((k_prolong_rows<8, VT>), dim3(nblocks((int64_t)n_act * 8)), dim3(NT), 0, st, n_act, rows, rp, ci, v, x, y); break;
        default: hipLaunchKernelGGL((k_prolong_rows<16, VT>), dim3(nblocks((int64_t)n_act * 16)), dim3(NT), 0, st, n_act, rows, rp, ci, v, x, y); break;
    }
}

// Restriction fused with the first Chebyshev step of the coarse level (zero initial guess):
//   b_c = R r ;  d_c = x_c = (1/theta_c) D_c^-1 b_c       -- one launch instead of two on every level of every cycle
template <int L, typename VT>
__global__ void __launch_bounds__(NT)
k_restrict_first(int n_rows, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci, const VT* __restrict__ v,
                 const double* __restrict__ x, double* __restrict__ y, double c, const double* __restrict__ dinv,
                 double* __restrict__ d, double* __restrict__ xo) {
    const int gid = blockIdx.x * NT + threadIdx.x;
    const int row = gid / L;
    const int lane = threadIdx.x & (L - 1);
    double s = 0.0;
    if (row < n_rows) s = row_dot4<L, VT>(rp[row], rp[row + 1], lane, ci, v, x);
#pragma unroll
    for (int o = L >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, L);
    if (lane == 0 && row < n_rows) {
        y[row] = s;
        const double t = c * dinv[row] * s;
        d[row] = t;
        xo[row] = t;
    }
}
template <typename VT>
static void launch_restrict_first_t(hipStream_t st, int lanes, int n_rows, const int32_t* rp, const int32_t* ci, const VT* v,
                                    const double* x, double* y, double c, const double* dinv, double* d, double* xo) {
    if (n_rows <= 0) return;
    switch (lanes) {
        case 2: hipLaunchKernelGGL((k_restrict_first<2, VT>), dim3(nblocks((int64_t)n_rows * 2)), dim3(NT), 0, st, n_rows, rp, ci, v, x, y, c, dinv, d, xo); break;
        case 4: hipLaunchKernelGGL((k_restrict_first<4, VT>), dim3(nblocks((int64_t)n_rows * 4)), dim3(NT), 0, st, n_rows, rp, ci, v, x, y, c, dinv, d, xo); break;
        case 8: hipLaunchKernelGGL((k_restrict_first<8, VT>), dim3(nblocks((int64_t)n_rows * 8)), dim3(NT), 0, st, n_rows, rp, ci, v, x, y, c, dinv, d, xo); break;
        case 16: hipLaunchKernelGGL((k_restrict_first<16, VT>), dim3(nblocks((int64_t)n_rows * 16)), dim3(NT), 0, st, n_rows, rp, ci, v, x, y, c, dinv, d, xo); break;
        case 32: hipLaunchKernelGGL((k_restrict_first<32, VT>), dim3(nblocks((int64_t)n_rows * 32)), dim3(NT), 0, st, n_rows, rp, ci, v, x, y, c, dinv, d, xo); break;
        default: hipLaunchKernelGGL((k_restrict_first<64, VT>), dim3(nblocks((int64_t)n_rows * 64)), dim3(NT), 0, st, n_rows, rp, ci, v, x, y, c, dinv, d, xo); break;
    }
}

// ------------------------------------------------------------------------------------------
// Fused V(1,1) legs (Chebyshev degree 1, zero initial guess).  With x0 = c Dinv b the pre-smoothing and the residual are
//   down (level 0):    r = b - c Pt b,              Pt = P Dinv   (columns of the pair-major P scaled once, at setup)
// and with S = (I - c2 Dinv A) Pprol (one sparse product at setup) the prolongation and the post-smoothing step are
//   up   (any level):  x = x0 + c2 Dinv r + S xc
// -- the same V-cycle, two gathers per level instead of four kernels, no intermediate iterate.
// FM 0: all four fields, 1: ion fields (both on vectors with 4 unknowns per node), 2: potential on COMPACT vectors [n_nodes],
// 3: potential on vectors with 4 unknowns per node (distributed levels keep the DoF layout their halo plans were built for)
// ------------------------------------------------------------------------------------------
template <int G, int FM, typename VT>
__global__ void __launch_bounds__(NT)
k_l0_down(int n_list, const int32_t* __restrict__ pair_ptr, const int32_t* __restrict__ pair_col, const VT* __restrict__ pt,
          const double* __restrict__ b, double c, double* __restrict__ r, const int32_t* __restrict__ nodes = nullptr) {
    // `nodes` (optional): the nodes to process -- interior / boundary split of the multi-GPU path (the halo of b runs next to the
    // interior nodes, the nodes with a ghost neighbour follow the join)
    const int idx = (blockIdx.x * NT + threadIdx.x) / G;
    const int n_nodes = n_list;
    const int node = (nodes && idx < n_list) ? nodes[idx] : idx;
    const bool in_range = idx < n_list;
    const int lane = threadIdx.x & (G - 1);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (in_range) {
        const int p0 = pair_ptr[node];
        const int deg = pair_ptr[node + 1] - p0;
        if (FM >= 2) {   // 4 B value + 4 B index + 8 B gather per pair: four predicated trips in flight per lane (see row_dot4)
            const int32_t* __restrict__ col = pair_col + p0;
            const VT* __restrict__ val = pt + p0;
            constexpr int ST = FM == 3 ? 4 : 1, OF = FM == 3 ? 3 : 0;
            for (int q = lane; q < deg; q += 4 * G) {
                const int qb = min(q + G, deg - 1), qc = min(q + 2 * G, deg - 1), qd = min(q + 3 * G, deg - 1);
                const int na = col[q], nb = col[qb], nc = col[qc], nd = col[qd];
                const double va = (double)val[q], vb = (double)val[qb], vc = (double)val[qc], vd = (double)val[qd];
                const double xa = b[ST * (size_t)na + OF], xb = b[ST * (size_t)nb + OF], xc = b[ST * (size_t)nc + OF], xd = b[ST * (size_t)nd + OF];
                s3 += va * xa;
                if (q + G < deg) s3 += vb * xb;
                if (q + 2 * G < deg) s3 += vc * xc;
                if (q + 3 * G < deg) s3 += vd * xd;
            }
        } else {         // four predicated trips in flight per lane: 4 x (4 B index + 16/32 B values + 32 B gather)
            for (int q = lane; q < deg; q += 4 * G) {
                int nbv[4];
                double av[4][4];
                double2 xa[4], xb[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int qi = min(q + i * G, deg - 1);
                    nbv[i] = pair_col[p0 + qi];
                    if (sizeof(VT) == 4) {
                        const float4 pq = *reinterpret_cast<const float4*>(pt + 4 * (size_t)(p0 + qi));
                        av[i][0] = pq.x; av[i][1] = pq.y; av[i][2] = pq.z; av[i][3] = pq.w;
                    } else {
                        const double2 u = *reinterpret_cast<const double2*>(pt + 4 * (size_t)(p0 + qi));
                        const double2 w = *reinterpret_cast<const double2*>(pt + 4 * (size_t)(p0 + qi) + 2);
                        av[i][0] = u.x; av[i][1] = u.y; av[i][2] = w.x; av[i][3] = w.y;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xa[i] = *reinterpret_cast<const double2*>(b + 4 * (size_t)nbv[i]);
                    xb[i] = *reinterpret_cast<const double2*>(b + 4 * (size_t)nbv[i] + 2);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (i == 0 || q + i * G < deg) {
                        s0 += av[i][0] * xa[i].x;
                        s1 += av[i][1] * xa[i].y;
                        s2 += av[i][2] * xb[i].x;
                        if (FM == 0) s3 += av[i][3] * xb[i].y;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = G >> 1; o > 0; o >>= 1) {
        if (FM < 2) {
            s0 += __shfl_xor(s0, o, G);
            s1 += __shfl_xor(s1, o, G);
            s2 += __shfl_xor(s2, o, G);
        }
        if (FM != 1) s3 += __shfl_xor(s3, o, G);
    }
    (void)n_nodes;
    if (lane == 0 && in_range) {
        if (FM == 2) {
            r[node] = b[node] - c * s3;
        } else if (FM == 3) {
            r[4 * (size_t)node + 3] = b[4 * (size_t)node + 3] - c * s3;
        } else {
            const size_t i = 4 * (size_t)node;
            const double2 b0 = *reinterpret_cast<const double2*>(b + i);
            const double2 b1 = *reinterpret_cast<const double2*>(b + i + 2);
            *reinterpret_cast<double2*>(r + i) = make_double2(b0.x - c * s0, b0.y - c * s1);
            *reinterpret_cast<double2*>(r + i + 2) = make_double2(b1.x - c * s2, FM == 0 ? b1.y - c * s3 : 0.0);
        }
    }
}
template <int FM, typename VT>
static void launch_l0_down_t(hipStream_t st, int G, int n_nodes, const int32_t* pp, const int32_t* pc, const VT* pt, const double* b, double c, double* r,
                             const int32_t* nodes = nullptr) {
    if (n_nodes <= 0) return;
    switch (G) {
        case 2: hipLaunchKernelGGL((k_l0_down<2, FM, VT>), dim3(nblocks((int64_t)n_nodes * 2)), dim3(NT), 0, st, n_nodes, pp, pc, pt, b, c, r, nodes); break;
        case 4: hipLaunchKernelGGL((k_l0_down<4, FM, VT>), dim3(nblocks((int64_t)n_nodes * 4)), dim3(NT), 0, st, n_nodes, pp, pc, pt, b, c, r, nodes); break;
        case 8: hipLaunchKernelGGL((k_l0_down<8, FM, VT>), dim3(nblocks((int64_t)n_nodes * 8)), dim3(NT), 0, st, n_nodes, pp, pc, pt, b, c, r, nodes); break;
        case 16: hipLaunchKernelGGL((k_l0_down<16, FM, VT>), dim3(nblocks((int64_t)n_nodes * 16)), dim3(NT), 0, st, n_nodes, pp, pc, pt, b, c, r, nodes); break;
        default: hipLaunchKernelGGL((k_l0_down<32, FM, VT>), dim3(nblocks((int64_t)n_nodes * 32)), dim3(NT), 0, st, n_nodes, pp, pc, pt, b, c, r, nodes); break;
    }
}
template <typename VT>
static void launch_l0_down(hipStream_t st, int fm, int G, int n_nodes, const int32_t* pp, const int32_t* pc, const VT* pt, const double* b, double c, double* r,
                           const int32_t* nodes = nullptr) {
    if (fm == 2) launch_l0_down_t<2, VT>(st, G, n_nodes, pp, pc, pt, b, c, r, nodes);
    else if (fm == 3) launch_l0_down_t<3, VT>(st, G, n_nodes, pp, pc, pt, b, c, r, nodes);
    else if (fm == 1) launch_l0_down_t<1, VT>(st, G, n_nodes, pp, pc, pt, b, c, r, nodes);
    else launch_l0_down_t<0, VT>(st, G, n_nodes, pp, pc, pt, b, c, r, nodes);
}

// MODE 0: z[row] = x0 + c2 dinv r + S xc for the listed rows (rows == nullptr: all rows 0..n_act-1); x0 = xin[row] when xin is given,
//         else c dinv b.   MODE 1: compact potential vectors (row = node): z[4 row + 3] = the same + cc[row] b[row]  (the
//         Cahouet-Chabard Schur term of the block-triangular preconditioner, so that no separate kernel adds it)
template <int L, typename VT, int MODE>
__global__ void __launch_bounds__(NT)
k_level_up(int n_act, const int32_t* __restrict__ rows, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci,
           const VT* __restrict__ v, const double* __restrict__ xc, const double* __restrict__ dinv, const double* __restrict__ b,
           const double* __restrict__ r, const double* xin, double c, double c2, double* z, const double* __restrict__ cc) {
    const int gid = blockIdx.x * NT + threadIdx.x;
    const int i = gid / L;
    const int lane = threadIdx.x & (L - 1);
    double s = 0.0;
    int row = 0;
    double di = 0.0, bi = 0.0, ri = 0.0, x0 = 0.0, ci_ = 0.0;
    if (i < n_act) {
        const int k0 = rp[i], e = rp[i + 1];
        if (lane == 0) {   // the epilogue's operands travel together with the gathers instead of after the reduction
            row = rows ? rows[i] : i;
            di = dinv[row];
            ri = r[row];
            if (xin) x0 = xin[row];
            if (!xin || MODE == 1) bi = b[row];
            if (MODE == 1 && cc) ci_ = cc[row];
        }
        s = row_dot4<L, VT>(k0, e, lane, ci, v, xc);
    }
#pragma unroll
    for (int o = L >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, L);
    if (lane == 0 && i < n_act) {
        const double base = xin ? x0 : c * di * bi;
        const double out = base + c2 * di * ri + s;
        if (MODE == 1) z[4 * (size_t)row + 3] = out + ci_ * bi;
        else z[row] = out;
    }
}
template <typename VT, int MODE>
static void launch_level_up_t(hipStream_t st, int lanes, int n_act, const int32_t* rows, const int32_t* rp, const int32_t* ci, const VT* v,
                              const double* xc, const double* dinv, const double* b, const double* r, const double* xin, double c, double c2,
                              double* z, const double* cc) {
    if (n_act <= 0) return;
#define KNP_UP(LL) hipLaunchKernelGGL((k_level_up<LL, VT, MODE>), dim3(nblocks((int64_t)n_act * LL)), dim3(NT), 0, st, n_act, rows, rp, ci, v, xc, dinv, b, r, xin, c, c2, z, cc)
    switch (lanes) {
        case 2: KNP_UP(2); break;
        case 4: KNP_UP(4); break;
        case 8: KNP_UP(8); break;
        case 16: KNP_UP(16); break;
        case 32: KNP_UP(32); break;
        default: KNP_UP(64); break;
    }
#undef KNP_UP
}

// ------------------------------------------------------------------------------------------
// Node-blocked legs of the fused cycle (KnpBlockedCsr): the NF fields of a node are decoupled and share one pattern, so a
// row is a NODE: one float4 per entry ({v0, v1, v2, column} for NF == 3), one gather of NF consecutive unknowns, NF sums per
// lane.  XS = unknowns per column node in x (4 on level 0, where vectors hold four unknowns per node; NF on coarse levels).
// ------------------------------------------------------------------------------------------
template <int NF, int XS>
__device__ __forceinline__ void bload_x(const double* __restrict__ x, int j, double xv[4]) {
    const double* __restrict__ p = x + (size_t)XS * j;
    if (XS == 4) {   // 32 B aligned
        const double2 a = *reinterpret_cast<const double2*>(p);
        const double2 b = *reinterpret_cast<const double2*>(p + 2);
        xv[0] = a.x; xv[1] = a.y; xv[2] = b.x; xv[3] = b.y;
    } else if (NF == 3 && XS == 3) {
        // three unknowns per node, 24-byte records: every other one starts 8 bytes off a 16-byte boundary -- one 16-byte load of the
        // aligned pair and one 8-byte load of the remaining unknown, chosen by address arithmetic (no divergence), instead of three
        // 8-byte gathers
        const int odd = (int)((reinterpret_cast<uintptr_t>(p) >> 3) & 1);      // (from the address: sub-vectors need not start on 16 bytes)
        const double2 pr = *reinterpret_cast<const double2*>(p + odd);
        const double s1 = p[odd ? 0 : 2];
        xv[0] = odd ? s1 : pr.x;
        xv[1] = odd ? pr.x : pr.y;
        xv[2] = odd ? pr.y : s1;
        xv[3] = 0.0;
    } else {
        xv[0] = p[0]; xv[1] = p[1]; xv[2] = p[2];
        xv[3] = NF == 4 ? p[3] : 0.0;
    }
}
// four predicated entries in flight per lane (each: 16 B entry + 24/32 B gather)
template <int L, int NF, int XS>
__device__ __forceinline__ void brow_dot(int k0, int e, int lane, const float4* __restrict__ ev, const int32_t* __restrict__ ci,
                                         const double* __restrict__ x, double s[4]) {
    for (int q = k0 + lane; q < e; q += 4 * L) {
        int qq[4];
        float4 a[4];
        int j[4];
        double xv[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qq[i] = min(q + i * L, e - 1);
            a[i] = ev[qq[i]];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) j[i] = NF == 3 ? __float_as_int(a[i].w) : ci[qq[i]];
#pragma unroll
        for (int i = 0; i < 4; ++i) bload_x<NF, XS>(x, j[i], xv[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i == 0 || q + i * L < e) {
                s[0] += (double)a[i].x * xv[i][0];
                s[1] += (double)a[i].y * xv[i][1];
                s[2] += (double)a[i].z * xv[i][2];
                if (NF == 4) s[3] += (double)a[i].w * xv[i][3];
            }
        }
    }
}
template <int L, int NF>
__device__ __forceinline__ void breduce(double s[4]) {
#pragma unroll
    for (int o = L >> 1; o > 0; o >>= 1) {
        s[0] += __shfl_xor(s[0], o, L);
        s[1] += __shfl_xor(s[1], o, L);
        s[2] += __shfl_xor(s[2], o, L);
        if (NF == 4) s[3] += __shfl_xor(s[3], o, L);
    }
}
// After the butterfly every lane of the group holds all NF sums: lane k (< NF) finishes field k, so the epilogue's loads and
// stores are one instruction over NF lanes instead of NF instructions on lane 0 (groups of 2 lanes: lane 0 does all of them).
template <int NF>
__device__ __forceinline__ double bpick(const double s[4], int k) {
    return k == 0 ? s[0] : k == 1 ? s[1] : (NF == 4 && k == 3) ? s[3] : s[2];
}
// b_c = R r  [; d_c = x_c = c Dinv_c b_c when dinv is given]          rows: coarse nodes (NF unknowns each)
template <int L, int NF, int XS>
__global__ void __launch_bounds__(NT)
k_brestrict(int n_rows, const int32_t* __restrict__ rp, const float4* __restrict__ ev, const int32_t* __restrict__ ci,
            const double* __restrict__ x, double* __restrict__ y, double c, const double* __restrict__ dinv,
            double* __restrict__ d, double* __restrict__ xo) {
    const int row = (blockIdx.x * NT + threadIdx.x) / L;
    const int lane = threadIdx.x & (L - 1);
    constexpr int KL = L >= NF ? 1 : NF;   // fields finished per epilogue lane
    const bool fin = row < n_rows && (L >= NF ? lane < NF : lane == 0);
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    double di[KL];
    if (row < n_rows) {
        if (fin && dinv) {
#pragma unroll
            for (int k = 0; k < KL; ++k) di[k] = dinv[(size_t)NF * row + (L >= NF ? lane : k)];
        }
        brow_dot<L, NF, XS>(rp[row], rp[row + 1], lane, ev, ci, x, s);
    }
    breduce<L, NF>(s);
    if (fin) {
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            const int f = L >= NF ? lane : k;
            const size_t i = (size_t)NF * row + f;
            const double sk = bpick<NF>(s, f);
            y[i] = sk;
            if (dinv) {
                const double t = c * di[k] * sk;
                d[i] = t;
                xo[i] = t;
            }
        }
    }
}
// r = b - A x on a coarse level (NF unknowns per node in every vector)
template <int L, int NF>
__global__ void __launch_bounds__(NT)
k_bresidual(int n_rows, const int32_t* __restrict__ rp, const float4* __restrict__ ev, const int32_t* __restrict__ ci,
            const double* __restrict__ x, const double* __restrict__ b, double* __restrict__ y) {
    const int row = (blockIdx.x * NT + threadIdx.x) / L;
    const int lane = threadIdx.x & (L - 1);
    constexpr int KL = L >= NF ? 1 : NF;
    const bool fin = row < n_rows && (L >= NF ? lane < NF : lane == 0);
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    double bv[KL];
    if (row < n_rows) {
        if (fin) {
#pragma unroll
            for (int k = 0; k < KL; ++k) bv[k] = b[(size_t)NF * row + (L >= NF ? lane : k)];
        }
        brow_dot<L, NF, NF>(rp[row], rp[row + 1], lane, ev, ci, x, s);
    }
    breduce<L, NF>(s);
    if (fin) {
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            const int f = L >= NF ? lane : k;
            y[(size_t)NF * row + f] = bv[k] - bpick<NF>(s, f);
        }
    }
}
// z = x0 + c2 Dinv r + S xc for the NF unknowns of every node row; x0 = xin when given, else c Dinv b.  RS = unknowns per ROW node
// in dinv, b, r, xin, z (4 on level 0 -- the fourth one, the potential, is not touched when NF == 3)
template <int L, int NF, int RS>
__global__ void __launch_bounds__(NT)
k_blevel_up(int n_rows, const int32_t* __restrict__ rp, const float4* __restrict__ ev, const int32_t* __restrict__ ci,
            const double* __restrict__ xc, const double* __restrict__ dinv, const double* __restrict__ b, const double* __restrict__ r,
            const double* xin, double c, double c2, double* z) {
    const int row = (blockIdx.x * NT + threadIdx.x) / L;
    const int lane = threadIdx.x & (L - 1);
    constexpr int KL = L >= NF ? 1 : NF;
    const bool fin = row < n_rows && (L >= NF ? lane < NF : lane == 0);
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    double base[KL];
    if (row < n_rows) {
        if (fin) {   // the epilogue's operands travel together with the gathers
#pragma unroll
            for (int k = 0; k < KL; ++k) {
                const size_t i = (size_t)RS * row + (L >= NF ? lane : k);
                const double di = dinv[i];
                base[k] = (xin ? xin[i] : c * di * b[i]) + c2 * di * r[i];
            }
        }
        brow_dot<L, NF, NF>(rp[row], rp[row + 1], lane, ev, ci, xc, s);
    }
    breduce<L, NF>(s);
    if (fin) {
#pragma unroll
        for (int k = 0; k < KL; ++k) {
            const int f = L >= NF ? lane : k;
            z[(size_t)RS * row + f] = base[k] + bpick<NF>(s, f);
        }
    }
}
#define KNP_BL_SWITCH(LANES, CALL) \
    switch (LANES) {               \
        case 2: CALL(2); break;    \
        case 4: CALL(4); break;    \
        case 8: CALL(8); break;    \
        case 16: CALL(16); break;  \
        default: CALL(32); break;  \
    }
template <int NF, int XS>
static void launch_brestrict_t(hipStream_t st, const KnpBlockedCsr& M, const double* x, double* y, double c, const double* dinv, double* d, double* xo) {
    if (M.n_rows <= 0) return;
#define KNP_BR(LL) hipLaunchKernelGGL((k_brestrict<LL, NF, XS>), dim3(nblocks((int64_t)M.n_rows * LL)), dim3(NT), 0, st, M.n_rows, M.rp, M.ev, M.ci, x, y, c, dinv, d, xo)
    KNP_BL_SWITCH(M.lanes, KNP_BR)
#undef KNP_BR
}
static void launch_brestrict(hipStream_t st, int nf, int xs, const KnpBlockedCsr& M, const double* x, double* y, double c, const double* dinv, double* d, double* xo) {
    if (nf == 4) launch_brestrict_t<4, 4>(st, M, x, y, c, dinv, d, xo);
    else if (xs == 4) launch_brestrict_t<3, 4>(st, M, x, y, c, dinv, d, xo);
    else launch_brestrict_t<3, 3>(st, M, x, y, c, dinv, d, xo);
}
template <int NF>
static void launch_bresidual_t(hipStream_t st, const KnpBlockedCsr& M, const double* x, const double* b, double* y) {
    if (M.n_rows <= 0) return;
#define KNP_BA(LL) hipLaunchKernelGGL((k_bresidual<LL, NF>), dim3(nblocks((int64_t)M.n_rows * LL)), dim3(NT), 0, st, M.n_rows, M.rp, M.ev, M.ci, x, b, y)
    KNP_BL_SWITCH(M.lanes, KNP_BA)
#undef KNP_BA
}
template <int NF, int RS>
static void launch_blevel_up_t(hipStream_t st, const KnpBlockedCsr& M, const double* xc, const double* dinv, const double* b, const double* r,
                               const double* xin, double c, double c2, double* z) {
    if (M.n_rows <= 0) return;
#define KNP_BU(LL) hipLaunchKernelGGL((k_blevel_up<LL, NF, RS>), dim3(nblocks((int64_t)M.n_rows * LL)), dim3(NT), 0, st, M.n_rows, M.rp, M.ev, M.ci, xc, dinv, b, r, xin, c, c2, z)
    KNP_BL_SWITCH(M.lanes, KNP_BU)
#undef KNP_BU
}
static void launch_blevel_up(hipStream_t st, int nf, int rs, const KnpBlockedCsr& M, const double* xc, const double* dinv, const double* b, const double* r,
                             const double* xin, double c, double c2, double* z) {
    if (nf == 4) launch_blevel_up_t<4, 4>(st, M, xc, dinv, b, r, xin, c, c2, z);
    else if (rs == 4) launch_blevel_up_t<3, 4>(st, M, xc, dinv, b, r, xin, c, c2, z);
    else launch_blevel_up_t<3, 3>(st, M, xc, dinv, b, r, xin, c, c2, z);
}

// setup helpers of the fused cycle
//   At = c A Dinv on the pattern of a CSR level operator (levels >= 1 in fused form, see amg_vcycle)
template <typename VI, typename VO>
__global__ void __launch_bounds__(NT)
k_build_at(int64_t nnz, const int32_t* __restrict__ ci, const VI* __restrict__ a, const double* __restrict__ dinv, double c, VO* __restrict__ out) {
    for (int64_t k = (int64_t)blockIdx.x * NT + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * NT)
        out[k] = (VO)(c * (double)a[k] * dinv[ci[k]]);
}
//   Pt = P Dinv (pair-major, 4 fields per pair) and its compact potential part
template <typename VT>
__global__ void __launch_bounds__(NT)
k_build_pt(int64_t n_pairs, const int32_t* __restrict__ pair_col, const double* __restrict__ pv, const double* __restrict__ dinv,
           VT* __restrict__ pt, VT* __restrict__ pt_phi) {
    for (int64_t p = (int64_t)blockIdx.x * NT + threadIdx.x; p < n_pairs; p += (int64_t)gridDim.x * NT) {
        const int nb = pair_col[p];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const double val = pv[4 * p + f] * dinv[4 * (size_t)nb + f];
            if (pt) pt[4 * p + f] = (VT)val;
            if (f == 3 && pt_phi) pt_phi[p] = (VT)val;
        }
    }
}
__global__ void __launch_bounds__(NT) k_gather_stride4(int n, const double* __restrict__ in, int off, double* __restrict__ out) {
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) out[i] = in[4 * (size_t)i + off];
}
__global__ void __launch_bounds__(NT) k_shift2(int64_t n, const int32_t* __restrict__ in, int32_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) out[i] = in[i] >> 2;
}

static int pick_lanes(double avg_nnz_per_row, int role = 0) {   // role 0 level operator, 1 prolongator, 2 restrictor
    // lanes per row of the generic CSR kernels (AMG level operators, transfer operators).  About 6 entries per lane:
    // measured on MI355X, halving the lane counts of the first version (3 per lane) gave +4 % on square512
    static const double scale = getenv("KNP_LANE_SCALE") ? atof(getenv("KNP_LANE_SCALE")) : 1.0;   // tuning knobs
    static const double scale_p = getenv("KNP_LANE_SCALE_P") ? atof(getenv("KNP_LANE_SCALE_P")) : 1.0;
    static const double scale_r = getenv("KNP_LANE_SCALE_R") ? atof(getenv("KNP_LANE_SCALE_R")) : 1.0;
    avg_nnz_per_row *= scale * (role == 1 ? scale_p : role == 2 ? scale_r : 1.0);
    // row_dot4 keeps four entries per lane in flight: lanes = entries / 4, rounded up to a power of two
    if (avg_nnz_per_row <= 8.0) return 2;
    if (avg_nnz_per_row <= 16.0) return 4;
    if (avg_nnz_per_row <= 32.0) return 8;
    if (avg_nnz_per_row <= 80.0) return 16;
    if (avg_nnz_per_row <= 192.0) return 32;
    return 64;
}

// ------------------------------------------------------------------------------------------
// K5/K6: orthogonalisation and vector kernels
// ------------------------------------------------------------------------------------------
// ---- GMRES bookkeeping on the device (one wave): the host only reads back the residual estimate and a flag --------
// gm layout (doubles): H [(m+1) x m, column major] | cs [m] | sn [m] | g [m+1] | y [m] | state {||w'||^2, flag, residual}
struct GmLayout {
    int m;
    __host__ __device__ int H(int i, int j) const { return i + j * (m + 1); }
    __host__ __device__ int cs() const { return (m + 1) * m; }
    __host__ __device__ int sn() const { return cs() + m; }
    __host__ __device__ int g() const { return sn() + m; }
    __host__ __device__ int y() const { return g() + m + 1; }
    __host__ __device__ int st() const { return y() + m; }
    __host__ __device__ int size() const { return st() + 8; }
};
// ||w'||^2 < GM_CANCEL ||w||^2: Pythagoras has lost 8 of 16 digits, take the explicit norm.  The Arnoldi relation holds exactly
// for ANY value used consistently as h_{j+1,j} and as the normalisation of v_{j+1}; an inexact norm only makes |v_{j+1}| differ
// from 1 by that relative error, which perturbs the least-squares weights, not the Krylov space (well preconditioned systems
// routinely have |w'| ~ 1e-3 |w|: a tighter guard would pay a second reduction in most of their iterations).
static constexpr double GM_CANCEL = 1e-8;

__device__ __forceinline__ void givens_body(GmLayout L, int j, int has_ns, double inv_cnt, const double* __restrict__ red, int explicit_slot,
                                            double* __restrict__ gm, double* mirror, volatile int64_t* seq, int64_t seq_val) {
    double* st = gm + L.st();
    double nrm2;
    bool cancel = false;
    if (explicit_slot < 0) {
        const double ww = red[j + 1 + has_ns];
        double s = 0.0;
        for (int i = 0; i <= j; ++i) s += red[i] * red[i];
        if (has_ns) s += red[j + 1] * red[j + 1] * inv_cnt;   // component along the normalised null-space vector
        nrm2 = ww - s;
        cancel = !(nrm2 > GM_CANCEL * ww) && ww > 0.0;        // (NaNs fall through to the breakdown test below)
    } else {
        nrm2 = red[explicit_slot];
    }
    double flag = 0.0, res = 0.0;
    if (cancel) {
        flag = 1.0;
        nrm2 = 1.0;
    } else {
        double* h = gm + L.H(0, j);
        for (int i = 0; i <= j; ++i) h[i] = red[i];
        h[j + 1] = sqrt(fmax(nrm2, 0.0));
        double* cs = gm + L.cs();
        double* sn = gm + L.sn();
        double* g = gm + L.g();
        for (int i = 0; i < j; ++i) {
            const double t = cs[i] * h[i] + sn[i] * h[i + 1];
            h[i + 1] = -sn[i] * h[i] + cs[i] * h[i + 1];
            h[i] = t;
        }
        const double den = hypot(h[j], h[j + 1]);
        if (!(den > 0.0) || !isfinite(den) || !(nrm2 == nrm2)) {
            flag = 2.0;
            nrm2 = 1.0;
        } else {
            cs[j] = h[j] / den;
            sn[j] = h[j + 1] / den;
            h[j] = den;
            h[j + 1] = 0.0;
            g[j + 1] = -sn[j] * g[j];
            g[j] = cs[j] * g[j];
            res = fabs(g[j + 1]);
            if (!(nrm2 > 0.0)) nrm2 = 1.0;   // exact breakdown (w' = 0): the next basis vector is irrelevant, avoid 1/0
        }
    }
    st[0] = nrm2; st[1] = flag; st[2] = res;
    if (mirror) { mirror[0] = res; mirror[1] = flag; }
    if (seq) {
        __threadfence_system();
        *seq = seq_val;
    }
}
__global__ void k_givens(GmLayout L, int j, int has_ns, double inv_cnt, const double* __restrict__ red, int explicit_slot,
                         double* __restrict__ gm, double* mirror, volatile int64_t* seq, int64_t seq_val) {
    if (threadIdx.x != 0) return;
    givens_body(L, j, has_ns, inv_cnt, red, explicit_slot, gm, mirror, seq, seq_val);
}

__device__ __forceinline__ void proj_norm_body(double s, double ww, double* __restrict__ red, int slot_out, double inv_count, double cancel, double* mirror,
                                               volatile int64_t* seq, int64_t seq_val) {
    const double nrm2 = ww - s * s * inv_count;
    const bool bad = !(nrm2 > cancel * ww) && ww > 0.0;
    red[slot_out] = bad ? ww : nrm2;
    red[slot_out + 1] = bad ? 1.0 : 0.0;
    if (mirror) { mirror[slot_out] = red[slot_out]; mirror[slot_out + 1] = red[slot_out + 1]; }
    if (seq) {
        __threadfence_system();
        *seq = seq_val;
    }
}
// ONE GPU (no all-reduce between the partial sums and their use): the second reduction stage and the scalar bookkeeping in ONE
// single-block kernel instead of k_reduce_partials + k_givens (or + k_proj_norm) -- one launch at the ~4.5 us floor less per GMRES
// iteration and per norm.  One wave per row, 16 partial sums per lane, fixed order: deterministic.
//   mode 1: Givens step of iteration j on red[slot0 .. slot0 + nred)      mode 2: gauge-projected norm from {s, w.w}
__global__ void __launch_bounds__(NT)
k_reduce_fin(int mode, int nb, int nred, const double* __restrict__ partial, double* __restrict__ red, int slot0, GmLayout L, int j, int has_ns,
             double inv_cnt, double* __restrict__ gm, int slot_out, double cancel, double* mirror, volatile int64_t* seq, int64_t seq_val) {
    __shared__ double s_red[RED_SLOTS];
    static_assert(RED_BLOCKS == 1024, "k_reduce_fin: 16 partial sums per lane");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int row = wave; row < nred; row += NT / 64) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int b = lane + 64 * k;
            v[k] = b < nb ? partial[(size_t)row * RED_BLOCKS + b] : 0.0;
        }
        double a = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += v[k];
        a = wave_sum(a);
        if (lane == 0) s_red[row] = a;
    }
    __syncthreads();
    if (threadIdx.x < nred) red[slot0 + threadIdx.x] = s_red[threadIdx.x];
    if (threadIdx.x == 0) {
        if (mode == 1) givens_body(L, j, has_ns, inv_cnt, s_red, -1, gm, mirror, seq, seq_val);
        else proj_norm_body(s_red[0], s_red[1], red, slot_out, inv_cnt, cancel, mirror, seq, seq_val);
    }
}
static bool fin_ok(const knp_ctx* ctx);

// (Measured and dropped in round 3: finishing the reduction inside k_multi_dot -- last block done, device-scope atomics, then the
// Givens step in that block -- costs 14 us per launch on this part: the 8 XCDs have separate L2s, so every device-scope round
// trip (write-through stores, the counter, the loads of the partial sums) goes to the memory side at 2-3 us each, more than the
// two ~4.7 us launches it replaces (512^2: 0.649 vs 0.621 ms per step).  Kernel boundaries are the cheaper synchronisation here.)
// partial[(i)*RED_BLOCKS + blk] = sum over this block's elements of V_i . w, i = i0 .. i0+G-1 (< m)
// NS: additionally accumulate the sum of the potential entries of w into row `m` of partial -- the coefficient
// of w along the (unnormalised) null-space vector, so that the gauge projection rides on the same reduction.
// WW: also accumulate w.w into the row after those (row m + NS): with it the norm of the orthogonalised vector follows from
// the same reduction (Pythagoras), and one all-reduce per GMRES iteration is enough.
template <int G, bool NS, bool WW>
__global__ void __launch_bounds__(NT)
k_multi_dot(int n, int64_t ldv, int i0, int m, const double* __restrict__ V, const double* __restrict__ w,
            double* __restrict__ partial) {
    __shared__ double sm[NT / 64];
    double acc[G];
    double ans = 0.0, aww = 0.0;
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.0;
    // n is a multiple of 4 (4 unknowns per node): two entries per trip as one 16-byte load per vector
    const int n2 = n >> 1;
    for (int e2 = blockIdx.x * NT + threadIdx.x; e2 < n2; e2 += gridDim.x * NT) {
        const double2 we = *reinterpret_cast<const double2*>(w + 2 * (size_t)e2);
        if (NS && (e2 & 1)) ans += we.y;          // entry 2*e2+1 is a potential iff e2 is odd
        if (WW) aww += we.x * we.x + we.y * we.y;
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (i0 + g < m) {
                const double2 v = *reinterpret_cast<const double2*>(V + (int64_t)(i0 + g) * ldv + 2 * (size_t)e2);
                acc[g] += v.x * we.x + v.y * we.y;
            }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (i0 + g >= m) break;      // (uniform) the short cycles of a well preconditioned solve use 1-3 of the G accumulators
        double r = block_sum(acc[g], sm);
        if (threadIdx.x == 0) partial[(size_t)(i0 + g) * RED_BLOCKS + blockIdx.x] = r;
    }
    if (NS) {
        double r = block_sum(ans, sm);
        if (threadIdx.x == 0) partial[(size_t)m * RED_BLOCKS + blockIdx.x] = r;
    }
    if (WW) {
        double r = block_sum(aww, sm);
        if (threadIdx.x == 0) partial[(size_t)(m + (NS ? 1 : 0)) * RED_BLOCKS + blockIdx.x] = r;
    }
}

// out[slot0 + i] = sum_b partial[i*RED_BLOCKS + b], one block per i
__global__ void __launch_bounds__(NT) k_reduce_partials(int nb, const double* __restrict__ partial,
                                                        double* __restrict__ out, int slot0, double* host_mirror,
                                                        volatile int64_t* seq = nullptr, int64_t seq_val = 0) {
    __shared__ double sm[NT / 64];
    const int i = blockIdx.x;
    double a = 0.0;
    for (int b = threadIdx.x; b < nb; b += NT) a += partial[(size_t)i * RED_BLOCKS + b];
    a = block_sum(a, sm);
    if (threadIdx.x == 0) {
        out[slot0 + i] = a;
        if (host_mirror) host_mirror[slot0 + i] = a;   // pinned, device-visible: no copy kernel for the read-back
        if (seq) {                                     // single-block launches only: publish "everything before me is done"
            __threadfence_system();
            *seq = seq_val;
        }
    }
}

// partial[blk] = sum a.b
__global__ void __launch_bounds__(NT) k_dot(int n, const double* __restrict__ a, const double* __restrict__ b,
                                            double* __restrict__ partial) {
    __shared__ double sm[NT / 64];
    double acc = 0.0;
    for (int e = blockIdx.x * NT + threadIdx.x; e < n; e += gridDim.x * NT) acc += a[e] * b[e];
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// out = in / sqrt(*nrm2)
__global__ void __launch_bounds__(NT) k_scale_rsqrt(int n, const double* __restrict__ in, const double* __restrict__ nrm2,
                                                    double* __restrict__ out) {
    const double inv = 1.0 / sqrt(*nrm2);
    for (int e = blockIdx.x * NT + threadIdx.x; e < n; e += gridDim.x * NT) out[e] = in[e] * inv;
}
// out = (in - mean on the potential entries) / sqrt(*nrm2), mean = *phi_sum * inv_count: the gauge projection and the normalisation
// of the initial Krylov vector in one pass (pc_apply_norm left the vector unprojected)
__global__ void __launch_bounds__(NT) k_scale_rsqrt_proj(int n, const double* __restrict__ in, const double* __restrict__ nrm2,
                                                         const double* __restrict__ phi_sum, double inv_count, double* __restrict__ out,
                                                         double* __restrict__ gm_g = nullptr, int m = 0) {
    if (gm_g && blockIdx.x == 0)   // the start of a GMRES cycle: g = (beta, 0, ..., 0) (was k_gm_init)
        for (int i = threadIdx.x; i <= m; i += NT) gm_g[i] = (i == 0) ? sqrt(*nrm2) : 0.0;
    const double inv = 1.0 / sqrt(*nrm2);
    const double mean = (*phi_sum) * inv_count;
    for (int e = blockIdx.x * NT + threadIdx.x; e < n; e += gridDim.x * NT) out[e] = ((e & 3) == 3 ? in[e] - mean : in[e]) * inv;
}
// red[out] = |z - ns (ns.z)|^2 = z.z - s^2/cnt from the reduced {s = sum of the potential entries, z.z} (Pythagoras); flag 1 when
// that difference has lost half of its digits (the caller then projects explicitly and reduces again, cf. GM_CANCEL)
__global__ void k_proj_norm(double* __restrict__ red, int slot_s, int slot_out, double inv_count, double cancel, double* mirror,
                            volatile int64_t* seq, int64_t seq_val) {
    if (threadIdx.x != 0) return;
    proj_norm_body(red[slot_s], red[slot_s + 1], red, slot_out, inv_count, cancel, mirror, seq, seq_val);
}
// x += sum_i y[i] V_i
__global__ void __launch_bounds__(NT) k_lincomb(int n, int64_t ldv, int m, const double* __restrict__ V,
                                                const double* __restrict__ y, double* __restrict__ x) {
    for (int e = blockIdx.x * NT + threadIdx.x; e < n; e += gridDim.x * NT) {
        double xe = x[e];
        for (int i = 0; i < m; ++i) xe += y[i] * V[(int64_t)i * ldv + e];
        x[e] = xe;
    }
}

__global__ void k_gm_init(GmLayout L, double* __restrict__ gm, const double* __restrict__ beta2) {
    const int t = threadIdx.x;
    for (int i = t; i <= L.m; i += blockDim.x) gm[L.g() + i] = (i == 0) ? sqrt(*beta2) : 0.0;
}

// column j of the Hessenberg matrix from the reduced values red = {h_0..h_j, [ns sum], w.w}: norm of the orthogonalised vector by
// Pythagoras (or, explicit_slot >= 0, the explicitly reduced one), previous Givens rotations, the new rotation, the residual
// estimate |g_{j+1}|; published to pinned host memory together with the sequence word the host spins on.
// v_{j+1} = (w - sum_i h[i] V_i - gauge part) / ||.||  in one pass (state: {||w'||^2, flag}); with flag != 0 the vector is left
// unnormalised (the explicit-norm fallback normalises it afterwards)
__global__ void __launch_bounds__(NT)
k_update_scale(int n, int64_t ldv, int m, const double* __restrict__ V, const double* __restrict__ h, const double* __restrict__ w,
               const double* __restrict__ st, double ns_scale, double* __restrict__ out) {
    const double inv = st[1] == 0.0 ? 1.0 / sqrt(st[0]) : 1.0;
    const double mean = ns_scale != 0.0 ? h[m] * ns_scale : 0.0;
    const int n2 = n >> 1;
    for (int e2 = blockIdx.x * NT + threadIdx.x; e2 < n2; e2 += gridDim.x * NT) {
        double2 we = *reinterpret_cast<const double2*>(w + 2 * (size_t)e2);
        if (e2 & 1) we.y -= mean;
        for (int i = 0; i < m; ++i) {
            const double2 v = *reinterpret_cast<const double2*>(V + (int64_t)i * ldv + 2 * (size_t)e2);
            const double hi = h[i];
            we.x -= hi * v.x;
            we.y -= hi * v.y;
        }
        *reinterpret_cast<double2*>(out + 2 * (size_t)e2) = make_double2(we.x * inv, we.y * inv);
    }
}
__global__ void __launch_bounds__(NT) k_scale_inplace_rsqrt(int n, const double* __restrict__ nrm2, double* __restrict__ v) {
    const double inv = 1.0 / sqrt(*nrm2);
    for (int e = blockIdx.x * NT + threadIdx.x; e < n; e += gridDim.x * NT) v[e] *= inv;
}
// y = H(0:jd,0:jd)^-1 g(0:jd) (upper triangular after the rotations)
__global__ void k_gm_solve_y(GmLayout L, int jd, double* __restrict__ gm) {
    if (threadIdx.x != 0) return;
    double* y = gm + L.y();
    const double* g = gm + L.g();
    for (int i = jd - 1; i >= 0; --i) {
        double s = g[i];
        for (int k = i + 1; k < jd; ++k) s -= gm[L.H(i, k)] * y[k];
        y[i] = s / gm[L.H(i, i)];
    }
}

// short cycles (jd <= 8, the usual case with a good preconditioner): every block solves the small triangular system itself from
// a copy in LDS and applies x += V y in the same kernel -- no separate k_gm_solve_y launch; block 0 also stores y
__global__ void __launch_bounds__(NT) k_lincomb_solve(GmLayout L, int jd, double* __restrict__ gm, int n, int64_t ldv,
                                                      const double* __restrict__ V, double* __restrict__ x) {
    __shared__ double sH[8 * 8], sg[8], sy[8];
    if (threadIdx.x < jd * jd) sH[threadIdx.x] = gm[L.H(threadIdx.x % jd, threadIdx.x / jd)];   // sH[i + jd*k] = H(i, k)
    if (threadIdx.x < jd) sg[threadIdx.x] = gm[L.g() + threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = jd - 1; i >= 0; --i) {
            double s = sg[i];
            for (int k = i + 1; k < jd; ++k) s -= sH[i + jd * k] * sy[k];
            sy[i] = s / sH[i + jd * i];
        }
        if (blockIdx.x == 0)
            for (int i = 0; i < jd; ++i) gm[L.y() + i] = sy[i];
    }
    __syncthreads();
    for (int e = blockIdx.x * NT + threadIdx.x; e < n; e += gridDim.x * NT) {
        double xe = x[e];
        for (int i = 0; i < jd; ++i) xe += sy[i] * V[(int64_t)i * ldv + e];
        x[e] = xe;
    }
}

__global__ void __launch_bounds__(NT) k_axpy(int n, double a, const double* __restrict__ x, double* __restrict__ y) {
    for (int e = blockIdx.x * NT + threadIdx.x; e < n; e += gridDim.x * NT) y[e] += a * x[e];
}
__global__ void __launch_bounds__(NT) k_fill(int n, double a, double* __restrict__ y) {
    for (int e = blockIdx.x * NT + threadIdx.x; e < n; e += gridDim.x * NT) y[e] = a;
}
// y = d .* x
__global__ void __launch_bounds__(NT) k_diag_scale(int n, double a, const double* __restrict__ d,
                                                   const double* __restrict__ x, double* __restrict__ y) {
    for (int e = blockIdx.x * NT + threadIdx.x; e < n; e += gridDim.x * NT) y[e] = a * d[e] * x[e];
}

// ------------------------------------------------------------------------------------------
// K8: gauge projection
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT) k_phi_sum(int n_nodes, const double* __restrict__ z, double* __restrict__ partial) {
    __shared__ double sm[NT / 64];
    double acc = 0.0;
    for (int n = blockIdx.x * NT + threadIdx.x; n < n_nodes; n += gridDim.x * NT) acc += z[(size_t)4 * n + 3];
    acc = block_sum(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(NT) k_phi_sub(int n_nodes, const double* __restrict__ sum, double inv_count,
                                                double* __restrict__ z) {
    const double mean = (*sum) * inv_count;
    for (int n = blockIdx.x * NT + threadIdx.x; n < n_nodes; n += gridDim.x * NT) z[(size_t)4 * n + 3] -= mean;
}
__global__ void __launch_bounds__(NT) k_fill_phi(int n_nodes, double val, double* __restrict__ z) {
    for (int n = blockIdx.x * NT + threadIdx.x; n < n_nodes; n += gridDim.x * NT) {
        z[(size_t)4 * n + 0] = 0.0;
        z[(size_t)4 * n + 1] = 0.0;
        z[(size_t)4 * n + 2] = 0.0;
        z[(size_t)4 * n + 3] = val;
    }
}


// ------------------------------------------------------------------------------------------
// Deflation of partition-cut near-null modes (constants of a potential block on one connected
// component): s_c = sum of the potential entries of r over nodes of mode c ; z += Z Einv s
// ------------------------------------------------------------------------------------------
template <int G>
__global__ void __launch_bounds__(NT)
k_defl_sums(int n_nodes, int c0, int m, const int32_t* __restrict__ node_mode, const double* __restrict__ r,
            double* __restrict__ partial) {
    __shared__ double sm[NT / 64];
    double acc[G];
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = 0.0;
    for (int n = blockIdx.x * NT + threadIdx.x; n < n_nodes; n += gridDim.x * NT) {
        const int md = node_mode[n];
        const double v = r[(size_t)4 * n + 3];
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] += (md == c0 + g) ? v : 0.0;
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        double t = block_sum(acc[g], sm);
        if (threadIdx.x == 0 && c0 + g < m) partial[(size_t)(c0 + g) * RED_BLOCKS + blockIdx.x] = t;
    }
}
__global__ void __launch_bounds__(NT)
k_defl_add(int n_nodes, int m, const int32_t* __restrict__ node_mode, const double* __restrict__ einv,
           const double* __restrict__ s, double* __restrict__ z) {
    const int n = blockIdx.x * NT + threadIdx.x;
    if (n >= n_nodes) return;
    const int md = node_mode[n];
    if (md < 0) return;
    double y = 0.0;
    for (int c = 0; c < m; ++c) y += einv[md * m + c] * s[c];
    z[(size_t)4 * n + 3] += y;
}

// ------------------------------------------------------------------------------------------
// Dirichlet conditions (MMS path; KNPEMIx_problem.py:106-134): the row of a constrained DoF becomes the
// identity row (the right-hand side entry is set by the caller).  DOLFINx additionally zeroes the column and
// lifts the right-hand side; the solution is the same.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT)
k_dirichlet_rows_A(int n_bc, const int32_t* __restrict__ bc_dofs, const int32_t* __restrict__ pair_ptr,
                   const int32_t* __restrict__ pair_col, const int32_t* __restrict__ node_gv, const uint8_t* __restrict__ node_side,
                   const int32_t* __restrict__ gptr, double* __restrict__ ac, double* __restrict__ at, double* __restrict__ ax) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= n_bc) return;
    const int row = bc_dofs[i], node = row >> 2, f = row & 3;
    for (int p = pair_ptr[node]; p < pair_ptr[node + 1]; ++p) {
        const double diag = pair_col[p] == node ? 1.0 : 0.0;
        if (f < 3) {
            ac[6 * (size_t)p + f] = diag;
            at[4 * (size_t)p + f] = 0.0;
        } else {
            ac[6 * (size_t)p + 3] = 0.0; ac[6 * (size_t)p + 4] = 0.0; ac[6 * (size_t)p + 5] = 0.0;
            at[4 * (size_t)p + 3] = diag;
        }
    }
    const int A = node_gv[node];
    if (A >= 0)
        for (int s = gptr[A]; s < gptr[A + 1]; ++s) ax[8 * (size_t)s + 4 * node_side[node] + f] = 0.0;
}
__global__ void __launch_bounds__(NT)
k_dirichlet_rows_P(int n_bc, const int32_t* __restrict__ bc_dofs, const int32_t* __restrict__ pair_ptr,
                   const int32_t* __restrict__ pair_col, double* __restrict__ pv) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= n_bc) return;
    const int row = bc_dofs[i], node = row >> 2, f = row & 3;
    for (int p = pair_ptr[node]; p < pair_ptr[node + 1]; ++p) pv[(size_t)4 * p + f] = (pair_col[p] == node) ? 1.0 : 0.0;
}

// ------------------------------------------------------------------------------------------
// K7: vertex-block Jacobi built from A.  blk[n*16 + {0..2: d_j, 3..5: u_j, 6..8: v_j, 9: s,
//                                         10..12: ux_j (k rows -> other side's phi), 13: sx}]
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT)
k_vbj_extract(int n_nodes_owned, const int32_t* __restrict__ pair_ptr, const int32_t* __restrict__ pair_col,
              const double* __restrict__ ac, const double* __restrict__ at, const double* __restrict__ ax,
              const int32_t* __restrict__ node_gv, const uint8_t* __restrict__ node_side,
              const int32_t* __restrict__ gdiag, double* __restrict__ blk) {
    const int n = blockIdx.x * NT + threadIdx.x;
    if (n >= n_nodes_owned) return;
    const int p0 = pair_ptr[n], deg = pair_ptr[n + 1] - p0;
    size_t ps = p0;
    for (int q = 0; q < deg; ++q)
        if (pair_col[p0 + q] == n) { ps = (size_t)p0 + q; break; }
    double* o = blk + (size_t)16 * n;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        o[j] = ac[6 * ps + j];
        o[3 + j] = at[4 * ps + j];
        o[6 + j] = ac[6 * ps + 3 + j];
    }
    o[9] = at[4 * ps + 3];
    const int A = node_gv[n];
    if (A >= 0) {
        const size_t s = (size_t)gdiag[A];
        const int sd = node_side[n];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[10 + j] = ax[8 * s + 4 * sd + j];
    } else {
        o[10] = o[11] = o[12] = o[13] = 0.0;
    }
}

__global__ void __launch_bounds__(NT)
k_vbj_apply(int n_nodes_owned, const uint8_t* __restrict__ node_side, const int32_t* __restrict__ node_gv,
            const int32_t* __restrict__ gv_node_e, const double* __restrict__ blk, const double* __restrict__ r,
            double* __restrict__ z) {
    const int n = blockIdx.x * NT + threadIdx.x;
    if (n >= n_nodes_owned) return;
    const int A = node_gv[n];
    const double* B = blk + (size_t)16 * n;
    if (A < 0) {
        double rp = r[(size_t)4 * n + 3], S = B[9];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            rp -= B[6 + j] * r[(size_t)4 * n + j] / B[j];
            S -= B[6 + j] * B[3 + j] / B[j];
        }
        const double yp = rp / S;
#pragma unroll
        for (int j = 0; j < 3; ++j) z[(size_t)4 * n + j] = (r[(size_t)4 * n + j] - B[3 + j] * yp) / B[j];
        z[(size_t)4 * n + 3] = yp;
        return;
    }
    if (node_side[n] != 0) return;  // the intra node of a membrane vertex solves the coupled 8x8
    const int m = gv_node_e[A];
    const double* C = blk + (size_t)16 * m;
    double a11 = B[9], a12 = B[13], a21 = C[13], a22 = C[9];
    double r1 = r[(size_t)4 * n + 3], r2 = r[(size_t)4 * m + 3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double vi = B[6 + j] / B[j], ve = C[6 + j] / C[j];
        a11 -= vi * B[3 + j];
        a12 -= vi * B[10 + j];
        r1 -= vi * r[(size_t)4 * n + j];
        a22 -= ve * C[3 + j];
        a21 -= ve * C[10 + j];
        r2 -= ve * r[(size_t)4 * m + j];
    }
    const double det = a11 * a22 - a12 * a21;
    const double pi = (a22 * r1 - a12 * r2) / det;
    const double pe = (a11 * r2 - a21 * r1) / det;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        z[(size_t)4 * n + j] = (r[(size_t)4 * n + j] - B[3 + j] * pi - B[10 + j] * pe) / B[j];
        z[(size_t)4 * m + j] = (r[(size_t)4 * m + j] - C[3 + j] * pe - C[10 + j] * pi) / C[j];
    }
    z[(size_t)4 * n + 3] = pi;
    z[(size_t)4 * m + 3] = pe;
}

// ------------------------------------------------------------------------------------------
// AMG pieces
// ------------------------------------------------------------------------------------------
// One Chebyshev step with the solution update fused (ping-pong buffers xin -> xout):
//   d = c1*d + c2 * Dinv*(b - A xin) ;  xout = xin + d          (L lanes per row)
template <int L, typename VT>
__global__ void __launch_bounds__(NT)
k_cheby_step(int n_rows, const int32_t* __restrict__ rp, const int32_t* __restrict__ ci, const VT* __restrict__ v,
             const double* __restrict__ dinv, const double* __restrict__ b, const double* __restrict__ xin,
             double c1, double c2, double* __restrict__ d, double* __restrict__ xout) {
    const int gid = blockIdx.x * NT + threadIdx.x;
    const int row = gid / L;
    const int lane = threadIdx.x & (L - 1);
    double s = 0.0;
    if (row < n_rows) s = row_dot4<L, VT>(rp[row], rp[row + 1], lane, ci, v, xin);
#pragma unroll
    for (int o = L >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, L);
    if (lane == 0 && row < n_rows) {
        const double dn = (c1 != 0.0 ? c1 * d[row] : 0.0) + c2 * dinv[row] * (b[row] - s);
        d[row] = dn;
        xout[row] = xin[row] + dn;
    }
}
template <typename VT>
static void launch_cheby_t(hipStream_t st, int lanes, int n_rows, const int32_t* rp, const int32_t* ci, const VT* v,
                           const double* dinv, const double* b, const double* xin, double c1, double c2, double* d, double* xout) {
    if (n_rows <= 0) return;
    switch (lanes) {
        case 2: hipLaunchKernelGGL((k_cheby_step<2, VT>), dim3(nblocks((int64_t)n_rows * 2)), dim3(NT), 0, st, n_rows, rp, ci, v, dinv, b, xin, c1, c2, d, xout); break;
        case 4: hipLaunchKernelGGL((k_cheby_step<4, VT>), dim3(nblocks((int64_t)n_rows * 4)), dim3(NT), 0, st, n_rows, rp, ci, v, dinv, b, xin, c1, c2, d, xout); break;
        case 8: hipLaunchKernelGGL((k_cheby_step<8, VT>), dim3(nblocks((int64_t)n_rows * 8)), dim3(NT), 0, st, n_rows, rp, ci, v, dinv, b, xin, c1, c2, d, xout); break;
        case 16: hipLaunchKernelGGL((k_cheby_step<16, VT>), dim3(nblocks((int64_t)n_rows * 16)), dim3(NT), 0, st, n_rows, rp, ci, v, dinv, b, xin, c1, c2, d, xout); break;
        case 32: hipLaunchKernelGGL((k_cheby_step<32, VT>), dim3(nblocks((int64_t)n_rows * 32)), dim3(NT), 0, st, n_rows, rp, ci, v, dinv, b, xin, c1, c2, d, xout); break;
        default: hipLaunchKernelGGL((k_cheby_step<64, VT>), dim3(nblocks((int64_t)n_rows * 64)), dim3(NT), 0, st, n_rows, rp, ci, v, dinv, b, xin, c1, c2, d, xout); break;
    }
}
static void launch_cheby(hipStream_t st, int lanes, int n_rows, const int32_t* rp, const int32_t* ci, const double* v, const float* vf,
                         const double* dinv, const double* b, const double* xin, double c1, double c2, double* d, double* xout) {
    if (vf) launch_cheby_t<float>(st, lanes, n_rows, rp, ci, vf, dinv, b, xin, c1, c2, d, xout);
    else launch_cheby_t<double>(st, lanes, n_rows, rp, ci, v, dinv, b, xin, c1, c2, d, xout);
}
// first Chebyshev step from a zero guess: d = x = c * Dinv * b
__global__ void __launch_bounds__(NT) k_cheby_first(int n, double c, const double* __restrict__ dinv,
                                                    const double* __restrict__ b, double* __restrict__ d,
                                                    double* __restrict__ x) {
    for (int e = blockIdx.x * NT + threadIdx.x; e < n; e += gridDim.x * NT) {
        const double t = c * dinv[e] * b[e];
        d[e] = t;
        x[e] = t;
    }
}
// y = M x, dense row-major n x n.  WPR waves per row (a 2500-row coarse inverse with one wave per row is 2.4 waves per SIMD:
// the row streams at one wave's latency; four waves per row keep ~10 waves per SIMD in flight), 16-byte loads, two trips in
// flight per lane; rows are combined through LDS.
template <typename VT, int WPR>
__global__ void __launch_bounds__(NT) k_dense_matvec(int n, const VT* __restrict__ M, const double* __restrict__ x,
                                                     double* __restrict__ y) {
    constexpr int RPB = (NT / 64) / WPR;          // rows per block
    __shared__ double sm[NT / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * RPB + wave / WPR;
    const int part = wave % WPR;
    const int t = part * 64 + lane;               // position within the row's WPR*64 lanes
    constexpr int TPR = WPR * 64;
    double s = 0.0;
    if (row < n) {
        const VT* __restrict__ m = M + (size_t)row * n;
        if (sizeof(VT) == 8 && (n & 1) == 0) {
            const double2* m2 = reinterpret_cast<const double2*>(m);
            const double2* x2 = reinterpret_cast<const double2*>(x);
            const int n2 = n >> 1;
            double s0 = 0.0, s1 = 0.0;
            int k = t;
            for (; k + TPR < n2; k += 2 * TPR) {
                const double2 a = m2[k], b = m2[k + TPR];
                const double2 xa = x2[k], xb = x2[k + TPR];
                s0 += a.x * xa.x + a.y * xa.y;
                s1 += b.x * xb.x + b.y * xb.y;
            }
            if (k < n2) {
                const double2 a = m2[k];
                const double2 xa = x2[k];
                s0 += a.x * xa.x + a.y * xa.y;
            }
            s = s0 + s1;
        } else if (sizeof(VT) == 4 && (n & 3) == 0) {
            const float4* m4 = reinterpret_cast<const float4*>(m);
            const double2* x2 = reinterpret_cast<const double2*>(x);
            const int n4 = n >> 2;
            double s0 = 0.0, s1 = 0.0;
            int k = t;
            for (; k + TPR < n4; k += 2 * TPR) {
                const float4 a = m4[k], b = m4[k + TPR];
                const double2 xa = x2[2 * k], xb = x2[2 * k + 1], xc = x2[2 * (k + TPR)], xd = x2[2 * (k + TPR) + 1];
                s0 += (double)a.x * xa.x + (double)a.y * xa.y + (double)a.z * xb.x + (double)a.w * xb.y;
                s1 += (double)b.x * xc.x + (double)b.y * xc.y + (double)b.z * xd.x + (double)b.w * xd.y;
            }
            if (k < n4) {
                const float4 a = m4[k];
                const double2 xa = x2[2 * k], xb = x2[2 * k + 1];
                s0 += (double)a.x * xa.x + (double)a.y * xa.y + (double)a.z * xb.x + (double)a.w * xb.y;
            }
            s = s0 + s1;
        } else {
            for (int k = t; k < n; k += TPR) s += (double)m[k] * x[k];
        }
    }
    s = wave_sum(s);
    if (WPR == 1) {
        if (lane == 0 && row < n) y[row] = s;
    } else {
        if (lane == 0) sm[wave] = s;
        __syncthreads();
        if (lane == 0 && part == 0 && row < n) {
            double r = 0.0;
#pragma unroll
            for (int q = 0; q < WPR; ++q) r += sm[wave + q];
            y[row] = r;
        }
    }
}
template <typename VT>
static void launch_dense_matvec(hipStream_t st, int n, const VT* M, const double* x, double* y) {
    if (n <= 0) return;
    if (n >= 512) hipLaunchKernelGGL((k_dense_matvec<VT, 4>), dim3(n), dim3(NT), 0, st, n, M, x, y);                       // one row per block
    else hipLaunchKernelGGL((k_dense_matvec<VT, 1>), dim3(nblocks((int64_t)n * 64)), dim3(NT), 0, st, n, M, x, y);       // one wave per row
}

// ------------------------------------------------------------------------------------------
// K9: Hodgkin-Huxley gating update (KNPEMIx_ionic_model.py:605-671)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(NT)
k_hh_update(int count, const double* __restrict__ phi_m, double* __restrict__ gn, double* __restrict__ gm,
            double* __restrict__ gh, double dt, double phi_rest, int rush_larsen, int substeps) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= count) return;
    const double V = 1000.0 * (phi_m[i] - phi_rest);
    const double an = 0.01e3 * (10.0 - V) / (exp((10.0 - V) / 10.0) - 1.0);
    const double bn = 0.125e3 * exp(-V / 80.0);
    const double am = 0.1e3 * (25.0 - V) / (exp((25.0 - V) / 10.0) - 1.0);
    const double bm = 4.0e3 * exp(-V / 18.0);
    const double ah = 0.07e3 * exp(-V / 20.0);
    const double bh = 1.0e3 / (exp((30.0 - V) / 10.0) + 1.0);
    double n = gn[i], m = gm[i], h = gh[i];
    if (rush_larsen) {
        // `substeps` frozen-coefficient exponential steps of dt/substeps == one exact step of dt
        const double tn = an + bn, tm = am + bm, th = ah + bh;
        const double ninf = an / tn, minf = am / tm, hinf = ah / th;
        n = ninf + (n - ninf) * exp(-dt * tn);
        m = minf + (m - minf) * exp(-dt * tm);
        h = hinf + (h - hinf) * exp(-dt * th);
    } else {
        const double dto = dt / substeps;
        for (int s = 0; s < substeps; ++s) {
            n += dto * an * (1.0 - n) - dto * bn * n;
            m += dto * am * (1.0 - m) - dto * bm * m;
            h += dto * ah * (1.0 - h) - dto * bh * h;
        }
    }
    gn[i] = n;
    gm[i] = m;
    gh[i] = h;
}

// ------------------------------------------------------------------------------------------
// K10: pack / unpack
// ------------------------------------------------------------------------------------------
struct OutPtrs {
    double* ki[3];
    double* ke[3];
    double *phi_i, *phi_e, *phi_m;
};
__global__ void __launch_bounds__(NT)
k_pack(int n_nodes, const int32_t* __restrict__ node_vertex, const uint8_t* __restrict__ node_side, OutPtrs f,
       double* __restrict__ x) {
    const int n = blockIdx.x * NT + threadIdx.x;
    if (n >= n_nodes) return;
    const int v = node_vertex[n];
    const int s = node_side[n];
#pragma unroll
    for (int j = 0; j < 3; ++j) x[(size_t)4 * n + j] = (s ? f.ke[j] : f.ki[j])[v];
    x[(size_t)4 * n + 3] = (s ? f.phi_e : f.phi_i)[v];
}
__global__ void __launch_bounds__(NT)
k_unpack(int n_v, const int32_t* __restrict__ node_i, const int32_t* __restrict__ node_e, const double* __restrict__ x,
         OutPtrs f) {
    const int v = blockIdx.x * NT + threadIdx.x;
    if (v >= n_v) return;
    const int ni = node_i[v], ne = node_e[v];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        f.ki[j][v] = ni >= 0 ? x[(size_t)4 * ni + j] : 0.0;
        f.ke[j][v] = ne >= 0 ? x[(size_t)4 * ne + j] : 0.0;
    }
    const double pi = ni >= 0 ? x[(size_t)4 * ni + 3] : 0.0;
    const double pe = ne >= 0 ? x[(size_t)4 * ne + 3] : 0.0;
    f.phi_i[v] = pi;
    f.phi_e[v] = pe;
    f.phi_m[v] = pi - pe;
}

// partial[blk] = max |v| over the block's share
__global__ void __launch_bounds__(NT) k_absmax(int64_t n, const double* __restrict__ v, double* __restrict__ partial) {
    __shared__ double sm[NT / 64];
    double a = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * NT + threadIdx.x; e < n; e += (int64_t)gridDim.x * NT) a = fmax(a, fabs(v[e]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a = fmax(a, __shfl_xor(a, o, 64));
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) sm[w] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < NT / 64; ++k) a = fmax(a, sm[k]);
        partial[blockIdx.x] = a;
    }
}

// L2 norms: partial[blk] (intra), partial[RED_BLOCKS + blk] (extra)
template <int DIM>
__global__ void __launch_bounds__(NT)
k_l2(int n_c, const int32_t* __restrict__ cells, const uint8_t* __restrict__ side, const double* __restrict__ coords,
     const double* __restrict__ phi_i, const double* __restrict__ phi_e, double* __restrict__ partial) {
    __shared__ double sm[NT / 64];
    double acc[2] = {0.0, 0.0};
    for (int c = blockIdx.x * NT + threadIdx.x; c < n_c; c += gridDim.x * NT) {
        int v[DIM + 1];
#pragma unroll
        for (int a = 0; a <= DIM; ++a) v[a] = cells[(size_t)c * (DIM + 1) + a];
        double e[DIM][DIM];
#pragma unroll
        for (int a = 0; a < DIM; ++a)
#pragma unroll
            for (int d = 0; d < DIM; ++d) e[a][d] = coords[(size_t)v[a + 1] * DIM + d] - coords[(size_t)v[0] * DIM + d];
        double det;
        if (DIM == 2) {
            det = e[0][0] * e[1][1] - e[0][1] * e[1][0];
        } else {
            det = e[0][0] * (e[1][1] * e[2][2] - e[1][2] * e[2][1]) - e[0][1] * (e[1][0] * e[2][2] - e[1][2] * e[2][0]) +
                  e[0][2] * (e[1][0] * e[2][1] - e[1][1] * e[2][0]);
        }
        const double vol = fabs(det) / (DIM == 2 ? 2.0 : 6.0);
        const int s = side[c];
        const double* u = s ? phi_e : phi_i;
        double su = 0.0, suu = 0.0;
#pragma unroll
        for (int a = 0; a <= DIM; ++a) {
            const double ua = u[v[a]];
            su += ua;
            suu += ua * ua;
        }
        // u^T M u with M = vol (1 + delta)/((d+1)(d+2))
        acc[s] += vol * (su * su + suu) / ((DIM + 1.0) * (DIM + 2.0));
    }
    double r0 = block_sum(acc[0], sm);
    double r1 = block_sum(acc[1], sm);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = r0;
        partial[RED_BLOCKS + blockIdx.x] = r1;
    }
}

// ==========================================================================================
// host side
// ==========================================================================================
template <typename T>
static int dev_upload(knp_ctx* ctx, T** dst, const std::vector<T>& src) {
    *dst = nullptr;
    size_t bytes = std::max<size_t>(src.size(), 1) * sizeof(T);
    HIPCHK(hipMalloc((void**)dst, bytes));
    if (!src.empty()) HIPCHK(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return KNP_OK;
}
template <typename T>
static int dev_upload_raw(knp_ctx* ctx, T** dst, const T* src, size_t n) {
    *dst = nullptr;
    HIPCHK(hipMalloc((void**)dst, std::max<size_t>(n, 1) * sizeof(T)));
    if (n) HIPCHK(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    return KNP_OK;
}
// host-side passes over whole operators of the AMG upload, in parallel (10^8 entries on the finest levels of a 10^7-unknown problem)
static bool cols_in_range(const int32_t* ci, int64_t nnz, int64_t hi) {
    int64_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad) num_threads(knp_host_threads())
    for (int64_t k = 0; k < nnz; ++k) bad += (ci[k] < 0 || ci[k] >= hi) ? 1 : 0;
    return bad == 0;
}
// Uninitialised host array: its pages are first touched by the parallel loop that fills it (a std::vector value-initialises -- and
// page-faults -- serially, which on arrays of 10^8 entries cost more than the fill itself)
template <typename T>
struct HostBuf {
    T* p = nullptr;
    size_t n = 0;
    explicit HostBuf(size_t n_) : p(static_cast<T*>(std::malloc(std::max<size_t>(n_, 1) * sizeof(T)))), n(n_) {}
    ~HostBuf() { std::free(p); }
    HostBuf(const HostBuf&) = delete;
    HostBuf& operator=(const HostBuf&) = delete;
    HostBuf(HostBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    T& operator[](size_t i) { return p[i]; }
    const T* data() const { return p; }
    size_t size() const { return n; }
};
template <typename T>
static int dev_upload(knp_ctx* ctx, T** dst, const HostBuf<T>& src) {
    if (!src.p) { ctx->err = "out of host memory"; return KNP_E_STATE; }
    return dev_upload_raw(ctx, dst, src.data(), src.size());
}
static HostBuf<float> to_float(const double* v, int64_t nnz) {
    HostBuf<float> out((size_t)std::max<int64_t>(nnz, 0));
    if (!out.p) return out;
#pragma omp parallel for schedule(static) num_threads(knp_host_threads())
    for (int64_t k = 0; k < nnz; ++k) out[(size_t)k] = (float)v[k];
    return out;
}
template <typename T>
static void dev_free(T*& p) {
    if (p) (void)hipFree((void*)p);
    p = nullptr;
}

static DevParams make_params(const knp_ctx* ctx) {
    DevParams P;
    P.dt = ctx->dt; P.F = ctx->F; P.C_M = ctx->C_M; P.psi = ctx->psi;
    for (int j = 0; j < 3; ++j) {
        P.z[j] = ctx->z[j]; P.Di[j] = ctx->Di[j]; P.De[j] = ctx->De[j];
        P.dz2i[j] = P.Di[j] * P.z[j] * P.z[j];
        P.dz2e[j] = P.De[j] * P.z[j] * P.z[j];
        P.rFz[j] = 1.0 / (P.F * P.z[j]);
        P.cmFz[j] = P.C_M * P.rFz[j];
    }
    P.rF = 1.0 / P.F;
    return P;
}
static FieldPtrs make_fields(const knp_fields* f) {
    FieldPtrs o;
    for (int j = 0; j < 3; ++j) { o.ki[j] = f->k_i[j]; o.ke[j] = f->k_e[j]; }
    o.phim = f->phi_m;
    for (int k = 0; k < KNP_MAX_AUX; ++k) o.aux[k] = f->aux[k];
    return o;
}

// timing events are recycled through a pool: creating them inside the timed region costs host time per launch
static hipEvent_t prof_event(knp_ctx* ctx) {
    if (!ctx->prof_pool.empty()) {
        hipEvent_t e = ctx->prof_pool.back();
        ctx->prof_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

// profiling scope: records a pair of events around a group of launches when enabled
struct ProfScope {
    knp_ctx* c;
    hipEvent_t a = nullptr, b = nullptr;
    int cls;
    ProfScope(knp_ctx* ctx, int cls_) : c(ctx), cls(cls_) {
        if ((c->prof_on >> cls) & 1) {
            a = prof_event(c);
            b = prof_event(c);
            if (!a || !b) { a = b = nullptr; return; }
            (void)hipEventRecord(a, c->stream);
        }
    }
    ~ProfScope() {
        if (a && b) {
            (void)hipEventRecord(b, c->stream);
            c->prof_recs.push_back({a, b, cls});
        }
    }
};

static int prof_collect(knp_ctx* ctx) {
    if (ctx->prof_recs.empty()) return KNP_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->stream2) HIPCHK(hipStreamSynchronize(ctx->stream2));
    if (ctx->stream_asm) HIPCHK(hipStreamSynchronize(ctx->stream_asm));
    for (auto& r : ctx->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            ctx->prof_ms[r.cls] += ms;
            ctx->prof_n[r.cls] += 1;
        }
        ctx->prof_pool.push_back(r.a);
        ctx->prof_pool.push_back(r.b);
    }
    ctx->prof_recs.clear();
    return KNP_OK;
}

// recycle the events of launches that have completed, without synchronising (called once per solve)
static void prof_collect_ready(knp_ctx* ctx) {
    if (ctx->prof_recs.size() < 64) return;
    size_t keep = 0;
    for (size_t i = 0; i < ctx->prof_recs.size(); ++i) {
        auto r = ctx->prof_recs[i];
        float ms = 0.f;
        if (hipEventQuery(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            ctx->prof_ms[r.cls] += ms;
            ctx->prof_n[r.cls] += 1;
            ctx->prof_pool.push_back(r.a);
            ctx->prof_pool.push_back(r.b);
        } else {
            ctx->prof_recs[keep++] = r;
        }
    }
    ctx->prof_recs.resize(keep);
    (void)hipGetLastError();   // hipEventQuery reports hipErrorNotReady through the sticky error
}

extern "C" {

const char* knp_last_error(const knp_ctx* ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

// Transposed contribution lists of k_assemble_nodes_tr (layout described there).  Skipped (the pair-major lists stay in use) when
// the zero padding would add more than half to the stored contributions -- meshes whose edges have very uneven cell counts.
static int build_transposed_contribs(knp_ctx* ctx) {
    const KnpHostGraph& g = ctx->g;
    const int no = g.n_nodes_owned;
    std::vector<int64_t> meta((size_t)std::max(no, 1), 0);
    int64_t total = 0;
    for (int n = 0; n < no; ++n) {
        const int p0 = g.pair_ptr[n], p1 = g.pair_ptr[n + 1];
        int selfq = 255, tmax = 0;
        for (int p = p0; p < p1; ++p) {
            if (g.pair_col[p] == n) { if (p - p0 < 255) selfq = p - p0; else return KNP_OK; }
            tmax = std::max(tmax, g.contrib_ptr[p + 1] - g.contrib_ptr[p]);
        }
        tmax = (tmax + 1) & ~1;
        if (tmax > 254) return KNP_OK;
        const int dod = (p1 - p0) - (selfq < 255 ? 1 : 0);
        meta[n] = total | ((int64_t)tmax << 48) | ((int64_t)selfq << 56);
        total += (int64_t)tmax * dod;
    }
    if (total >= (1LL << 47) || (double)total > 1.5 * (double)g.contrib_k.size() + 1024.0) return KNP_OK;
    std::vector<double> tk((size_t)std::max<int64_t>(total, 1), 0.0);
    std::vector<uint8_t> ts((size_t)std::max<int64_t>(total, 1), 0);
#pragma omp parallel for schedule(static) num_threads(knp_host_threads())
    for (int n = 0; n < no; ++n) {
        const int p0 = g.pair_ptr[n], p1 = g.pair_ptr[n + 1];
        const int64_t base = meta[n] & 0xffffffffffffLL;
        const int selfq = (int)((meta[n] >> 56) & 0xff);
        const int dod = (p1 - p0) - (selfq < 255 ? 1 : 0);
        for (int p = p0; p < p1; ++p) {
            const int q = p - p0;
            if (q == selfq) continue;
            const int j = q - (q > selfq ? 1 : 0);
            int t = 0;
            for (int c = g.contrib_ptr[p]; c < g.contrib_ptr[p + 1]; ++c, ++t) {
                tk[(size_t)(base + (int64_t)t * dod + j)] = g.contrib_k[c];
                ts[(size_t)(base + (int64_t)t * dod + j)] = g.contrib_slot[c];
            }
        }
    }
    KCHK(dev_upload(ctx, &ctx->d_tc_meta, meta));
    KCHK(dev_upload(ctx, &ctx->d_tc_k, tk));
    KCHK(dev_upload(ctx, &ctx->d_tc_slot, ts));
    ctx->n_tc = total;
    return KNP_OK;
}

// per (node, cell of its list) the local neighbour index of each of the cell's vertices, in the cell's vertex order (fused cell means)
static int build_fused_cell_means(knp_ctx* ctx, const knp_mesh_desc* mesh, int G0) {
    const KnpHostGraph& g = ctx->g;
    const char* ef = getenv("KNP_ASM_FUSED_MEANS");
    if (ef && atoi(ef) == 0) return KNP_OK;
    const int no = g.n_nodes_owned, nv1 = g.nv1;
    int dmax = 0;
    for (int n = 0; n < no; ++n) dmax = std::max(dmax, g.pair_ptr[n + 1] - g.pair_ptr[n]);
    const size_t lds = (size_t)(NT / G0) * (g.max_node_cells + dmax) * 3 * sizeof(double);
    if (dmax <= 0 || dmax > 255 || lds > 64 * 1024 || g.node_cell.empty()) return KNP_OK;
    std::vector<uint8_t> ncv((size_t)nv1 * g.node_cell.size());
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad) num_threads(knp_host_threads())
    for (int n = 0; n < no; ++n) {
        const int32_t* pc = g.pair_col.data() + g.pair_ptr[n];
        const int deg = g.pair_ptr[n + 1] - g.pair_ptr[n];
        for (int i = g.node_cell_ptr[n]; i < g.node_cell_ptr[n + 1]; ++i) {
            const int c = g.node_cell[i];
            const int32_t* nodes = mesh->cell_side[c] ? g.node_e.data() : g.node_i.data();
            for (int a = 0; a < nv1; ++a) {
                const int nb = nodes[mesh->cells[(size_t)c * nv1 + a]];
                const int32_t* it = std::lower_bound(pc, pc + deg, nb);
                if (it == pc + deg || *it != nb) { ++bad; ncv[(size_t)nv1 * i + a] = 0; }
                else ncv[(size_t)nv1 * i + a] = (uint8_t)(it - pc);
            }
        }
    }
    if (bad) return KNP_OK;   // a vertex of a node's cell that is not among its pairs: keep the separate cell-mean pass
    KCHK(dev_upload(ctx, &ctx->d_ncv, ncv));
    HIPCHK(hipMalloc((void**)&ctx->d_knod, (size_t)4 * std::max(g.n_nodes, 1) * sizeof(double)));
    HIPCHK(hipMemset(ctx->d_knod, 0, (size_t)4 * std::max(g.n_nodes, 1) * sizeof(double)));
    ctx->asm_dmax = dmax;
    ctx->n_node_cells = (int64_t)g.node_cell.size();
    return KNP_OK;
}

int knp_create(knp_ctx** out, const knp_mesh_desc* mesh) {
    if (!out) return KNP_E_ARG;
    *out = nullptr;
    knp_ctx* ctx = new (std::nothrow) knp_ctx();
    if (!ctx) return KNP_E_ALLOC;
    *out = ctx;  // returned even on failure so that knp_last_error can be read; caller destroys
    int rc = knp_build_graph(mesh, ctx->g);
    if (rc != KNP_OK) { ctx->err = ctx->g.error; return rc; }
    KnpHostGraph& g = ctx->g;
    HIPCHK(hipGetDevice(&ctx->device));
    ctx->max_prog = -1;
    for (int f = 0; f < g.n_g; ++f) {
        if (mesh->gamma_prog[f] < 0) { ctx->err = "negative gamma_prog id"; return KNP_E_MESH; }
        ctx->max_prog = std::max(ctx->max_prog, (int)mesh->gamma_prog[f]);
    }
    ctx->n_pairs = (int64_t)g.pair_col.size();
    ctx->n_contrib = (int64_t)g.contrib_cell.size();
    ctx->n_gp = (int64_t)g.gcol.size();
    ctx->n_gc = (int64_t)g.gc_facet.size();
    ctx->nnz = g.nnz;
    ctx->n_dof_owned = 4 * g.n_nodes_owned;
    ctx->n_dof_local = 4 * g.n_nodes;
    const int dim = g.dim, nv1 = g.nv1;
    KCHK(dev_upload_raw(ctx, &ctx->d_cells, mesh->cells, (size_t)g.n_c * nv1));
    KCHK(dev_upload_raw(ctx, &ctx->d_cell_side, mesh->cell_side, (size_t)g.n_c));
    KCHK(dev_upload_raw(ctx, &ctx->d_coords, mesh->coords, (size_t)g.n_v * dim));
    KCHK(dev_upload(ctx, &ctx->d_node_vertex, g.node_vertex));
    KCHK(dev_upload(ctx, &ctx->d_node_side, g.node_side));
    KCHK(dev_upload(ctx, &ctx->d_node_i, g.node_i));
    KCHK(dev_upload(ctx, &ctx->d_node_e, g.node_e));
    KCHK(dev_upload(ctx, &ctx->d_pair_ptr, g.pair_ptr));
    KCHK(dev_upload(ctx, &ctx->d_pair_col, g.pair_col));
    KCHK(dev_upload(ctx, &ctx->d_pair_row, g.pair_row));
    KCHK(dev_upload(ctx, &ctx->d_pair_M, g.pair_M));
    KCHK(dev_upload(ctx, &ctx->d_pair_K, g.pair_K));
    {   // {M, K} interleaved per pair: what the SpMV on A reads instead of the six stored time-invariant entries
        std::vector<double2> mk(g.pair_M.size());
        for (size_t i = 0; i < mk.size(); ++i) mk[i] = make_double2(g.pair_M[i], g.pair_K[i]);
        KCHK(dev_upload(ctx, &ctx->d_pair_MK, mk));
    }
    KCHK(dev_upload(ctx, &ctx->d_contrib_ptr, g.contrib_ptr));

    KCHK(dev_upload(ctx, &ctx->d_contrib_k, g.contrib_k));
    {   // staged assembly: LDS for the cell means of every node of a block; 48 KB per block keeps >= 3 blocks per CU
        const char* es = getenv("KNP_ASM_STAGE");
        const bool want = !(es && atoi(es) == 0);
        const double avg_deg0 = g.n_nodes_owned ? (double)g.pair_col.size() / g.n_nodes_owned : 1.0;
        const int G0 = avg_deg0 <= 4.5 ? 4 : avg_deg0 <= 9.0 ? 8 : avg_deg0 <= 20.0 ? 16 : 32;
        const size_t lds = (size_t)(NT / G0) * g.max_node_cells * 3 * sizeof(double);
        if (want && g.max_node_cells > 0 && g.max_node_cells <= 255 && lds <= 48 * 1024) {
            ctx->asm_stage = g.max_node_cells;
            KCHK(dev_upload(ctx, &ctx->d_node_cell_ptr, g.node_cell_ptr));
            KCHK(dev_upload(ctx, &ctx->d_node_cell, g.node_cell));
            ctx->n_node_cells = (int64_t)g.node_cell.size();
            KCHK(dev_upload(ctx, &ctx->d_contrib_slot, g.contrib_slot));
            const char* et = getenv("KNP_ASM_TRANSPOSED");
            if (!(et && atoi(et) == 0)) KCHK(build_transposed_contribs(ctx));
            KCHK(build_fused_cell_means(ctx, mesh, G0));
        } else {
            KCHK(dev_upload(ctx, &ctx->d_contrib_cell, g.contrib_cell));
        }
    }
    KCHK(dev_upload(ctx, &ctx->d_fv, g.fv));
    KCHK(dev_upload(ctx, &ctx->d_fmeas, g.fmeas));
    KCHK(dev_upload_raw(ctx, &ctx->d_gamma_prog, mesh->gamma_prog, (size_t)g.n_g));
    KCHK(dev_upload_raw(ctx, &ctx->d_qp, mesh->q_pts, (size_t)g.n_q * dim));
    KCHK(dev_upload_raw(ctx, &ctx->d_qw, mesh->q_w, (size_t)g.n_q));
    KCHK(dev_upload(ctx, &ctx->d_gv_vertex, g.gv_vertex));
    KCHK(dev_upload(ctx, &ctx->d_gv_node_i, g.gv_node_i));
    KCHK(dev_upload(ctx, &ctx->d_gv_node_e, g.gv_node_e));
    KCHK(dev_upload(ctx, &ctx->d_node_gv, g.node_gv));
    KCHK(dev_upload(ctx, &ctx->d_gptr, g.gptr));
    KCHK(dev_upload(ctx, &ctx->d_gcol, g.gcol));
    KCHK(dev_upload(ctx, &ctx->d_grow, g.grow));
    KCHK(dev_upload(ctx, &ctx->d_gq_i, g.gq_i));
    KCHK(dev_upload(ctx, &ctx->d_gq_e, g.gq_e));
    KCHK(dev_upload(ctx, &ctx->d_gdiag, g.gdiag));
    KCHK(dev_upload(ctx, &ctx->d_gcptr, g.gcptr));
    KCHK(dev_upload(ctx, &ctx->d_gc_facet, g.gc_facet));
    KCHK(dev_upload(ctx, &ctx->d_gc_lab, g.gc_lab));
    KCHK(dev_upload(ctx, &ctx->d_gx_i, g.gx_i));
    KCHK(dev_upload(ctx, &ctx->d_gx_e, g.gx_e));
    HIPCHK(hipMalloc((void**)&ctx->d_at, std::max<int64_t>(4 * ctx->n_pairs, 2) * sizeof(double)));
    HIPCHK(hipMalloc((void**)&ctx->d_ac, std::max<int64_t>(6 * ctx->n_pairs, 2) * sizeof(double)));
    HIPCHK(hipMalloc((void**)&ctx->d_ax, std::max<int64_t>(8 * ctx->n_gp, 2) * sizeof(double)));
    HIPCHK(hipMemset(ctx->d_at, 0, std::max<int64_t>(4 * ctx->n_pairs, 2) * sizeof(double)));
    HIPCHK(hipMemset(ctx->d_ac, 0, std::max<int64_t>(6 * ctx->n_pairs, 2) * sizeof(double)));
    HIPCHK(hipMemset(ctx->d_ax, 0, std::max<int64_t>(8 * ctx->n_gp, 2) * sizeof(double)));
    {   // lumped mass per owned node (row sums of the P1 mass matrix): diagonal Schur term of the potential
        std::vector<double> ML(std::max(g.n_nodes_owned, 1), 0.0);
        for (int n = 0; n < g.n_nodes_owned; ++n)
            for (int pq = g.pair_ptr[n]; pq < g.pair_ptr[n + 1]; ++pq) ML[n] += g.pair_M[pq];
        KCHK(dev_upload(ctx, &ctx->d_ML, ML));
        HIPCHK(hipMalloc((void**)&ctx->d_cc, std::max<size_t>(g.n_nodes_owned, 1) * sizeof(double)));
        HIPCHK(hipMemset(ctx->d_cc, 0, std::max<size_t>(g.n_nodes_owned, 1) * sizeof(double)));
    }
    // P (block-Jacobi form) is stored pair-major on the device: p_vals[4*pair + field]
    HIPCHK(hipMalloc((void**)&ctx->d_p_vals, std::max<int64_t>(4 * ctx->n_pairs, 1) * sizeof(double)));
    HIPCHK(hipMemset(ctx->d_p_vals, 0, std::max<int64_t>(4 * ctx->n_pairs, 1) * sizeof(double)));
    const int npk = dim * (dim + 1) / 2;
    HIPCHK(hipMalloc((void**)&ctx->d_cbar, (size_t)4 * std::max(g.n_c, 1) * sizeof(double)));
    HIPCHK(hipMemset(ctx->d_cbar, 0, (size_t)4 * std::max(g.n_c, 1) * sizeof(double)));
    HIPCHK(hipMalloc((void**)&ctx->d_fmat, std::max<size_t>((size_t)6 * npk * g.n_g, 1) * sizeof(double)));
    HIPCHK(hipMalloc((void**)&ctx->d_fvec, std::max<size_t>((size_t)7 * dim * g.n_g, 1) * sizeof(double)));
    HIPCHK(hipMalloc((void**)&ctx->d_partial, (size_t)RED_SLOTS * RED_BLOCKS * sizeof(double)));
    HIPCHK(hipMalloc((void**)&ctx->d_red, RED_SLOTS * sizeof(double)));
    HIPCHK(hipMemset(ctx->d_red, 0, RED_SLOTS * sizeof(double)));
    HIPCHK(hipHostMalloc((void**)&ctx->h_red, RED_SLOTS * sizeof(double), hipHostMallocMapped));
    if (hipHostGetDevicePointer((void**)&ctx->h_red_dev, ctx->h_red, 0) != hipSuccess) ctx->h_red_dev = nullptr;
    HIPCHK(hipHostMalloc((void**)&ctx->h_seq, 64, hipHostMallocMapped));
    *ctx->h_seq = 0;
    if (hipHostGetDevicePointer((void**)&ctx->h_seq_dev, (void*)ctx->h_seq, 0) != hipSuccess) { ctx->h_seq_dev = nullptr; }
    if (getenv("KNP_NO_SPIN")) ctx->h_seq_dev = nullptr;
    HIPCHK(hipMalloc((void**)&ctx->d_defl_einv, DEFL_MAX * DEFL_MAX * sizeof(double)));
    HIPCHK(hipMalloc((void**)&ctx->d_y, RED_SLOTS * sizeof(double)));
    HIPCHK(hipMalloc((void**)&ctx->d_vbj, std::max<size_t>((size_t)16 * g.n_nodes_owned, 1) * sizeof(double)));
    ctx->n_red_blocks = std::min(RED_BLOCKS, nblocks(ctx->n_dof_owned));
    {
        const double avg_deg = g.n_nodes_owned ? (double)ctx->n_pairs / g.n_nodes_owned : 1.0;
        const int lanes_per_pair = avg_deg <= 4.5 ? 4 : avg_deg <= 9.0 ? 8 : avg_deg <= 20.0 ? 16 : 32;   // one lane per pair of a node
        // SpMV on A: two pairs in flight per lane, half the lanes (measured on MI355X, rounds 2-3: 512^2 20.6 -> 19.9 us, cube 136^3
        // 448 -> 411 us against one pair per lane)
        ctx->spmv_group = std::max(4, lanes_per_pair / 2);
        const char* ef = getenv("KNP_ASM_FULL");
        ctx->asm_full = (ef && atoi(ef) > 0) ? 1 : 0;
        const char* e = getenv("KNP_SPMV");
        if (e && atoi(e) > 0) ctx->spmv_group = atoi(e);
        ctx->asm_group = lanes_per_pair;
        const char* ea = getenv("KNP_ASM_GROUP");
        if (ea && atoi(ea) > 0 && ctx->asm_stage == 0) ctx->asm_group = atoi(ea);   // (the staged variant sized its LDS for the default)
        // the level-0 preconditioner kernels move half the bytes per node pair (fp32 P, no cross block): fewer lanes per node
        ctx->pc_group = std::max(4, lanes_per_pair / 2);
        const char* ep = getenv("KNP_PC_GROUP");
        if (ep && atoi(ep) > 0) ctx->pc_group = atoi(ep);
    }
    if (g.n_nodes > g.n_nodes_owned) {   // multi-GPU: rows that touch a ghost column wait for the halo, the others do not
        std::vector<int32_t> li, lb;
        const int no = g.n_nodes_owned;
        for (int n = 0; n < no; ++n) {
            bool bnd = false;
            for (int pq = g.pair_ptr[n]; pq < g.pair_ptr[n + 1] && !bnd; ++pq) bnd = g.pair_col[pq] >= no;
            const int A = g.node_gv[n];
            if (A >= 0 && !bnd) {
                const std::vector<int32_t>& gx = g.node_side[n] ? g.gx_e : g.gx_i;
                for (int sgp = g.gptr[A]; sgp < g.gptr[A + 1] && !bnd; ++sgp) bnd = gx[sgp] >= no;
            }
            (bnd ? lb : li).push_back(n);
        }
        ctx->n_int = (int)li.size(); ctx->n_bnd = (int)lb.size();
        KCHK(dev_upload(ctx, &ctx->d_nodes_int, li));
        KCHK(dev_upload(ctx, &ctx->d_nodes_bnd, lb));
    }
    // free the big host arrays that are no longer needed (pattern kept for export)
    std::vector<int32_t>().swap(g.contrib_cell);
    std::vector<double>().swap(g.contrib_k);
    std::vector<int32_t>().swap(g.node_cell);
    std::vector<uint8_t>().swap(g.contrib_slot);
    return KNP_OK;
}

int knp_destroy(knp_ctx* ctx) {
    if (!ctx) return KNP_OK;
    (void)hipDeviceSynchronize();
    if (ctx->stream2) { (void)hipEventDestroy(ctx->ev_fork); (void)hipEventDestroy(ctx->ev_join); (void)hipStreamDestroy(ctx->stream2); }
    if (ctx->stream3) { (void)hipEventDestroy(ctx->ev_x); (void)hipEventDestroy(ctx->ev_halo); (void)hipStreamDestroy(ctx->stream3); }
    if (ctx->stream_asm) { (void)hipEventDestroy(ctx->ev_fork_asm); (void)hipEventDestroy(ctx->ev_asm); (void)hipStreamDestroy(ctx->stream_asm); }
    dev_free(ctx->d_t2_s); dev_free(ctx->d_w2_s); dev_free(ctx->d_wb); dev_free(ctx->d_partial_s);
    dev_free(ctx->d_nodes_int); dev_free(ctx->d_nodes_bnd);
    knp_p2p_free(ctx);
    knp_jit_release(ctx);
    for (auto& r : ctx->prof_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto& e : ctx->prof_pool) (void)hipEventDestroy(e);
    for (auto& e : ctx->tm_events) (void)hipEventDestroy(e);
    dev_free(ctx->d_cells); dev_free(ctx->d_cell_side); dev_free(ctx->d_coords);
    dev_free(ctx->d_node_vertex); dev_free(ctx->d_node_side); dev_free(ctx->d_node_i); dev_free(ctx->d_node_e);
    dev_free(ctx->d_pair_ptr); dev_free(ctx->d_pair_col); dev_free(ctx->d_pair_row);
    dev_free(ctx->d_pair_M); dev_free(ctx->d_pair_K); dev_free(ctx->d_pair_MK);
    dev_free(ctx->d_contrib_ptr); dev_free(ctx->d_contrib_cell); dev_free(ctx->d_contrib_k);
    dev_free(ctx->d_node_cell_ptr); dev_free(ctx->d_node_cell); dev_free(ctx->d_contrib_slot); dev_free(ctx->d_ncv); dev_free(ctx->d_knod);
    dev_free(ctx->d_tc_meta); dev_free(ctx->d_tc_k); dev_free(ctx->d_tc_slot);
    dev_free(ctx->d_fv); dev_free(ctx->d_fmeas); dev_free(ctx->d_gamma_prog); dev_free(ctx->d_qp); dev_free(ctx->d_qw);
    dev_free(ctx->d_gv_vertex); dev_free(ctx->d_gv_node_i); dev_free(ctx->d_gv_node_e); dev_free(ctx->d_node_gv);
    dev_free(ctx->d_gptr); dev_free(ctx->d_gcol); dev_free(ctx->d_grow); dev_free(ctx->d_gq_i); dev_free(ctx->d_gq_e);
    dev_free(ctx->d_gdiag); dev_free(ctx->d_gcptr); dev_free(ctx->d_gc_facet); dev_free(ctx->d_gc_lab);
    dev_free(ctx->d_at); dev_free(ctx->d_ac); dev_free(ctx->d_ax); dev_free(ctx->d_gx_i); dev_free(ctx->d_gx_e);
    dev_free(ctx->d_p_vals); dev_free(ctx->d_px);
    dev_free(ctx->d_cbar); dev_free(ctx->d_fmat); dev_free(ctx->d_fvec);
    dev_free(ctx->d_partial); dev_free(ctx->d_red); dev_free(ctx->d_y); dev_free(ctx->d_vbj); dev_free(ctx->d_gm);
    if (ctx->h_red) (void)hipHostFree(ctx->h_red);
    if (ctx->h_seq) (void)hipHostFree((void*)ctx->h_seq);
    dev_free(ctx->d_V); dev_free(ctx->d_w); dev_free(ctx->d_t);
    for (auto& p : ctx->progs) { dev_free(p.d_code); dev_free(p.d_consts); if (p.h_consts) (void)hipHostFree(p.h_consts); }
    dev_free(ctx->d_prog_code); dev_free(ctx->d_prog_consts); dev_free(ctx->d_prog_len); dev_free(ctx->d_prog_nconsts);
    for (int h = 0; h < KNP_MAX_HIER; ++h) free_hier(ctx->hier[h]);
    dev_free(ctx->d_p_vals_f);
    dev_free(ctx->d_ML); dev_free(ctx->d_cc); dev_free(ctx->d_t2); dev_free(ctx->d_w2);
    dev_free(ctx->d_defl_mode); dev_free(ctx->d_defl_einv); dev_free(ctx->d_bc_dofs);
    delete ctx;
    return KNP_OK;
}

int knp_set_stream(knp_ctx* ctx, void* s) {
    CHECK_CTX(ctx);
    ctx->stream = (hipStream_t)s;
    return KNP_OK;
}
int knp_set_comm(knp_ctx* ctx, knp_halo_fn halo, knp_allreduce_fn ar, void* user) {
    CHECK_CTX(ctx);
    ctx->halo = halo; ctx->allreduce = ar; ctx->comm_user = user;
    ctx->phi_count_cached = -1;
    return KNP_OK;
}

int knp_get_sizes(const knp_ctx* ctx, int64_t* s) {
    if (!ctx || !s) return KNP_E_ARG;
    for (int i = 0; i < KNP_SZ_COUNT; ++i) s[i] = 0;
    s[KNP_SZ_N_NODES] = ctx->g.n_nodes;
    s[KNP_SZ_N_NODES_OWNED] = ctx->g.n_nodes_owned;
    s[KNP_SZ_N_DOF_LOCAL] = ctx->n_dof_local;
    s[KNP_SZ_N_DOF_OWNED] = ctx->n_dof_owned;
    s[KNP_SZ_NNZ] = ctx->nnz;
    s[KNP_SZ_N_PAIRS] = ctx->n_pairs;
    s[KNP_SZ_N_CONTRIB] = ctx->n_contrib;
    s[KNP_SZ_N_GAMMA_VERTS] = ctx->g.n_gv;
    s[KNP_SZ_N_GAMMA_PAIRS] = ctx->n_gp;
    s[KNP_SZ_NNZ_P] = 4 * ctx->n_pairs;
    s[KNP_SZ_N_PHI_OWNED] = ctx->g.n_nodes_owned;
    s[KNP_SZ_NNZ_P_PHI] = phi_block_nnz(ctx);
    return KNP_OK;
}
int knp_get_layout(const knp_ctx* ctx, int32_t* ni, int32_t* ne) {
    if (!ctx || !ni || !ne) return KNP_E_ARG;
    std::memcpy(ni, ctx->g.node_i.data(), ctx->g.node_i.size() * sizeof(int32_t));
    std::memcpy(ne, ctx->g.node_e.data(), ctx->g.node_e.size() * sizeof(int32_t));
    return KNP_OK;
}
int knp_get_csr_pattern(const knp_ctx* cctx, int32_t* rp, int32_t* ci) {
    knp_ctx* ctx = const_cast<knp_ctx*>(cctx);
    if (!ctx || !rp || !ci) return KNP_E_ARG;
    KCHK(knp_build_csr_pattern(ctx->g));
    std::memcpy(rp, ctx->g.rowptr.data(), ctx->g.rowptr.size() * sizeof(int32_t));
    std::memcpy(ci, ctx->g.colind.data(), ctx->g.colind.size() * sizeof(int32_t));
    return KNP_OK;
}
// A in the CSR order of knp_get_csr_pattern, gathered on the host from the pair-major device arrays (parity hook, not a hot path)
int knp_get_csr_values(const knp_ctx* cctx, double* vals) {
    knp_ctx* ctx = const_cast<knp_ctx*>(cctx);
    if (!ctx || !vals) return KNP_E_ARG;
    KCHK(knp_build_csr_pattern(ctx->g));
    KCHK(join_asm(ctx));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    const KnpHostGraph& g = ctx->g;
    std::vector<double> at((size_t)4 * ctx->n_pairs), ac((size_t)6 * ctx->n_pairs), ax((size_t)8 * ctx->n_gp);
    if (!at.empty()) HIPCHK(hipMemcpy(at.data(), ctx->d_at, at.size() * sizeof(double), hipMemcpyDeviceToHost));
    if (!ac.empty()) HIPCHK(hipMemcpy(ac.data(), ctx->d_ac, ac.size() * sizeof(double), hipMemcpyDeviceToHost));
    if (!ax.empty()) HIPCHK(hipMemcpy(ax.data(), ctx->d_ax, ax.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (int n = 0; n < g.n_nodes_owned; ++n) {
        const int p0 = g.pair_ptr[n], deg = g.pair_ptr[n + 1] - p0;
        const int A = g.node_gv[n], sd = g.node_side[n];
        const int s0 = A >= 0 ? g.gptr[A] : 0, x = A >= 0 ? g.gptr[A + 1] - g.gptr[A] : 0;
        for (int f = 0; f < 3; ++f) {
            double* v = vals + g.rowptr[(size_t)4 * n + f];
            for (int q = 0; q < deg; ++q) { v[2 * q] = ac[(size_t)6 * (p0 + q) + f]; v[2 * q + 1] = at[(size_t)4 * (p0 + q) + f]; }
            for (int r = 0; r < x; ++r) v[2 * deg + r] = ax[(size_t)8 * (s0 + r) + 4 * sd + f];
        }
        double* v = vals + g.rowptr[(size_t)4 * n + 3];
        for (int q = 0; q < deg; ++q) {
            for (int f = 0; f < 3; ++f) v[4 * q + f] = ac[(size_t)6 * (p0 + q) + 3 + f];
            v[4 * q + 3] = at[(size_t)4 * (p0 + q) + 3];
        }
        for (int r = 0; r < x; ++r) v[4 * deg + r] = ax[(size_t)8 * (s0 + r) + 4 * sd + 3];
    }
    return KNP_OK;
}
int knp_get_precond_csr(const knp_ctx* cctx, int32_t* rp, int32_t* ci, double* vals) {
    knp_ctx* ctx = const_cast<knp_ctx*>(cctx);
    if (!ctx || !rp || !ci || !vals) return KNP_E_ARG;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HostBuf<double> pm((size_t)4 * ctx->n_pairs);
    if (!pm.p) { ctx->err = "out of host memory"; return KNP_E_STATE; }
    HIPCHK(hipMemcpy(pm.p, ctx->d_p_vals, pm.size() * sizeof(double), hipMemcpyDeviceToHost));
    const KnpHostGraph& g = ctx->g;
#pragma omp parallel for schedule(static) num_threads(knp_host_threads())
    for (int n = 0; n < g.n_nodes_owned; ++n) {
        const int p0 = g.pair_ptr[n], deg = g.pair_ptr[n + 1] - p0;
        for (int f = 0; f < 4; ++f) {
            rp[(size_t)4 * n + f] = 4 * p0 + f * deg;
            for (int q = 0; q < deg; ++q) {
                ci[(size_t)4 * p0 + (size_t)f * deg + q] = 4 * g.pair_col[p0 + q] + f;
                vals[(size_t)4 * p0 + (size_t)f * deg + q] = pm[(size_t)4 * (p0 + q) + f];
            }
        }
    }
    rp[(size_t)4 * g.n_nodes_owned] = (int32_t)(4 * ctx->n_pairs);
    return KNP_OK;
}
int knp_matrix_max_abs(knp_ctx* ctx, double* out) {
    CHECK_CTX(ctx);
    if (!out) return KNP_E_ARG;
    if (!ctx->have_A) { ctx->err = "matrix not assembled"; return KNP_E_STATE; }
    side_discard(ctx);   // d_partial is also the side stream's reduction scratch (knp_gmres_prepare)
    KCHK(join_asm(ctx));
    const double* arr[3] = {ctx->d_at, ctx->d_ac, ctx->d_ax};
    const int64_t len[3] = {4 * ctx->n_pairs, 6 * ctx->n_pairs, 8 * ctx->n_gp};
    std::vector<double> h(3 * RED_BLOCKS, 0.0);
    for (int k = 0; k < 3; ++k)
        if (len[k] > 0) hipLaunchKernelGGL(k_absmax, dim3(RED_BLOCKS), dim3(NT), 0, ctx->stream, len[k], arr[k], ctx->d_partial + (size_t)k * RED_BLOCKS);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(h.data(), ctx->d_partial, h.size() * sizeof(double), hipMemcpyDeviceToHost));
    double m = 0.0;
    for (int k = 0; k < 3; ++k)
        if (len[k] > 0)
            for (int b = 0; b < RED_BLOCKS; ++b) m = std::max(m, h[(size_t)k * RED_BLOCKS + b]);
    *out = m;
    return KNP_OK;
}

int knp_set_params(knp_ctx* ctx, double dt, double F, double C_M, double psi, int32_t n_ions, const double* z,
                   const double* Di, const double* De) {
    CHECK_CTX(ctx);
    if (n_ions != 3 || !z || !Di || !De) { ctx->err = "n_ions must be 3 (Na, K, Cl; KNPEMIx_problem.py:980-981)"; return KNP_E_ARG; }
    if (!(dt > 0) || !(psi > 0) || F == 0.0) { ctx->err = "dt, psi must be positive and F non-zero"; return KNP_E_ARG; }
    ctx->dt = dt; ctx->F = F; ctx->C_M = C_M; ctx->psi = psi; ctx->n_ions = 3;
    ctx->asm_dt = -1.0;   // constant blocks must be rewritten
    for (int j = 0; j < 3; ++j) {
        if (z[j] == 0.0 || !(Di[j] > 0) || !(De[j] > 0)) { ctx->err = "valence must be non-zero and diffusivities positive"; return KNP_E_ARG; }
        ctx->z[j] = z[j]; ctx->Di[j] = Di[j]; ctx->De[j] = De[j];
    }
    return KNP_OK;
}

static int validate_program(knp_ctx* ctx, int n_instr, const int32_t* code, int n_consts, int* n_regs_out) {
    int max_reg = -1;
    for (int i = 0; i < n_instr; ++i) {
        const int op = code[4 * i], d = code[4 * i + 1], a = code[4 * i + 2], b = code[4 * i + 3];
        auto regok = [&](int r) { if (r > max_reg) max_reg = r; return r >= 0 && r < KNP_MAX_PROG_REGS; };
        bool ok = true;
        switch (op) {
            case KNP_OP_CONST: ok = regok(d) && a >= 0 && a < n_consts; break;
            case KNP_OP_KI: case KNP_OP_KE: ok = regok(d) && a >= 0 && a < 3; break;
            case KNP_OP_PHIM: ok = regok(d); break;
            case KNP_OP_AUX: ok = regok(d) && a >= 0 && a < KNP_MAX_AUX; break;
            case KNP_OP_X: ok = regok(d) && a >= 0 && a < ctx->g.dim; break;
            case KNP_OP_NEG: case KNP_OP_LN: case KNP_OP_EXP: case KNP_OP_SQRT: case KNP_OP_ABS: case KNP_OP_NOT:
            case KNP_OP_MOV: case KNP_OP_POWI: ok = regok(d) && regok(a); break;
            case KNP_OP_OUT: ok = a >= 0 && a < 3 && regok(b); break;
            default: ok = (op >= KNP_OP_ADD && op <= KNP_OP_SEL) && regok(d) && regok(a) && regok(b); break;
        }
        if (!ok) { ctx->err = "invalid membrane program instruction " + std::to_string(i); return KNP_E_ARG; }
    }
    *n_regs_out = max_reg + 1;
    return KNP_OK;
}

int knp_set_program(knp_ctx* ctx, int32_t id, int32_t n_instr, const int32_t* code, int32_t n_consts, const double* consts) {
    CHECK_CTX(ctx);
    if (id < 0 || id > 4096 || n_instr < 0 || (n_instr && !code) || n_consts < 0 || (n_consts && !consts)) { ctx->err = "bad program arguments"; return KNP_E_ARG; }
    int n_regs = 0;
    KCHK(validate_program(ctx, n_instr, code, n_consts, &n_regs));
    if ((int)ctx->progs.size() <= id) ctx->progs.resize(id + 1);
    KnpProgram& p = ctx->progs[id];
    p.n_regs = n_regs;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dev_free(p.d_code); dev_free(p.d_consts);
    if (p.h_consts) { (void)hipHostFree(p.h_consts); p.h_consts = nullptr; }
    p.n_instr = n_instr; p.n_consts = n_consts;
    p.h_code.assign(code, code + (size_t)4 * n_instr);
    KCHK(dev_upload_raw(ctx, &p.d_code, code, (size_t)4 * n_instr));
    KCHK(dev_upload_raw(ctx, &p.d_consts, consts, (size_t)n_consts));
    ctx->progs_dirty = true;
    return KNP_OK;
}
int knp_set_program_constants(knp_ctx* ctx, int32_t id, int32_t n_consts, const double* consts) {
    CHECK_CTX(ctx);
    if (id < 0 || id >= (int)ctx->progs.size() || n_consts != ctx->progs[id].n_consts) { ctx->err = "program id / constant count mismatch"; return KNP_E_ARG; }
    if (n_consts) {
        // staged through a pinned buffer of the ctx: the copy is stream-ordered and the caller's (pageable) buffer is free
        // on return -- no synchronisation in the per-step path.  A second update before the first copy ran would only
        // make that copy carry the newer values.
        KnpProgram& pr = ctx->progs[id];
        if (!pr.h_consts) HIPCHK(hipHostMalloc((void**)&pr.h_consts, n_consts * sizeof(double), hipHostMallocDefault));
        std::memcpy(pr.h_consts, consts, n_consts * sizeof(double));
        HIPCHK(hipMemcpyAsync(pr.d_consts, pr.h_consts, n_consts * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    return KNP_OK;
}
static int sync_program_table(knp_ctx* ctx) {
    if (!ctx->progs_dirty) return KNP_OK;
    const size_t np = std::max<size_t>(ctx->progs.size(), 1);
    std::vector<int32_t*> codes(np, nullptr);
    std::vector<double*> consts(np, nullptr);
    std::vector<int32_t> lens(np, 0), ncs(np, 0);
    ctx->prog_regs = ctx->prog_len_cap = ctx->prog_consts_cap = 0;
    for (size_t i = 0; i < ctx->progs.size(); ++i) {
        const KnpProgram& pr = ctx->progs[i];
        codes[i] = pr.d_code; consts[i] = pr.d_consts; lens[i] = pr.n_instr; ncs[i] = pr.n_consts;
        ctx->prog_regs = std::max(ctx->prog_regs, pr.n_regs);
        ctx->prog_len_cap = std::max(ctx->prog_len_cap, pr.n_instr);
        ctx->prog_consts_cap = std::max(ctx->prog_consts_cap, pr.n_consts);
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dev_free(ctx->d_prog_code); dev_free(ctx->d_prog_consts); dev_free(ctx->d_prog_len); dev_free(ctx->d_prog_nconsts);
    KCHK(dev_upload(ctx, &ctx->d_prog_code, codes));
    KCHK(dev_upload(ctx, &ctx->d_prog_consts, consts));
    KCHK(dev_upload(ctx, &ctx->d_prog_len, lens));
    KCHK(dev_upload(ctx, &ctx->d_prog_nconsts, ncs));
    ctx->progs_dirty = false;
    knp_jit_build(ctx);   // native code for the membrane programs when hiprtc is available; else the interpreter runs
    return KNP_OK;
}

const char* knp_jit_status(knp_ctx* ctx) {
    if (!ctx) return "";
    if (ctx->progs_dirty && !ctx->progs.empty()) (void)sync_program_table(ctx);
    return ctx->jit_msg.c_str();
}

int knp_set_dirichlet(knp_ctx* ctx, int32_t n, const int32_t* dofs) {
    CHECK_CTX(ctx);
    side_discard(ctx);
    if (n < 0 || (n > 0 && !dofs)) { ctx->err = "bad Dirichlet arguments"; return KNP_E_ARG; }
    for (int i = 0; i < n; ++i)
        if (dofs[i] < 0 || dofs[i] >= ctx->n_dof_owned) { ctx->err = "Dirichlet dof out of range (owned dofs only)"; return KNP_E_ARG; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dev_free(ctx->d_bc_dofs);
    ctx->n_bc = 0;
    if (n > 0) {
        KCHK(dev_upload_raw(ctx, &ctx->d_bc_dofs, dofs, (size_t)n));
        ctx->n_bc = n;
    }
    ctx->asm_dt = -1.0;   // constant blocks of the touched rows must be rewritten if the set changes
    return KNP_OK;
}

int knp_set_sources(knp_ctx* ctx, const double* const* fi, const double* const* fe) {
    CHECK_CTX(ctx);
    ctx->have_sources = false;
    for (int j = 0; j < 3; ++j) {
        ctx->src_i[j] = fi ? fi[j] : nullptr;
        ctx->src_e[j] = fe ? fe[j] : nullptr;
        if (ctx->src_i[j] || ctx->src_e[j]) ctx->have_sources = true;
    }
    return KNP_OK;
}

static int check_fields(knp_ctx* ctx, const knp_fields* f, bool need_phim) {
    if (!f) { ctx->err = "null fields"; return KNP_E_ARG; }
    for (int j = 0; j < 3; ++j)
        if (!f->k_i[j] || !f->k_e[j]) { ctx->err = "null concentration field"; return KNP_E_ARG; }
    if (need_phim && !f->phi_m) { ctx->err = "null phi_m field"; return KNP_E_ARG; }
    if (!(ctx->dt > 0)) { ctx->err = "knp_set_params has not been called"; return KNP_E_STATE; }
    return KNP_OK;
}

// cc = psi / (sum_j z_j^2 k_j) / M_lumped at every owned node (Schur term of the block-triangular preconditioner)
static void launch_schur_diag(knp_ctx* ctx, const FieldPtrs& f) {
    const KnpHostGraph& g = ctx->g;
    hipLaunchKernelGGL(k_schur_diag, dim3(nblocks(g.n_nodes_owned)), dim3(NT), 0, ctx->stream, g.n_nodes_owned, ctx->psi, ctx->z[0], ctx->z[1],
                       ctx->z[2], ctx->d_node_vertex, ctx->d_node_side, f, ctx->d_ML, ctx->d_cc);
    ctx->have_cc = true;
}

// join the matrix assembly that runs on its own stream (knp_assemble_matrix_async): everything that reads A, or writes what
// that assembly reads or uses as scratch (the cell means, the facet matrices), calls this first
static int join_asm(knp_ctx* ctx) {
    if (ctx->asm_pending) {
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_asm, 0));
        ctx->asm_pending = false;
    }
    return KNP_OK;
}

static int assemble_matrix_on_stream(knp_ctx* ctx, const knp_fields* fields, bool with_schur_diag) {
    const KnpHostGraph& g = ctx->g;
    const DevParams P = make_params(ctx);
    FieldPtrs f = make_fields(fields);
    ProfScope ps(ctx, 3);
    launch_cell_means(ctx, f);
    // The (k,k) and (phi,k) blocks do not depend on the previous solution (SURVEY 3.2 obs. 1): after the first
    // assembly only the K[k_prev]-type and membrane entries are rewritten, unless KNP_ASM_FULL=1 asks for the
    // reference's behaviour (A.zeroEntries() + full re-assembly, KNPEMIx_solver.py:110-115).
    const bool td_only = ctx->have_A && !ctx->asm_full && ctx->asm_dt == ctx->dt;
    if (td_only) launch_assemble_nodes<false, true>(ctx, P, ctx->d_at, ctx->d_ac);
    else launch_assemble_nodes<false, false>(ctx, P, ctx->d_at, ctx->d_ac);
    ctx->asm_dt = ctx->dt;
    if (g.n_g > 0) {
        if (g.dim == 2)
            hipLaunchKernelGGL((k_gamma_facets<2, true, false, 8, 1, NT, 1>), dim3(nblocks((int64_t)g.n_g * 8)), dim3(NT), 0, ctx->stream, g.n_g, g.n_q, P,
                               ctx->d_fv, ctx->d_fmeas, ctx->d_qp, ctx->d_qw, f, 0, ctx->d_coords, ctx->d_gamma_prog,
                               (const int32_t* const*)nullptr, (const int32_t*)nullptr, (const double* const*)nullptr,
                               (const int32_t*)nullptr, 0, 0, ctx->d_fmat, ctx->d_fvec);
        else if (gamma_many(g)) {  // 4 lanes per facet: the 36 points exactly (see gamma_many)
            // (matrix part: 9 points in flight per lane 0.76 ms on 1.5 M facets, 3 points 0.82, 1 point 0.95 -- unlike the mechanism currents)
            hipLaunchKernelGGL((k_gamma_facets<3, true, false, 4, 9, 64, 1>), dim3((unsigned)(((int64_t)g.n_g * 4 + 63) / 64)), dim3(64), 0, ctx->stream, g.n_g, g.n_q, P,
                               ctx->d_fv, ctx->d_fmeas, ctx->d_qp, ctx->d_qw, f, 0, ctx->d_coords, ctx->d_gamma_prog,
                               (const int32_t* const*)nullptr, (const int32_t*)nullptr, (const double* const*)nullptr,
                               (const int32_t*)nullptr, 0, 0, ctx->d_fmat, ctx->d_fvec);
        }
        else
            hipLaunchKernelGGL((k_gamma_facets<3, true, false, 16, 3, 64, 2>), dim3((unsigned)(((int64_t)g.n_g * 16 + 63) / 64)), dim3(64), 0, ctx->stream, g.n_g, g.n_q, P,
                               ctx->d_fv, ctx->d_fmeas, ctx->d_qp, ctx->d_qw, f, 0, ctx->d_coords, ctx->d_gamma_prog,
                               (const int32_t* const*)nullptr, (const int32_t*)nullptr, (const double* const*)nullptr,
                               (const int32_t*)nullptr, 0, 0, ctx->d_fmat, ctx->d_fvec);
        if (ctx->n_gp)
            hipLaunchKernelGGL((k_gamma_pairs<false>), dim3(nblocks(ctx->n_gp)), dim3(NT), 0, ctx->stream, ctx->n_gp, g.n_g, g.dim, P,
                               ctx->d_grow, ctx->d_gptr, ctx->d_gv_node_i, ctx->d_gv_node_e, ctx->d_gq_i, ctx->d_gq_e,
                               ctx->d_gcptr, ctx->d_gc_facet, ctx->d_gc_lab, ctx->d_fmeas, ctx->d_fmat, ctx->d_pair_ptr,
                               ctx->d_at, ctx->d_ax);
    }
    if (ctx->n_bc > 0)
        hipLaunchKernelGGL(k_dirichlet_rows_A, dim3(nblocks(ctx->n_bc)), dim3(NT), 0, ctx->stream, ctx->n_bc, ctx->d_bc_dofs, ctx->d_pair_ptr,
                           ctx->d_pair_col, ctx->d_node_gv, ctx->d_node_side, ctx->d_gptr, ctx->d_ac, ctx->d_at, ctx->d_ax);
    // The Schur diagonal depends on the fields only.  knp_assemble_rhs of the same step has already written it; while a
    // side-stream preconditioner application (knp_gmres_prepare) is in flight it READS d_cc, so it must not be rewritten here
    // (nor by the asynchronous form, which runs next to the right-hand side assembly that writes it).
    if (with_schur_diag && !ctx->prep_b) launch_schur_diag(ctx, f);
    HIPCHK(hipGetLastError());
    ctx->have_A = true;
    if (ctx->pc_kind == KNP_PC_VBJACOBI) {
        hipLaunchKernelGGL(k_vbj_extract, dim3(nblocks(g.n_nodes_owned)), dim3(NT), 0, ctx->stream, g.n_nodes_owned,
                           ctx->d_pair_ptr, ctx->d_pair_col, ctx->d_ac, ctx->d_at, ctx->d_ax, ctx->d_node_gv, ctx->d_node_side,
                           ctx->d_gdiag, ctx->d_vbj);
        HIPCHK(hipGetLastError());
    }
    return KNP_OK;
}

int knp_assemble_matrix(knp_ctx* ctx, const knp_fields* fields) {
    CHECK_CTX(ctx);
    KCHK(check_fields(ctx, fields, false));
    KCHK(join_asm(ctx));
    return assemble_matrix_on_stream(ctx, fields, true);
}

// The same assembly on the library's own stream: it depends on the previous solution only, not on the gating update or the
// right-hand side of the step, so the caller may enqueue the right-hand side chain (and knp_gmres_prepare) on the main stream
// while it runs.  Every later call that needs A joins it (join_asm).  Falls back to the in-line form where the matrix is consumed
// at once (vertex-block Jacobi extracts its blocks) or every kernel class is being timed.
int knp_assemble_matrix_async(knp_ctx* ctx, const knp_fields* fields) {
    CHECK_CTX(ctx);
    KCHK(check_fields(ctx, fields, false));
    KCHK(join_asm(ctx));
    static const bool off = getenv("KNP_ASM_ASYNC") && atoi(getenv("KNP_ASM_ASYNC")) == 0;
    if (off || ctx->pc_kind == KNP_PC_VBJACOBI || (ctx->prof_on & ~1) || !ctx->have_A || ctx->prep_b)
        return assemble_matrix_on_stream(ctx, fields, true);
    if (!ctx->stream_asm) {
        HIPCHK(hipStreamCreateWithFlags(&ctx->stream_asm, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork_asm, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_asm, hipEventDisableTiming));
    }
    HIPCHK(hipEventRecord(ctx->ev_fork_asm, ctx->stream));            // after everything that wrote the fields (unpack)
    HIPCHK(hipStreamWaitEvent(ctx->stream_asm, ctx->ev_fork_asm, 0));
    hipStream_t main_stream = ctx->stream;
    ctx->stream = ctx->stream_asm;
    const int rc = assemble_matrix_on_stream(ctx, fields, false);
    ctx->stream = main_stream;
    KCHK(rc);
    HIPCHK(hipEventRecord(ctx->ev_asm, ctx->stream_asm));
    ctx->asm_pending = true;
    return KNP_OK;
}

// Form of the potential block of P.  0 (default): the reference's block-Jacobi form, - (C_M/F) M_Gamma on each side, sides uncoupled
// (KNPEMIx_problem.py:735-738).  1: the potential block of A itself at assembly time, + (C_M/F) M_Gamma and the phi_i-phi_e
// coupling (:637-638) -- what `btcc` builds its potential hierarchy on: in membrane-dominated meshes the coupling is the dominant
// term of that block, and the uncoupled form costs 40 % more GMRES iterations (preconditioners).  Call before
// knp_assemble_precond; knp_get_precond_phi_csr exports the coupled block.
int knp_pc_set_coupled_potential(knp_ctx* ctx, int32_t on) {
    CHECK_CTX(ctx);
    side_discard(ctx);
    ctx->pc_coupled_phi = on ? 1 : 0;
    if (on && !ctx->d_px) {
        HIPCHK(hipMalloc((void**)&ctx->d_px, (size_t)std::max<int64_t>(ctx->n_gp, 1) * sizeof(double)));
        HIPCHK(hipMemset(ctx->d_px, 0, (size_t)std::max<int64_t>(ctx->n_gp, 1) * sizeof(double)));
    }
    ctx->have_P = false;
    return KNP_OK;
}
// number of entries / CSR of the potential block of P on NODE-indexed rows and columns (owned rows, local columns): same-side node
// pairs and, in the coupled form, the other side's node at every membrane neighbour
static int64_t phi_block_nnz(const knp_ctx* ctx) {
    const KnpHostGraph& g = ctx->g;
    int64_t nnz = g.pair_ptr[g.n_nodes_owned];
    if (ctx->pc_coupled_phi)
        for (int n = 0; n < g.n_nodes_owned; ++n) {
            const int A = g.node_gv[n];
            if (A >= 0) nnz += g.gptr[A + 1] - g.gptr[A];
        }
    return nnz;
}
int knp_get_precond_phi_csr(const knp_ctx* cctx, int32_t* rp, int32_t* ci, double* vals) {
    knp_ctx* ctx = const_cast<knp_ctx*>(cctx);
    if (!ctx || !rp || !ci || !vals) return KNP_E_ARG;
    if (!ctx->have_P) { ctx->err = "P not assembled"; return KNP_E_STATE; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    std::vector<double> pm((size_t)4 * ctx->n_pairs), px;
    HIPCHK(hipMemcpy(pm.data(), ctx->d_p_vals, pm.size() * sizeof(double), hipMemcpyDeviceToHost));
    if (ctx->pc_coupled_phi && ctx->n_gp > 0) {
        px.resize((size_t)ctx->n_gp);
        HIPCHK(hipMemcpy(px.data(), ctx->d_px, px.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    const KnpHostGraph& g = ctx->g;
    int64_t k = 0;
    for (int n = 0; n < g.n_nodes_owned; ++n) {
        rp[n] = (int32_t)k;
        for (int p = g.pair_ptr[n]; p < g.pair_ptr[n + 1]; ++p) { ci[k] = g.pair_col[p]; vals[k++] = pm[(size_t)4 * p + 3]; }
        const int A = g.node_gv[n];
        if (ctx->pc_coupled_phi && A >= 0) {
            const std::vector<int32_t>& gx = g.node_side[n] ? g.gx_e : g.gx_i;
            for (int s = g.gptr[A]; s < g.gptr[A + 1]; ++s) { ci[k] = gx[s]; vals[k++] = px[s]; }
        }
    }
    rp[g.n_nodes_owned] = (int32_t)k;
    return KNP_OK;
}

int knp_assemble_precond(knp_ctx* ctx, const knp_fields* fields) {
    CHECK_CTX(ctx);
    side_discard(ctx);
    KCHK(join_asm(ctx));
    KCHK(check_fields(ctx, fields, false));
    const KnpHostGraph& g = ctx->g;
    const DevParams P = make_params(ctx);
    FieldPtrs f = make_fields(fields);
    ProfScope ps(ctx, 3);
    launch_cell_means(ctx, f);
    launch_assemble_nodes<true, false>(ctx, P, ctx->d_p_vals, nullptr);
    if (g.n_g > 0 && ctx->n_gp)
        hipLaunchKernelGGL((k_gamma_pairs<true>), dim3(nblocks(ctx->n_gp)), dim3(NT), 0, ctx->stream, ctx->n_gp, g.n_g, g.dim, P,
                           ctx->d_grow, ctx->d_gptr, ctx->d_gv_node_i, ctx->d_gv_node_e, ctx->d_gq_i, ctx->d_gq_e,
                           ctx->d_gcptr, ctx->d_gc_facet, ctx->d_gc_lab, ctx->d_fmeas, ctx->d_fmat, ctx->d_pair_ptr,
                           ctx->d_p_vals, nullptr, ctx->pc_coupled_phi ? ctx->d_px : nullptr);
    if (ctx->n_bc > 0)
        hipLaunchKernelGGL(k_dirichlet_rows_P, dim3(nblocks(ctx->n_bc)), dim3(NT), 0, ctx->stream, ctx->n_bc, ctx->d_bc_dofs, ctx->d_pair_ptr,
                           ctx->d_pair_col, ctx->d_p_vals);
    HIPCHK(hipGetLastError());
    ctx->have_P = true;
    return KNP_OK;
}

int knp_assemble_rhs(knp_ctx* ctx, const knp_fields* fields, double* b) {
    CHECK_CTX(ctx);
    side_discard(ctx);
    KCHK(check_fields(ctx, fields, true));
    if (!b) { ctx->err = "null b"; return KNP_E_ARG; }
    const KnpHostGraph& g = ctx->g;
    const DevParams P = make_params(ctx);
    FieldPtrs f = make_fields(fields);
    int n_aux = 0;
    for (int k = 0; k < KNP_MAX_AUX; ++k)
        if (fields->aux[k]) n_aux = k + 1;
    for (int k = 0; k < n_aux; ++k)
        if (!fields->aux[k]) { ctx->err = "aux fields must be contiguous from index 0"; return KNP_E_ARG; }
    if (g.n_g > 0) {
        KCHK(sync_program_table(ctx));
        // every facet's program must exist
        if ((int)ctx->progs.size() <= ctx->max_prog) { ctx->err = "a membrane facet refers to a program that was not set (knp_set_program)"; return KNP_E_STATE; }
        for (int i = 0; i <= ctx->max_prog; ++i)
            if (!ctx->progs[i].d_code) { ctx->err = "membrane program " + std::to_string(i) + " not set"; return KNP_E_STATE; }
    }
    ProfScope ps(ctx, 3);
    if (g.n_g > 0) {
        // dynamic LDS: register file [n_regs][NT] | constants | code of the block's program
        const int n_regs = std::max(ctx->prog_regs, 1), ccap = (std::max(ctx->prog_consts_cap, 1) + 1) & ~1;   // even: code stays 16-B aligned
#define GF_LAUNCH(D, L, Q, B, O)                                                                                              \
    do {                                                                                                                      \
        const size_t lds = ((size_t)n_regs * Q * B + ccap) * sizeof(double) + (size_t)4 * std::max(ctx->prog_len_cap, 1) * sizeof(int32_t); \
        if (lds > 160 * 1024) { ctx->err = "membrane programs need more LDS than a CU has"; return KNP_E_STATE; }             \
        if (lds != ctx->gamma_lds_set)                                                                                        \
            HIPCHK(hipFuncSetAttribute((const void*)k_gamma_facets<D, false, true, L, Q, B, O>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        ctx->gamma_lds_set = lds;                                                                                             \
        hipLaunchKernelGGL((k_gamma_facets<D, false, true, L, Q, B, O>), dim3((unsigned)(((int64_t)g.n_g * L + B - 1) / B)), dim3(B), lds, \
                           ctx->stream, g.n_g, g.n_q, P, ctx->d_fv, ctx->d_fmeas, ctx->d_qp, ctx->d_qw, f, n_aux, ctx->d_coords,  \
                           ctx->d_gamma_prog, (const int32_t* const*)ctx->d_prog_code, (const int32_t*)ctx->d_prog_len,         \
                           (const double* const*)ctx->d_prog_consts, (const int32_t*)ctx->d_prog_nconsts, n_regs, ccap,        \
                           ctx->d_fmat, ctx->d_fvec);                                                                          \
    } while (0)
        // measured on MI355X (cube64: 12 288 facets x 36 points; square512: 1 024 facets x 6 points): 3 points per lane and 16
        // lanes per facet in 3D (258 -> 170 us vs one point per lane), one point per lane in 2D
        void* jit = ctx->jit_fn[g.dim == 2 ? 0 : 1];
        const bool many = gamma_many(g) && ctx->jit_fn[2];
        if (many) jit = ctx->jit_fn[2];
        if (jit) {   // run-time compiled programs (knp_jit.cpp): same kernel source, native mechanism code, no LDS register file
            int n_g = g.n_g, n_q = g.n_q, n_aux_ = n_aux, nr = 0, cc0 = 0;
            DevParams Pp = P;
            FieldPtrs ff = f;
            const int32_t* fv = ctx->d_fv; const double* fmeas = ctx->d_fmeas; const double* qp = ctx->d_qp; const double* qw = ctx->d_qw;
            const double* coords = ctx->d_coords; const int32_t* gp = ctx->d_gamma_prog;
            const int32_t* const* pc = (const int32_t* const*)ctx->d_prog_code; const int32_t* pl = ctx->d_prog_len;
            const double* const* pk = (const double* const*)ctx->d_prog_consts; const int32_t* pn = ctx->d_prog_nconsts;
            double* fmat = ctx->d_fmat; double* fvec = ctx->d_fvec;
            void* args[] = {&n_g, &n_q, &Pp, &fv, &fmeas, &qp, &qw, &ff, &n_aux_, &coords, &gp, &pc, &pl, &pk, &pn, &nr, &cc0, &fmat, &fvec};
            const int L = g.dim == 2 ? 8 : many ? 4 : 16;
            HIPCHK(hipModuleLaunchKernel((hipFunction_t)jit, (unsigned)(((int64_t)g.n_g * L + 63) / 64), 1, 1, 64, 1, 1, 0, ctx->stream, args, nullptr));
        } else if (g.dim == 2) GF_LAUNCH(2, 8, 1, 64, 2);
        else GF_LAUNCH(3, 16, 3, 64, 2);
#undef GF_LAUNCH
    }
    FieldPtrs src;
    for (int j = 0; j < 3; ++j) { src.ki[j] = ctx->src_i[j]; src.ke[j] = ctx->src_e[j]; }
    src.phim = nullptr;
    for (int k = 0; k < KNP_MAX_AUX; ++k) src.aux[k] = nullptr;
#define KNP_RHS(GG) hipLaunchKernelGGL((k_rhs<GG>), dim3(nblocks((int64_t)g.n_nodes_owned * GG)), dim3(NT), 0, ctx->stream, g.n_nodes_owned, g.n_g, g.dim, ctx->dt, \
                                       ctx->d_node_vertex, ctx->d_node_side, ctx->d_pair_ptr, ctx->d_pair_col, ctx->d_pair_M, f, src,                     \
                                       ctx->have_sources ? 1 : 0, ctx->d_node_gv, ctx->d_gdiag, ctx->d_gcptr, ctx->d_gc_facet, ctx->d_gc_lab, ctx->d_fvec, b)
    if (g.n_nodes_owned > 0) {
        switch (ctx->pc_group) {
            case 4: KNP_RHS(4); break;
            case 8: KNP_RHS(8); break;
            case 16: KNP_RHS(16); break;
            default: KNP_RHS(32); break;
        }
    }
#undef KNP_RHS
    launch_schur_diag(ctx, f);   // before a possible knp_gmres_prepare forks the side stream (it reads d_cc)
    HIPCHK(hipGetLastError());
    return KNP_OK;
}

// ---- reductions: partial blocks -> d_red[slot], then (multi-GPU) all-reduce of a slot range ----
static int allreduce_slots(knp_ctx* ctx, int slot0, int count) {
    ++ctx->n_allreduce;   // counted on one GPU too: it is the number of reductions the algorithm needs
    if (ctx->p2p_red >= 0)   // native exchange: the summing kernel also fills the pinned mirror and publishes the sequence word
        return knp_p2p_allreduce(ctx, ctx->p2p_red, ctx->d_red + slot0, count, ctx->h_red_dev ? ctx->h_red_dev + slot0 : nullptr,
                                 ctx->h_red_dev ? ctx->h_seq_dev : nullptr, ++ctx->seq_counter);
    if (ctx->allreduce) {
        int rc = ctx->allreduce(ctx->comm_user, ctx->d_red + slot0, count);
        if (rc != 0) { ctx->err = "allreduce hook failed"; return KNP_E_STATE; }
    }
    return KNP_OK;
}
static int halo_update(knp_ctx* ctx, double* x) {
    ++ctx->n_halo;
    if (ctx->p2p_fine >= 0) return knp_p2p_halo_forward(ctx, ctx->p2p_fine, x);
    if (ctx->halo) {
        int rc = ctx->halo(ctx->comm_user, x);
        if (rc != 0) { ctx->err = "halo hook failed"; return KNP_E_STATE; }
    }
    return KNP_OK;
}
static int dot_to_slot(knp_ctx* ctx, const double* a, const double* b, int slot) {
    const int nb = ctx->n_red_blocks;
    hipLaunchKernelGGL(k_dot, dim3(nb), dim3(NT), 0, ctx->stream, ctx->n_dof_owned, a, b, ctx->d_partial);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(NT), 0, ctx->stream, nb, ctx->d_partial, ctx->d_red, slot, ctx->mirror(),
                       ctx->mirror() ? ctx->h_seq_dev : nullptr, ++ctx->seq_counter);
    return allreduce_slots(ctx, slot, 1);
}
static int read_slots_inner(knp_ctx* ctx, int slot0, int count, int64_t wait_seq);
static int read_slots(knp_ctx* ctx, int slot0, int count, int64_t wait_seq = 0) {
    ++ctx->n_readback;
    KCHK(read_slots_inner(ctx, slot0, count, wait_seq));
    return ctx->p2p ? knp_p2p_check(ctx) : KNP_OK;   // a peer that never arrived: stop here, not after max_it iterations
}
static int read_slots_inner(knp_ctx* ctx, int slot0, int count, int64_t wait_seq) {
    if (ctx->hook_allreduce() || !ctx->h_red_dev) {  // reduced over ranks in d_red: fetch; else the kernel already wrote the mirror
        HIPCHK(hipMemcpyAsync(ctx->h_red + slot0, ctx->d_red + slot0, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        return KNP_OK;
    }
    if (wait_seq > 0 && ctx->h_seq && ctx->h_seq_dev) {
        // spin on the pinned sequence word the last reduction kernel publishes (a few microseconds instead of a
        // full stream synchronisation); bounded: fall back to the synchronisation after ~50 ms
        for (int64_t spin = 0; spin < (int64_t)2000000; ++spin) {
            if (__atomic_load_n(ctx->h_seq, __ATOMIC_ACQUIRE) >= wait_seq) return KNP_OK;
            __builtin_ia32_pause();
        }
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return KNP_OK;
}

static int64_t global_phi_count(knp_ctx* ctx, int* rc) {
    // number of potential DoFs over all ranks (uses slot 63); cached (collective on first use)
    *rc = KNP_OK;
    if (ctx->phi_count_cached >= 0) return ctx->phi_count_cached;
    double v = (double)ctx->g.n_nodes_owned;
    if (hipMemcpyAsync(ctx->d_red + 63, &v, sizeof(double), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { *rc = KNP_E_HIP; return 0; }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { *rc = KNP_E_HIP; return 0; }
    *rc = allreduce_slots(ctx, 63, 1);
    if (*rc != KNP_OK) return 0;
    if (hipMemcpy(ctx->h_red + 63, ctx->d_red + 63, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { *rc = KNP_E_HIP; return 0; }
    ctx->phi_count_cached = (int64_t)llround(ctx->h_red[63]);
    return ctx->phi_count_cached;
}

// SUM over the ranks of a few host values (setup-time decisions that every rank must take identically); slots 104.. of d_red.
// Collective on distributed contexts, the identity on one GPU.
static int global_sum_small(knp_ctx* ctx, double* vals, int n) {
    constexpr int SLOT0 = 104;
    if (n < 0 || SLOT0 + n > RED_SLOTS) { ctx->err = "global_sum_small: too many values"; return KNP_E_ARG; }
    if (n == 0 || !(ctx->allreduce || ctx->p2p_red >= 0)) return KNP_OK;
    HIPCHK(hipMemcpyAsync(ctx->d_red + SLOT0, vals, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    KCHK(allreduce_slots(ctx, SLOT0, n));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(hipMemcpy(vals, ctx->d_red + SLOT0, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    return ctx->p2p ? knp_p2p_check(ctx) : KNP_OK;
}

int knp_set_nullspace(knp_ctx* ctx, int32_t on) {
    CHECK_CTX(ctx);
    side_discard(ctx);
    ctx->ns_on = on ? 1 : 0;
    return KNP_OK;
}

static int project_ns(knp_ctx* ctx, double* v) {
    int rc;
    const int64_t cnt = global_phi_count(ctx, &rc);
    KCHK(rc);
    if (cnt <= 0) return KNP_OK;
    const int no = ctx->g.n_nodes_owned;
    const int nb = std::min(RED_BLOCKS, nblocks(no));
    hipLaunchKernelGGL(k_phi_sum, dim3(nb), dim3(NT), 0, ctx->stream, no, v, ctx->d_partial);
    hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(NT), 0, ctx->stream, nb, ctx->d_partial, ctx->d_red, 62, (double*)nullptr);
    KCHK(allreduce_slots(ctx, 62, 1));
    hipLaunchKernelGGL(k_phi_sub, dim3(nblocks(no)), dim3(NT), 0, ctx->stream, no, ctx->d_red + 62, 1.0 / (double)cnt, v);
    HIPCHK(hipGetLastError());
    return KNP_OK;
}
int knp_project_nullspace(knp_ctx* ctx, double* v) {
    CHECK_CTX(ctx);
    side_discard(ctx);
    if (!v) return KNP_E_ARG;
    return project_ns(ctx, v);
}

static int ensure_work(knp_ctx* ctx, int restart) {
    if (!ctx->d_w) {
        HIPCHK(hipMalloc((void**)&ctx->d_w, std::max(ctx->n_dof_local, 1) * sizeof(double)));
        HIPCHK(hipMalloc((void**)&ctx->d_t, std::max(ctx->n_dof_local, 1) * sizeof(double)));
        HIPCHK(hipMemset(ctx->d_w, 0, std::max(ctx->n_dof_local, 1) * sizeof(double)));
        HIPCHK(hipMemset(ctx->d_t, 0, std::max(ctx->n_dof_local, 1) * sizeof(double)));
    }
    if (restart > 0 && (ctx->gm_restart < restart || !ctx->d_V)) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        dev_free(ctx->d_V);
        size_t bytes = (size_t)(restart + 1) * std::max(ctx->n_dof_local, 1) * sizeof(double);
        HIPCHK(hipMalloc((void**)&ctx->d_V, bytes));
        HIPCHK(hipMemset(ctx->d_V, 0, bytes));
        ctx->gm_restart = restart;
    }
    return KNP_OK;
}

static int spmv_A(knp_ctx* ctx, double* x, const double* b, double* y, bool residual) {
    const KnpHostGraph& g = ctx->g;
    KCHK(join_asm(ctx));
    hipEvent_t ea = nullptr, eb = nullptr;
    if (ctx->prof_on & 1) {   // class 0: events tied to the kernel's begin / end (recycled: no event creation in the timed loop)
        ea = prof_event(ctx);
        eb = prof_event(ctx);
        if (!ea || !eb) ea = eb = nullptr;
    }
    static const bool no_split = getenv("KNP_SPMV_SPLIT") && atoi(getenv("KNP_SPMV_SPLIT")) == 0;
    if (ctx->p2p_fine >= 0 && ctx->n_bnd > 0 && ctx->n_int > 0 && !no_split) {
        // native halo on its own stream while the interior rows run; the boundary rows follow the join ("overlap with
        // interior-row SpMV, then boundary-row SpMV").  The halo kernel occupies at most one block per CU while it waits.
        if (!ctx->stream3) {
            HIPCHK(hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_x, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&ctx->ev_halo, hipEventDisableTiming));
        }
        ++ctx->n_halo;
        HIPCHK(hipEventRecord(ctx->ev_x, ctx->stream));
        HIPCHK(hipStreamWaitEvent(ctx->stream3, ctx->ev_x, 0));
        hipStream_t main_stream = ctx->stream;
        ctx->stream = ctx->stream3;
        const int rc = knp_p2p_halo_forward(ctx, ctx->p2p_fine, x);
        ctx->stream = main_stream;
        KCHK(rc);
        HIPCHK(hipEventRecord(ctx->ev_halo, ctx->stream3));
        if (residual) launch_spmv_node<1>(ctx, ctx->n_int, ctx->d_nodes_int, x, b, y, ea, nullptr);
        else launch_spmv_node<0>(ctx, ctx->n_int, ctx->d_nodes_int, x, b, y, ea, nullptr);
        HIPCHK(hipStreamWaitEvent(ctx->stream, ctx->ev_halo, 0));
        if (residual) launch_spmv_node<1>(ctx, ctx->n_bnd, ctx->d_nodes_bnd, x, b, y, nullptr, eb);
        else launch_spmv_node<0>(ctx, ctx->n_bnd, ctx->d_nodes_bnd, x, b, y, nullptr, eb);
    } else {
        KCHK(halo_update(ctx, x));
        if (residual) launch_spmv_node<1>(ctx, g.n_nodes_owned, nullptr, x, b, y, ea, eb);
        else launch_spmv_node<0>(ctx, g.n_nodes_owned, nullptr, x, b, y, ea, eb);
    }
    if (ea && eb) ctx->prof_recs.push_back({ea, eb, 0});
    HIPCHK(hipGetLastError());
    return KNP_OK;
}

int knp_spmv(knp_ctx* ctx, const double* x, double* y) {
    CHECK_CTX(ctx);
    if (!x || !y) return KNP_E_ARG;
    if (!ctx->have_A) { ctx->err = "matrix not assembled"; return KNP_E_STATE; }
    return spmv_A(ctx, const_cast<double*>(x), nullptr, y, false);
}

int knp_nullspace_test(knp_ctx* ctx, double* out_norm) {
    CHECK_CTX(ctx);
    side_discard(ctx);
    if (!out_norm) return KNP_E_ARG;
    if (!ctx->have_A) { ctx->err = "matrix not assembled"; return KNP_E_STATE; }
    KCHK(ensure_work(ctx, 0));
    int rc;
    const int64_t cnt = global_phi_count(ctx, &rc);
    KCHK(rc);
    hipLaunchKernelGGL(k_fill_phi, dim3(nblocks(ctx->g.n_nodes)), dim3(NT), 0, ctx->stream, ctx->g.n_nodes,
                       1.0 / std::sqrt((double)std::max<int64_t>(cnt, 1)), ctx->d_w);
    KCHK(spmv_A(ctx, ctx->d_w, nullptr, ctx->d_t, false));
    KCHK(dot_to_slot(ctx, ctx->d_t, ctx->d_t, 61));
    KCHK(read_slots(ctx, 61, 1));
    *out_norm = std::sqrt(ctx->h_red[61]);
    return KNP_OK;
}

// ---- AMG ---------------------------------------------------------------------------------
static void free_blocked(KnpBlockedCsr& M) {
    dev_free(M.rp); dev_free(M.ev); dev_free(M.ci);
    M.n_rows = 0;
}
// Node-blocked copy of a scalar CSR whose rows come in groups of rs per row node (fields 0..nf-1 used, the others empty) and whose
// columns are cs*node + field with the field of the row (decoupled fields, sorted rows).  The column nodes of a node row are the
// UNION over its fields, absent entries stored as zeros (identical patterns by construction for the ion fields; the potential
// additionally couples the two sides of a membrane vertex).  The structure is VERIFIED here; a matrix that does not have it, or
// whose union would add more than a quarter to the stored values, is left without a blocked copy (out->rp == nullptr) and the
// cycle stays on the scalar kernels.
static int build_blocked(knp_ctx* ctx, int nf, int n_rows_scalar, int rs, int cs, const int32_t* rp, const int32_t* ci, const double* v,
                         KnpBlockedCsr* out, bool restrictor = false) {
    free_blocked(*out);
    if ((nf != 3 && nf != 4) || rs < nf || cs < nf || n_rows_scalar <= 0 || n_rows_scalar % rs != 0) return KNP_OK;
    const int nn = n_rows_scalar / rs;
    // two passes over the node rows, both in parallel (the level-0 operators of a 10^7-unknown problem have ~10^8 entries: the
    // single-threaded merge with push_back was most of the hierarchy upload): count + verify, prefix sum, fill
    std::vector<int32_t> brp((size_t)nn + 1, 0);
    std::vector<int32_t> cnt((size_t)nn, 0);
    int bad = 0;
    // merge of the nf sorted field rows of node row i by column node; emit(jmin, val) per union entry; false: structure violated
    auto walk = [&](int i, auto&& emit) -> bool {
        const int r0 = rs * i;
        for (int k = nf; k < rs; ++k)
            if (rp[r0 + k + 1] != rp[r0 + k]) return false;
        int64_t q[4], e[4];
        for (int k = 0; k < nf; ++k) { q[k] = rp[r0 + k]; e[k] = rp[r0 + k + 1]; }
        for (;;) {
            int jmin = INT32_MAX;
            for (int k = 0; k < nf; ++k)
                if (q[k] < e[k]) {
                    const int c = ci[q[k]];
                    if (c % cs != k) return false;                                   // a field couples to another one
                    if (q[k] > rp[r0 + k] && ci[q[k] - 1] >= c) return false;        // row not sorted
                    jmin = std::min(jmin, c / cs);
                }
            if (jmin == INT32_MAX) break;
            float val[4] = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < nf; ++k)
                if (q[k] < e[k] && ci[q[k]] / cs == jmin) val[k] = (float)v[q[k]++];
            emit(jmin, val);
        }
        return true;
    };
#pragma omp parallel for schedule(static) reduction(+ : bad) num_threads(knp_host_threads())
    for (int i = 0; i < nn; ++i) {
        int c = 0;
        if (!walk(i, [&](int, const float*) { ++c; })) ++bad;
        cnt[(size_t)i] = c;
    }
    if (bad) return KNP_OK;
    int64_t total = 0;
    for (int i = 0; i < nn; ++i) {
        total += cnt[(size_t)i];
        if (total > (int64_t)INT32_MAX) return KNP_OK;
        brp[(size_t)i + 1] = (int32_t)total;
    }
    HostBuf<float4> ev((size_t)total);
    HostBuf<int32_t> bci(nf == 4 ? (size_t)total : 0);
    if (!ev.p || !bci.p) { ctx->err = "out of host memory"; return KNP_E_STATE; }
#pragma omp parallel for schedule(static) num_threads(knp_host_threads())
    for (int i = 0; i < nn; ++i) {
        size_t o = (size_t)brp[(size_t)i];
        (void)walk(i, [&](int jmin, const float* val) {
            float4 t;
            t.x = val[0]; t.y = val[1]; t.z = val[2];
            if (nf == 4) { t.w = val[3]; bci[o] = jmin; }
            else { const int32_t jj = jmin; float w; memcpy(&w, &jj, 4); t.w = w; }
            ev[o++] = t;
        });
    }
    const int64_t nnz = (int64_t)ev.size();
    if ((double)nnz * nf > 1.25 * (double)rp[n_rows_scalar] + 64.0) return KNP_OK;
    KCHK(dev_upload(ctx, &out->rp, brp));
    KCHK(dev_upload(ctx, &out->ev, ev));
    if (nf == 4) KCHK(dev_upload(ctx, &out->ci, bci));
    out->n_rows = nn;
    out->nnz = nnz;
    static const double scale = getenv("KNP_LANE_SCALE_B") ? atof(getenv("KNP_LANE_SCALE_B")) : 1.0;
    // four entries in flight per lane; measured on MI355X (cube 136^3 / 512^2): about 4 entries per lane for S and the level
    // operators, 2 for the restrictors (long rows whose gathers of the fine residual miss the caches more often)
    const double avg = scale * (restrictor ? 2.0 : 1.0) * (double)nnz / nn;
    out->lanes = avg <= 10.0 ? 2 : avg <= 20.0 ? 4 : avg <= 44.0 ? 8 : avg <= 100.0 ? 16 : 32;
    return KNP_OK;
}
int knp_amg_set_node_fields(knp_ctx* ctx, int32_t hier, int32_t nf) {
    CHECK_CTX(ctx);
    if (hier < 0 || hier >= KNP_MAX_HIER || (nf != 0 && nf != 3 && nf != 4)) { ctx->err = "node fields: 0 (off), 3 or 4"; return KNP_E_ARG; }
    KnpAmgHier& H = ctx->hier[hier];
    if (H.levels < 1 || H.lv[0].n > 0) { ctx->err = "knp_amg_set_node_fields comes after knp_amg_reset and before the levels"; return KNP_E_STATE; }
    H.node_nf = nf;
    return KNP_OK;
}
static void free_hier(KnpAmgHier& H) {
    for (int l = 0; l < KNP_MAX_AMG_LEVELS; ++l) {
        KnpAmgLevel& L = H.lv[l];
        dev_free(L.A_rp); dev_free(L.A_ci); dev_free(L.A_v); dev_free(L.inv_diag);
        dev_free(L.P_rp); dev_free(L.P_ci); dev_free(L.P_v); dev_free(L.R_rp); dev_free(L.R_ci); dev_free(L.R_v);
        dev_free(L.A_vf); dev_free(L.P_vf); dev_free(L.R_vf); dev_free(L.P_act_rows); dev_free(L.P_act_rp); L.P_n_act = 0;
        dev_free(L.x); dev_free(L.b); dev_free(L.r); dev_free(L.d); dev_free(L.r2);
        dev_free(L.xs); dev_free(L.bs); dev_free(L.rs); dev_free(L.ds); dev_free(L.r2s);
        dev_free(L.S_rp); dev_free(L.S_ci); dev_free(L.S_v); dev_free(L.S_vf); dev_free(L.S_act_rows); dev_free(L.S_act_rp);
        dev_free(L.R_ci_c); dev_free(L.S_act_rows_c); dev_free(L.dinv_c);
        free_blocked(L.bA); free_blocked(L.bR); free_blocked(L.bS);
        dev_free(L.Rt_rp); dev_free(L.Rt_ci); dev_free(L.Rt_v); dev_free(L.Rt_vf); dev_free(L.U_rp); dev_free(L.U_ci); dev_free(L.U_v); dev_free(L.U_vf);
        free_blocked(L.bRt); free_blocked(L.bU); dev_free(L.cat); dev_free(L.cats);
        dev_free(L.At_v); dev_free(L.At_vf); L.lfused = 0; L.A_nnz = 0;
        L.S_rows = L.S_n_act = 0;
        L.n = L.n_coarse = 0;
    }
    dev_free(H.cinv); dev_free(H.cinv_f);
    dev_free(H.pt); dev_free(H.pt_phi); dev_free(H.pt_f); dev_free(H.pt_phi_f);
    dev_free(H.at0_rp); dev_free(H.at0_ci); dev_free(H.at0_v); dev_free(H.at0_vf); H.l0_upload = 0;
    H.nc = 0; H.levels = 0; H.native0 = 0; H.fused = 0; H.node_nf = 0; H.blocked = 0; H.cfused = 0;
}
int knp_amg_reset(knp_ctx* ctx, int32_t hier, int32_t n_levels, int32_t pre, int32_t post, int32_t cheby) {
    CHECK_CTX(ctx);
    side_discard(ctx);
    if (hier < 0 || hier >= KNP_MAX_HIER || n_levels < 1 || n_levels > KNP_MAX_AMG_LEVELS || pre < 0 || post < 0 || cheby < 1) { ctx->err = "bad AMG parameters"; return KNP_E_ARG; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    KnpAmgHier& H = ctx->hier[hier];
    free_hier(H);
    H.levels = n_levels; H.pre = pre; H.post = post; H.cheby = cheby;
    ctx->side_ws = false;      // the new levels need their own second set of work vectors (ensure_side_ws)
    return KNP_OK;
}
// compact row list of a prolongator when a good part of its rows is empty
// ``keep`` (optional, one per row): rows that must stay in the list although their matrix row is empty -- for S these are the
// unknowns the smoother solves on its own (non-zero inverse diagonal, no interpolation: decoupled 1x1 blocks, Dirichlet rows)
static int build_act_rows(knp_ctx* ctx, int n_rows, const int32_t* rp_in, int32_t** d_rows, int32_t** d_rp, int* n_act,
                          const double* keep = nullptr) {
    dev_free(*d_rows); dev_free(*d_rp);
    *n_act = 0;
    std::vector<int32_t> rows, rp(1, 0);
    for (int r = 0; r < n_rows; ++r)
        if (rp_in[r + 1] > rp_in[r] || (keep && keep[r] != 0.0)) { rows.push_back(r); rp.push_back(rp_in[r + 1]); }
    if ((double)rows.size() > 0.8 * n_rows || rows.empty()) return KNP_OK;   // dense enough: the plain kernel is fine
    // non-empty rows are contiguous in the value array only if no empty row lies between entries -- they are: CSR rows are
    // consecutive, empty rows contribute nothing, so rp of the compact list is the running end pointer
    KCHK(dev_upload(ctx, d_rows, rows));
    KCHK(dev_upload(ctx, d_rp, rp));
    *n_act = (int)rows.size();
    return KNP_OK;
}
static int build_prolong_rows(knp_ctx* ctx, KnpAmgLevel& L, int n_rows_P, const int32_t* P_rp) {
    return build_act_rows(ctx, n_rows_P, P_rp, &L.P_act_rows, &L.P_act_rp, &L.P_n_act);
}

int knp_amg_set_level(knp_ctx* ctx, int32_t hier, int32_t level, int32_t n_rows, int32_t n_cols_halo, const int32_t* A_rp,
                      const int32_t* A_ci, const double* A_v, const double* inv_diag, double lambda_max, int32_t n_coarse,
                      const int32_t* P_rp, const int32_t* P_ci, const double* P_v, const int32_t* R_rp, const int32_t* R_ci,
                      const double* R_v) {
    CHECK_CTX(ctx);
    if (hier < 0 || hier >= KNP_MAX_HIER) { ctx->err = "bad hierarchy index"; return KNP_E_ARG; }
    KnpAmgHier& H = ctx->hier[hier];
    if (level < 0 || level >= H.levels || n_rows <= 0 || !A_rp || !A_ci || !A_v || !inv_diag) { ctx->err = "bad AMG level arguments"; return KNP_E_ARG; }
    if (level == 0 && n_rows != ctx->n_dof_owned) { ctx->err = "AMG level 0 must have n_dof_owned rows"; return KNP_E_ARG; }
    KnpAmgLevel& L = H.lv[level];
    const int64_t nnzA = A_rp[n_rows];
    const int n_loc = std::max(n_cols_halo, n_rows);
    if (!cols_in_range(A_ci, nnzA, n_loc)) { ctx->err = "AMG level matrix column out of range"; return KNP_E_ARG; }
    L.n = n_rows; L.n_loc = n_loc; L.n_coarse = n_coarse; L.lambda_max = lambda_max; L.A_nnz = nnzA;
    KCHK(dev_upload_raw(ctx, &L.A_rp, A_rp, (size_t)n_rows + 1));
    KCHK(dev_upload_raw(ctx, &L.A_ci, A_ci, (size_t)nnzA));
    if (ctx->amg_fp32) {   // one copy only: fp32 when the preconditioner is stored in mixed precision
        HostBuf<float> tmp = to_float(A_v, nnzA);
        KCHK(dev_upload(ctx, &L.A_vf, tmp));
    } else {
        KCHK(dev_upload_raw(ctx, &L.A_v, A_v, (size_t)nnzA));
    }
    KCHK(dev_upload_raw(ctx, &L.inv_diag, inv_diag, (size_t)n_rows));
    L.A_lanes = pick_lanes((double)nnzA / n_rows);
    if (n_coarse > 0) {
        if (!P_rp || !P_ci || !P_v || !R_rp || !R_ci || !R_v) { ctx->err = "AMG transfer operators missing"; return KNP_E_ARG; }
        const int64_t nnzP = P_rp[n_rows], nnzR = R_rp[n_coarse];
        if (!cols_in_range(P_ci, nnzP, n_coarse)) { ctx->err = "AMG prolongator column out of range"; return KNP_E_ARG; }
        if (!cols_in_range(R_ci, nnzR, n_rows)) { ctx->err = "AMG restrictor column out of range"; return KNP_E_ARG; }
        KCHK(dev_upload_raw(ctx, &L.P_rp, P_rp, (size_t)n_rows + 1));
        KCHK(dev_upload_raw(ctx, &L.P_ci, P_ci, (size_t)nnzP));
        KCHK(dev_upload_raw(ctx, &L.R_rp, R_rp, (size_t)n_coarse + 1));
        KCHK(dev_upload_raw(ctx, &L.R_ci, R_ci, (size_t)nnzR));
        if (ctx->amg_fp32) {
            HostBuf<float> tp = to_float(P_v, nnzP), tr = to_float(R_v, nnzR);
            KCHK(dev_upload(ctx, &L.P_vf, tp));
            KCHK(dev_upload(ctx, &L.R_vf, tr));
        } else {
            KCHK(dev_upload_raw(ctx, &L.P_v, P_v, (size_t)nnzP));
            KCHK(dev_upload_raw(ctx, &L.R_v, R_v, (size_t)nnzR));
        }
        L.P_lanes = pick_lanes((double)nnzP / n_rows, 1);
        L.P_rows = n_rows;
        KCHK(build_prolong_rows(ctx, L, n_rows, P_rp));
        if (L.P_n_act > 0) L.P_lanes = pick_lanes((double)nnzP / L.P_n_act, 1);
        L.R_lanes = pick_lanes((double)nnzR / n_coarse, 2);
        L.R_nnz = nnzR;
        if (H.node_nf > 0 && ctx->amg_fp32) KCHK(build_blocked(ctx, H.node_nf, n_coarse, H.node_nf, level == 0 ? 4 : H.node_nf, R_rp, R_ci, R_v, &L.bR, true));
    }
    if (H.node_nf > 0 && ctx->amg_fp32 && level > 0) KCHK(build_blocked(ctx, H.node_nf, n_rows, H.node_nf, H.node_nf, A_rp, A_ci, A_v, &L.bA));
    HIPCHK(hipMalloc((void**)&L.x, (size_t)n_loc * sizeof(double)));
    HIPCHK(hipMalloc((void**)&L.b, (size_t)n_loc * sizeof(double)));
    HIPCHK(hipMemset(L.b, 0, (size_t)n_loc * sizeof(double)));
    HIPCHK(hipMalloc((void**)&L.r, (size_t)n_loc * sizeof(double)));
    HIPCHK(hipMalloc((void**)&L.d, (size_t)n_loc * sizeof(double)));
    HIPCHK(hipMalloc((void**)&L.r2, (size_t)n_loc * sizeof(double)));
    HIPCHK(hipMemset(L.r2, 0, (size_t)n_loc * sizeof(double)));
    HIPCHK(hipMemset(L.x, 0, (size_t)n_loc * sizeof(double)));
    HIPCHK(hipMemset(L.d, 0, (size_t)n_loc * sizeof(double)));
    return KNP_OK;
}
int knp_amg_set_level_prolongator(knp_ctx* ctx, int32_t hier, int32_t level, int32_t n_rows_P, const int32_t* P_rp, const int32_t* P_ci,
                                  const double* P_v) {
    CHECK_CTX(ctx);
    if (hier < 0 || hier >= KNP_MAX_HIER || level < 0 || level >= ctx->hier[hier].levels) { ctx->err = "bad arguments"; return KNP_E_ARG; }
    KnpAmgLevel& L = ctx->hier[hier].lv[level];
    if (L.n_coarse <= 0 || !P_rp || !P_ci || !P_v || n_rows_P < L.n || n_rows_P > L.n_loc) { ctx->err = "prolongator rows must cover the owned entries and at most the local ones"; return KNP_E_ARG; }
    const int64_t nnzP = P_rp[n_rows_P];
    if (!cols_in_range(P_ci, nnzP, L.n_coarse)) { ctx->err = "AMG prolongator column out of range"; return KNP_E_ARG; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dev_free(L.P_rp); dev_free(L.P_ci); dev_free(L.P_v); dev_free(L.P_vf);
    KCHK(dev_upload_raw(ctx, &L.P_rp, P_rp, (size_t)n_rows_P + 1));
    KCHK(dev_upload_raw(ctx, &L.P_ci, P_ci, (size_t)nnzP));
    if (ctx->amg_fp32) {
        HostBuf<float> tp = to_float(P_v, nnzP);
        KCHK(dev_upload(ctx, &L.P_vf, tp));
    } else {
        KCHK(dev_upload_raw(ctx, &L.P_v, P_v, (size_t)nnzP));
    }
    L.P_rows = n_rows_P;
    L.P_lanes = pick_lanes((double)nnzP / std::max(n_rows_P, 1), 1);
    KCHK(build_prolong_rows(ctx, L, n_rows_P, P_rp));
    if (L.P_n_act > 0) L.P_lanes = pick_lanes((double)nnzP / L.P_n_act, 1);
    return KNP_OK;
}
int knp_amg_set_level_smoothed(knp_ctx* ctx, int32_t hier, int32_t level, int32_t n_rows, const int32_t* S_rp, const int32_t* S_ci, const double* S_v) {
    CHECK_CTX(ctx);
    if (hier < 0 || hier >= KNP_MAX_HIER || level < 0 || level >= ctx->hier[hier].levels) { ctx->err = "bad arguments"; return KNP_E_ARG; }
    KnpAmgLevel& L = ctx->hier[hier].lv[level];
    if (L.n_coarse <= 0 || !S_rp || !S_ci || !S_v || n_rows != L.n) { ctx->err = "S needs a level with a coarser level below it and one row per level row"; return KNP_E_ARG; }
    const int64_t nnzS = S_rp[n_rows];
    if (!cols_in_range(S_ci, nnzS, L.n_coarse)) { ctx->err = "S column out of range"; return KNP_E_ARG; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dev_free(L.S_rp); dev_free(L.S_ci); dev_free(L.S_v); dev_free(L.S_vf);
    KCHK(dev_upload_raw(ctx, &L.S_rp, S_rp, (size_t)n_rows + 1));
    KCHK(dev_upload_raw(ctx, &L.S_ci, S_ci, (size_t)nnzS));
    if (ctx->amg_fp32) {
        HostBuf<float> t = to_float(S_v, nnzS);
        KCHK(dev_upload(ctx, &L.S_vf, t));
    } else {
        KCHK(dev_upload_raw(ctx, &L.S_v, S_v, (size_t)nnzS));
    }
    L.S_rows = n_rows;
    L.S_nnz = nnzS;
    std::vector<double> dinv_host((size_t)n_rows);
    HIPCHK(hipMemcpy(dinv_host.data(), L.inv_diag, (size_t)n_rows * sizeof(double), hipMemcpyDeviceToHost));
    KCHK(build_act_rows(ctx, n_rows, S_rp, &L.S_act_rows, &L.S_act_rp, &L.S_n_act, dinv_host.data()));
    L.S_lanes = pick_lanes((double)nnzS / std::max(L.S_n_act > 0 ? L.S_n_act : n_rows, 1), 1);
    const int nf = ctx->hier[hier].node_nf;
    if (nf > 0 && ctx->amg_fp32) KCHK(build_blocked(ctx, nf, n_rows, level == 0 ? 4 : nf, nf, S_rp, S_ci, S_v, &L.bS));
    return KNP_OK;
}
// Intermediate level of the fused cycle as two plain products (cgx_hip/amg.py coarse_fused_operators): Rt [n_coarse x n] and
// U [n x (n + n_coarse)], CSR with sorted columns.  Optional: without them the level runs restriction + residual and the S up-leg.
int knp_amg_set_level_coarse_fused(knp_ctx* ctx, int32_t hier, int32_t level, int32_t Rt_rows, const int32_t* Rt_rp, const int32_t* Rt_ci,
                                   const double* Rt_v, int32_t U_rows, const int32_t* U_rp, const int32_t* U_ci, const double* U_v) {
    CHECK_CTX(ctx);
    if (hier < 0 || hier >= KNP_MAX_HIER || level < 1 || level >= ctx->hier[hier].levels - 1) { ctx->err = "coarse-fused operators belong to an intermediate level"; return KNP_E_ARG; }
    KnpAmgLevel& L = ctx->hier[hier].lv[level];
    if (!Rt_rp || !Rt_ci || !Rt_v || !U_rp || !U_ci || !U_v || L.n_coarse <= 0 || Rt_rows != L.n_coarse || U_rows != L.n || L.dist || L.n_loc != L.n) {
        ctx->err = "coarse-fused operators: Rt is [n_coarse x n], U is [n x (n + n_coarse)], single-GPU levels only";
        return KNP_E_ARG;
    }
    const int64_t nnzR = Rt_rp[Rt_rows], nnzU = U_rp[U_rows];
    if (!cols_in_range(Rt_ci, nnzR, L.n)) { ctx->err = "Rt column out of range"; return KNP_E_ARG; }
    if (!cols_in_range(U_ci, nnzU, L.n + L.n_coarse)) { ctx->err = "U column out of range"; return KNP_E_ARG; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    dev_free(L.Rt_rp); dev_free(L.Rt_ci); dev_free(L.Rt_v); dev_free(L.Rt_vf); dev_free(L.U_rp); dev_free(L.U_ci); dev_free(L.U_v); dev_free(L.U_vf);
    KCHK(dev_upload_raw(ctx, &L.Rt_rp, Rt_rp, (size_t)Rt_rows + 1));
    KCHK(dev_upload_raw(ctx, &L.Rt_ci, Rt_ci, (size_t)nnzR));
    KCHK(dev_upload_raw(ctx, &L.U_rp, U_rp, (size_t)U_rows + 1));
    KCHK(dev_upload_raw(ctx, &L.U_ci, U_ci, (size_t)nnzU));
    if (ctx->amg_fp32) {
        HostBuf<float> t = to_float(Rt_v, nnzR), u = to_float(U_v, nnzU);
        KCHK(dev_upload(ctx, &L.Rt_vf, t));
        KCHK(dev_upload(ctx, &L.U_vf, u));
    } else {
        KCHK(dev_upload_raw(ctx, &L.Rt_v, Rt_v, (size_t)nnzR));
        KCHK(dev_upload_raw(ctx, &L.U_v, U_v, (size_t)nnzU));
    }
    L.Rt_lanes = pick_lanes((double)nnzR / std::max(Rt_rows, 1), 2);
    L.Rt_nnz = nnzR; L.U_nnz = nnzU;
    L.U_lanes = pick_lanes((double)nnzU / std::max(U_rows, 1), 0);
    const int nf = ctx->hier[hier].node_nf;
    if (nf > 0 && ctx->amg_fp32) {
        KCHK(build_blocked(ctx, nf, Rt_rows, nf, nf, Rt_rp, Rt_ci, Rt_v, &L.bRt, true));
        KCHK(build_blocked(ctx, nf, U_rows, nf, nf, U_rp, U_ci, U_v, &L.bU));
    }
    return KNP_OK;
}
int knp_amg_set_level_mode(knp_ctx* ctx, int32_t hier, int32_t level, int32_t distributed, int32_t repl_n) {
    CHECK_CTX(ctx);
    if (hier < 0 || hier >= KNP_MAX_HIER || level < 0 || level >= ctx->hier[hier].levels || repl_n < 0) { ctx->err = "bad arguments"; return KNP_E_ARG; }
    ctx->hier[hier].lv[level].dist = distributed ? 1 : 0;
    ctx->hier[hier].lv[level].repl_n = repl_n;
    return KNP_OK;
}
int knp_set_level_comm(knp_ctx* ctx, knp_level_comm_fn fn) {
    CHECK_CTX(ctx);
    ctx->level_comm = fn;
    return KNP_OK;
}
int knp_amg_set_precision(knp_ctx* ctx, int32_t fp32_storage) {
    CHECK_CTX(ctx);
    ctx->amg_fp32 = fp32_storage ? 1 : 0;   // takes effect for hierarchies uploaded afterwards
    return KNP_OK;
}
// level-0 data of the fused cycle: Pt = P Dinv with the hierarchy's own inverse diagonal (zero on the fields it does not act on);
// potential-only hierarchies additionally get node-indexed copies of the index arrays that refer to level-0 rows
__global__ void __launch_bounds__(NT) k_gather_rowptr4(int n, const int32_t* __restrict__ rp, int off, int32_t* __restrict__ out) {
    for (int i = blockIdx.x * NT + threadIdx.x; i <= n; i += gridDim.x * NT) out[i] = rp[i < n ? 4 * (size_t)i + off : 4 * (size_t)n];
}
static int build_fused_data(knp_ctx* ctx, KnpAmgHier& H) {
    dev_free(H.pt); dev_free(H.pt_phi); dev_free(H.pt_f); dev_free(H.pt_phi_f);
    dev_free(H.at0_rp); dev_free(H.at0_ci); dev_free(H.at0_v); dev_free(H.at0_vf);
    KnpAmgLevel& L = H.lv[0];
    dev_free(L.R_ci_c); dev_free(L.S_act_rows_c); dev_free(L.dinv_c);
    if (!H.native0 || H.levels < 2 || !L.inv_diag || !L.S_rp) return KNP_OK;
    if (H.l0_upload) {
        // potential hierarchy on its uploaded level-0 operator (rows 4 node + 3 of an n_dof x n_dof CSR, every other row empty): the
        // compact CSR of c A Dinv on node-indexed vectors -- row pointer of the potential rows, columns / 4, values scaled
        if (H.native0 != 3 || ctx->g.n_nodes != ctx->g.n_nodes_owned || !L.A_rp || L.A_nnz <= 0) { ctx->err = "level 0 from the uploaded operator: potential hierarchy on one GPU only"; return KNP_E_STATE; }
        const int nn = ctx->g.n_nodes_owned;
        const double c = 1.0 / (0.5 * (1.1 + 0.1) * L.lambda_max);
        const int nblk = (int)std::min<int64_t>(nblocks(L.A_nnz), 8192);
        HIPCHK(hipMalloc((void**)&H.at0_rp, ((size_t)nn + 1) * sizeof(int32_t)));
        HIPCHK(hipMalloc((void**)&H.at0_ci, (size_t)L.A_nnz * sizeof(int32_t)));
        hipLaunchKernelGGL(k_gather_rowptr4, dim3(std::min(nblocks(nn + 1), 4096)), dim3(NT), 0, ctx->stream, nn, L.A_rp, 3, H.at0_rp);
        hipLaunchKernelGGL(k_shift2, dim3(nblk), dim3(NT), 0, ctx->stream, L.A_nnz, L.A_ci, H.at0_ci);
        if (L.A_vf) {
            HIPCHK(hipMalloc((void**)&H.at0_vf, (size_t)L.A_nnz * sizeof(float)));
            hipLaunchKernelGGL((k_build_at<float, float>), dim3(nblk), dim3(NT), 0, ctx->stream, L.A_nnz, L.A_ci, L.A_vf, L.inv_diag, c, H.at0_vf);
        } else {
            HIPCHK(hipMalloc((void**)&H.at0_v, (size_t)L.A_nnz * sizeof(double)));
            hipLaunchKernelGGL((k_build_at<double, double>), dim3(nblk), dim3(NT), 0, ctx->stream, L.A_nnz, L.A_ci, L.A_v, L.inv_diag, c, H.at0_v);
        }
        H.at0_lanes = pick_lanes((double)L.A_nnz / std::max(nn, 1), 0);
    }
    const int64_t np = ctx->n_pairs;
    const int nblk = (int)std::min<int64_t>(nblocks(np), 8192);
    const bool phi = H.native0 == 3;
    // columns of ghost nodes are scaled with the ghost nodes' inverse diagonal: one halo of it (collective: every rank gets here)
    const double* dinv_loc = L.inv_diag;
    double* dinv_tmp = nullptr;
    if (ctx->g.n_nodes > ctx->g.n_nodes_owned) {
        HIPCHK(hipMalloc((void**)&dinv_tmp, (size_t)ctx->n_dof_local * sizeof(double)));
        HIPCHK(hipMemsetAsync(dinv_tmp, 0, (size_t)ctx->n_dof_local * sizeof(double), ctx->stream));
        HIPCHK(hipMemcpyAsync(dinv_tmp, L.inv_diag, (size_t)ctx->n_dof_owned * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
        KCHK(halo_update(ctx, dinv_tmp));
        dinv_loc = dinv_tmp;
    }
    if (H.l0_upload) {
        // (no pair-major P Dinv: the down-leg runs on at0)
    } else if (ctx->amg_fp32) {
        if (phi) HIPCHK(hipMalloc((void**)&H.pt_phi_f, std::max<int64_t>(np, 1) * sizeof(float)));
        else HIPCHK(hipMalloc((void**)&H.pt_f, std::max<int64_t>(4 * np, 1) * sizeof(float)));
        hipLaunchKernelGGL((k_build_pt<float>), dim3(nblk), dim3(NT), 0, ctx->stream, np, ctx->d_pair_col, ctx->d_p_vals, dinv_loc, H.pt_f, H.pt_phi_f);
    } else {
        if (phi) HIPCHK(hipMalloc((void**)&H.pt_phi, std::max<int64_t>(np, 1) * sizeof(double)));
        else HIPCHK(hipMalloc((void**)&H.pt, std::max<int64_t>(4 * np, 1) * sizeof(double)));
        hipLaunchKernelGGL((k_build_pt<double>), dim3(nblk), dim3(NT), 0, ctx->stream, np, ctx->d_pair_col, ctx->d_p_vals, dinv_loc, H.pt, H.pt_phi);
    }
    // The dense coarse inverse of the ION hierarchy (well conditioned: M + dt D K) is stored in fp32 like the other operators
    // of the mixed-precision preconditioner; the potential hierarchy's stays fp64 (nearly singular, cond ~1e8).
    dev_free(H.cinv_f);
    if (H.native0 == 2 && ctx->amg_fp32 && H.nc > 0 && H.cinv) {
        const int64_t nn2 = (int64_t)H.nc * H.nc;
        HIPCHK(hipMalloc((void**)&H.cinv_f, nn2 * sizeof(float)));
        hipLaunchKernelGGL(k_to_float, dim3(std::min<int64_t>(nblocks(nn2), 4096)), dim3(NT), 0, ctx->stream, nn2, H.cinv, H.cinv_f);
    }
    if (phi) {
        const int nn = ctx->g.n_nodes_owned;
        HIPCHK(hipMalloc((void**)&L.dinv_c, std::max(nn, 1) * sizeof(double)));
        hipLaunchKernelGGL(k_gather_stride4, dim3(std::min(nblocks(nn), 4096)), dim3(NT), 0, ctx->stream, nn, L.inv_diag, 3, L.dinv_c);
        int32_t nnzR = 0;
        HIPCHK(hipMemcpy(&nnzR, L.R_rp + L.n_coarse, sizeof(int32_t), hipMemcpyDeviceToHost));
        HIPCHK(hipMalloc((void**)&L.R_ci_c, std::max<size_t>(nnzR, 1) * sizeof(int32_t)));
        hipLaunchKernelGGL(k_shift2, dim3(std::min<int64_t>(nblocks(nnzR), 4096)), dim3(NT), 0, ctx->stream, (int64_t)nnzR, L.R_ci, L.R_ci_c);
        if (L.S_n_act > 0) {
            HIPCHK(hipMalloc((void**)&L.S_act_rows_c, (size_t)L.S_n_act * sizeof(int32_t)));
            hipLaunchKernelGGL(k_shift2, dim3(std::min<int64_t>(nblocks(L.S_n_act), 4096)), dim3(NT), 0, ctx->stream, (int64_t)L.S_n_act, L.S_act_rows, L.S_act_rows_c);
        }
    }
    HIPCHK(hipGetLastError());
    if (dinv_tmp) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        (void)hipFree(dinv_tmp);
    }
    return KNP_OK;
}

int knp_amg_use_native_level0(knp_ctx* ctx, int32_t hier, int32_t mode) {
    CHECK_CTX(ctx);
    if (hier < 0 || hier >= KNP_MAX_HIER || mode < 0 || mode > 4) { ctx->err = "bad arguments"; return KNP_E_ARG; }
    if (mode && !ctx->have_P) { ctx->err = "P not assembled"; return KNP_E_STATE; }
    ctx->hier[hier].l0_upload = mode == 4 ? 1 : 0;     // mode 4: potential only, level 0 = the uploaded operator (may couple the sides)
    if (mode == 4) mode = 3;
    ctx->hier[hier].native0 = mode;
    KCHK(build_fused_data(ctx, ctx->hier[hier]));
    if (mode && ctx->amg_fp32) {   // fp32 shadow of the pair-major P for the level-0 node kernels
        const int64_t n = 4 * ctx->n_pairs;
        if (!ctx->d_p_vals_f) HIPCHK(hipMalloc((void**)&ctx->d_p_vals_f, std::max<int64_t>(n, 1) * sizeof(float)));
        hipLaunchKernelGGL(k_to_float, dim3(std::min<int64_t>(nblocks(n), 4096)), dim3(NT), 0, ctx->stream, n, ctx->d_p_vals, ctx->d_p_vals_f);
        HIPCHK(hipGetLastError());
    } else if (!ctx->amg_fp32) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        dev_free(ctx->d_p_vals_f);
    }
    return KNP_OK;
}
int knp_amg_set_coarse(knp_ctx* ctx, int32_t hier, int32_t n, const double* inv) {
    CHECK_CTX(ctx);
    if (hier < 0 || hier >= KNP_MAX_HIER || n <= 0 || !inv) return KNP_E_ARG;
    KnpAmgHier& H = ctx->hier[hier];
    dev_free(H.cinv); dev_free(H.cinv_f);
    // the dense coarse inverse always stays fp64: the coarse potential block is nearly singular (cond ~1e8), so
    // its inverse has entries that cancel to many digits and fp32 rounding would destroy the product
    KCHK(dev_upload_raw(ctx, &H.cinv, inv, (size_t)n * n));
    H.nc = n;
    return KNP_OK;
}

// exchange step of a distributed hierarchy: op 0 forward halo of a level-l vector, op 1 reverse halo (ghost rows added
// to their owners), op 2 SUM of the replicated coarse right-hand side built on level l.  Native plans when attached,
// the caller's hook otherwise.  Errors are latched in ctx->comm_rc (checked at the end of the solve).
static void level_exchange(knp_ctx* ctx, int hidx, int l, int op, double* vec) {
    KnpAmgLevel& L = ctx->hier[hidx].lv[l];
    int rc = KNP_OK;
    if (op == 2 && L.p2p_repl >= 0) rc = knp_p2p_allreduce(ctx, L.p2p_repl, vec, L.repl_n, nullptr, nullptr, 0);
    else if (op == 0 && L.p2p_halo >= 0) rc = knp_p2p_halo_forward(ctx, L.p2p_halo, vec);
    else if (op == 1 && L.p2p_halo >= 0) rc = knp_p2p_halo_reverse(ctx, L.p2p_halo, vec);
    else if (ctx->level_comm) rc = ctx->level_comm(ctx->comm_user, hidx, l, op, vec) != 0 ? KNP_E_STATE : KNP_OK;
    if (rc != KNP_OK && ctx->comm_rc == KNP_OK) {
        ctx->comm_rc = rc;
        if (ctx->err.empty() || rc == KNP_E_STATE) ctx->err = "level exchange failed (hierarchy " + std::to_string(hidx) + ", level " + std::to_string(l) + "): " + ctx->err;
    }
}
static inline bool level_comm_on(const knp_ctx* ctx) { return ctx->level_comm != nullptr || ctx->p2p != nullptr; }

// Chebyshev smoothing sweep of A x = b on one level with ping-pong buffers.  On entry the iterate is in
// *cur (ignored when zero_guess); the sweep alternates between bufA and bufB and leaves *cur pointing at
// the buffer that holds the result.
static void amg_smooth(knp_ctx* ctx, KnpAmgHier& H, int l, const double* b, double** cur, double* bufA, double* bufB, bool zero_guess,
                       bool first_done = false, bool ghosts_current = false) {
    KnpAmgLevel& L = H.lv[l];
    const double lmax = 1.1 * L.lambda_max, lmin = 0.1 * L.lambda_max;  // smoothing interval [0.1, 1.1] * lambda_max
    const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
    const double sigma = theta / delta;
    double rho_old = 1.0 / sigma;
    const int deg = H.cheby;
    hipStream_t st = ctx->stream;
    auto other = [&](double* p) { return p == bufA ? bufB : bufA; };
    const bool native0 = (l == 0) && H.native0 > 0;
    const int hidx = (int)(&H - ctx->hier);
    bool skip_halo = ghosts_current;   // only the first operator application of the sweep may rely on it
    auto step = [&](double* xin, double c1, double c2, double* out) {
        if (L.dist && level_comm_on(ctx) && !skip_halo) level_exchange(ctx, hidx, l, 0, xin);
        skip_halo = false;
        if (native0)
            launch_pnode<0>(st, H.native0 - 1, ctx->pc_group, ctx->g.n_nodes_owned,
                            L.dist ? ctx->g.n_nodes : ctx->g.n_nodes_owned, ctx->d_pair_ptr,
                            ctx->d_pair_col, ctx->d_p_vals, ctx->d_p_vals_f, L.inv_diag, b, xin, c1, c2, L.d, out);
        else
            launch_cheby(st, L.A_lanes, L.n, L.A_rp, L.A_ci, L.A_v, L.A_vf, L.inv_diag, b, xin, c1, c2, L.d, out);
    };
    if (zero_guess) {
        if (!first_done)   // otherwise the restriction kernel of the finer level already wrote d and *cur
            hipLaunchKernelGGL(k_cheby_first, dim3(std::min(nblocks(L.n), 2048)), dim3(NT), 0, st, L.n, 1.0 / theta, L.inv_diag, b, L.d, *cur);
    } else {
        double* out = other(*cur);
        step(*cur, 0.0, 1.0 / theta, out);
        *cur = out;
    }
    for (int k = 1; k < deg; ++k) {
        const double rho = 1.0 / (2.0 * sigma - rho_old);
        double* out = other(*cur);
        step(*cur, rho * rho_old, 2.0 * rho / delta, out);
        *cur = out;
        rho_old = rho;
    }
}

// number of buffer flips one level performs (decides where to start so that the result lands in `want`)
static int amg_flips(const KnpAmgHier& H, bool last_no_dense) {
    const int deg = H.cheby;
    if (last_no_dense) {
        const int sweeps = H.pre + H.post;
        return sweeps > 0 ? (deg - 1) + (sweeps - 1) * deg : 0;
    }
    int f = 0;
    if (H.pre > 0) f += (deg - 1) + (H.pre - 1) * deg;
    f += H.post * deg;
    return f;
}

// V-cycle on level l for right-hand side b.  The result is written to `want` when non-null (level 0: the
// caller's z), otherwise to whichever of the level's two buffers the ping-pong ends in; returns that pointer.
static double* amg_vcycle(knp_ctx* ctx, KnpAmgHier& H, int l, const double* b, double* want, bool first_done = false) {
    hipStream_t st = ctx->stream;
    KnpAmgLevel& L = H.lv[l];
    const bool last = (l == H.levels - 1);
    double* bufA = want ? want : L.x;
    double* bufB = L.r2;
    if (last && H.nc > 0) {
        if (H.cinv_f)
            launch_dense_matvec<float>(st, H.nc, H.cinv_f, b, bufA);
        else
            launch_dense_matvec<double>(st, H.nc, H.cinv, b, bufA);
        return bufA;
    }
    const int flips = amg_flips(H, last);
    double* cur = (flips & 1) ? bufB : bufA;    // start so that the final iterate lands in bufA
    if (last) {  // no coarse inverse supplied: smooth only
        bool zero = true;
        for (int sw = 0; sw < H.pre + H.post; ++sw) { amg_smooth(ctx, H, l, b, &cur, bufA, bufB, zero, zero && first_done); zero = false; }
        if (zero) { hipLaunchKernelGGL(k_fill, dim3(std::min(nblocks(L.n), 2048)), dim3(NT), 0, st, L.n, 0.0, bufA); cur = bufA; }
        return cur;
    }
    KnpAmgLevel& C = H.lv[l + 1];
    const int nc = L.n_coarse;
    const int hidx = (int)(&H - ctx->hier);
    // Level 0 in fused form inside the level-by-level cycle (distributed hierarchies): pre-smoothing + residual as one gather
    // with P Dinv (after the halo of the INPUT: x0 = c Dinv b is local, its ghost values follow from the ghost b), and below
    // prolongation + post-smoothing as one gather with S; the levels in between keep their exchanges.
    const bool f0 = (l == 0) && H.l0_fused;
    const bool fl = (l > 0) && L.lfused;     // the same form on a CSR level: r = b - (c A Dinv) b after the halo of b; S up-leg below
    bool zero = true;
    if (fl) {
        if (L.dist && level_comm_on(ctx)) level_exchange(ctx, hidx, l, 0, const_cast<double*>(b));
        if (L.At_vf) launch_spmv_t<1, 0, float>(st, L.A_lanes, L.n, L.A_rp, L.A_ci, L.At_vf, b, b, L.r);
        else launch_spmv_t<1, 0, double>(st, L.A_lanes, L.n, L.A_rp, L.A_ci, L.At_v, b, b, L.r);
    } else if (f0) {
        const double c0 = 1.0 / (0.5 * (1.1 + 0.1) * L.lambda_max);
        const int nn = ctx->g.n_nodes_owned;
        const int fm = H.native0 == 1 ? 0 : H.native0 == 2 ? 1 : 3;
        const int Gp = std::max(2, ctx->pc_group / 2);
        auto down = [&](int n_list, const int32_t* nodes) {
            if (fm == 3) {
                if (H.pt_phi_f) launch_l0_down<float>(st, 3, Gp, n_list, ctx->d_pair_ptr, ctx->d_pair_col, H.pt_phi_f, b, c0, L.r, nodes);
                else launch_l0_down<double>(st, 3, Gp, n_list, ctx->d_pair_ptr, ctx->d_pair_col, H.pt_phi, b, c0, L.r, nodes);
            } else {
                if (H.pt_f) launch_l0_down<float>(st, fm, Gp, n_list, ctx->d_pair_ptr, ctx->d_pair_col, H.pt_f, b, c0, L.r, nodes);
                else launch_l0_down<double>(st, fm, Gp, n_list, ctx->d_pair_ptr, ctx->d_pair_col, H.pt, b, c0, L.r, nodes);
            }
        };
        static const bool no_split = getenv("KNP_PC_SPLIT") && atoi(getenv("KNP_PC_SPLIT")) == 0;
        if (L.dist && L.p2p_halo >= 0 && ctx->n_bnd > 0 && ctx->n_int > 0 && ctx->stream3 && !no_split) {
            // The halo of the input runs on its own stream next to the nodes without a ghost neighbour; the others follow the join --
            // the same overlap the SpMV on A has (same interior / boundary lists: P's graph is a subgraph of A's)
            (void)hipEventRecord(ctx->ev_x, ctx->stream);
            (void)hipStreamWaitEvent(ctx->stream3, ctx->ev_x, 0);
            ctx->stream = ctx->stream3;
            level_exchange(ctx, hidx, 0, 0, const_cast<double*>(b));
            ctx->stream = st;
            (void)hipEventRecord(ctx->ev_halo, ctx->stream3);
            down(ctx->n_int, ctx->d_nodes_int);
            (void)hipStreamWaitEvent(st, ctx->ev_halo, 0);
            down(ctx->n_bnd, ctx->d_nodes_bnd);
        } else {
            if (L.dist && level_comm_on(ctx)) level_exchange(ctx, hidx, 0, 0, const_cast<double*>(b));
            down(nn, nullptr);
        }
    } else {
    for (int sw = 0; sw < H.pre; ++sw) { amg_smooth(ctx, H, l, b, &cur, bufA, bufB, zero, zero && first_done); zero = false; }
    if (zero) hipLaunchKernelGGL(k_fill, dim3(std::min(nblocks(L.n), 2048)), dim3(NT), 0, st, L.n, 0.0, cur);
    // r = b - A x ; b_c = R r
    if (L.dist && level_comm_on(ctx)) level_exchange(ctx, hidx, l, 0, cur);
    if (l == 0 && H.native0 > 0)
        launch_pnode<1>(st, H.native0 - 1, ctx->pc_group, ctx->g.n_nodes_owned,
                        L.dist ? ctx->g.n_nodes : ctx->g.n_nodes_owned, ctx->d_pair_ptr,
                        ctx->d_pair_col, ctx->d_p_vals, ctx->d_p_vals_f, L.inv_diag, b, cur, 0.0, 0.0, L.d, L.r);
    else
        launch_spmv_mp<1>(st, L.A_lanes, L.n, L.A_rp, L.A_ci, L.A_v, L.A_vf, cur, b, L.r);
    }
    // the coarse level starts with a Chebyshev step from a zero guess unless it is the dense solve or has no smoothing
    // before its own restriction; that step is pointwise in b_c and is fused into the restriction when b_c is complete
    // after this kernel (no reverse halo / all-reduce to follow)
    const bool c_last = (l + 1 == H.levels - 1);
    const bool c_smooths_first = c_last ? (H.nc == 0 && H.pre + H.post > 0) : (H.pre > 0);
    const bool fuse_first = c_smooths_first && L.repl_n == 0 && !C.dist && !(!c_last && C.lfused);
    if (fuse_first) {
        const int cflips = amg_flips(H, c_last);
        double* c_cur = (cflips & 1) ? C.r2 : C.x;          // where the coarse level's ping-pong starts (see below)
        const double c_theta = 0.5 * (1.1 + 0.1) * C.lambda_max;
        if (L.R_vf) launch_restrict_first_t<float>(st, L.R_lanes, nc, L.R_rp, L.R_ci, L.R_vf, L.r, C.b, 1.0 / c_theta, C.inv_diag, C.d, c_cur);
        else launch_restrict_first_t<double>(st, L.R_lanes, nc, L.R_rp, L.R_ci, L.R_v, L.r, C.b, 1.0 / c_theta, C.inv_diag, C.d, c_cur);
    } else {
        launch_spmv_mp<0>(st, L.R_lanes, nc, L.R_rp, L.R_ci, L.R_v, L.R_vf, L.r, nullptr, C.b);
    }
    if (level_comm_on(ctx)) {
        if (L.repl_n > 0) level_exchange(ctx, hidx, l, 2, C.b);           // replicate the coarse rhs
        else if (C.dist) level_exchange(ctx, hidx, l + 1, 1, C.b);        // ghost rows -> owners
    }
    double* xc = amg_vcycle(ctx, H, l + 1, C.b, nullptr, fuse_first);
    if (C.dist && level_comm_on(ctx) && L.repl_n == 0) level_exchange(ctx, hidx, l + 1, 0, xc);
    if (f0 || fl) {   // x = c Dinv b + c Dinv r + S x_c, straight into the caller's vector
        const double c0 = 1.0 / (0.5 * (1.1 + 0.1) * L.lambda_max);
        const int n_act = L.S_n_act > 0 ? L.S_n_act : L.S_rows;
        const int32_t* rp = L.S_n_act > 0 ? L.S_act_rp : L.S_rp;
        const int32_t* rows = L.S_n_act > 0 ? L.S_act_rows : nullptr;
        if (L.S_vf) launch_level_up_t<float, 0>(st, L.S_lanes, n_act, rows, rp, L.S_ci, L.S_vf, xc, L.inv_diag, b, L.r, nullptr, c0, c0, bufA, nullptr);
        else launch_level_up_t<double, 0>(st, L.S_lanes, n_act, rows, rp, L.S_ci, L.S_v, xc, L.inv_diag, b, L.r, nullptr, c0, c0, bufA, nullptr);
        return bufA;
    }
    // x += P x_c (fused)
    // with prolongator rows for the ghost entries (distributed levels) the ghosts of `cur` stay current: they held the
    // pre-smoothed iterate since the halo before the residual, and get the same correction as on their owner
    const int p_rows = L.P_rows > 0 ? L.P_rows : L.n;
    if (L.P_n_act > 0) {
        if (L.P_vf) launch_prolong_rows_t<float>(st, L.P_lanes, L.P_n_act, L.P_act_rows, L.P_act_rp, L.P_ci, L.P_vf, xc, cur);
        else launch_prolong_rows_t<double>(st, L.P_lanes, L.P_n_act, L.P_act_rows, L.P_act_rp, L.P_ci, L.P_v, xc, cur);
    } else {
        launch_spmv_mp<2>(st, L.P_lanes, p_rows, L.P_rp, L.P_ci, L.P_v, L.P_vf, xc, nullptr, cur);
    }
    const bool ghosts_current = L.dist && p_rows == L.n_loc && p_rows > L.n;
    for (int sw = 0; sw < H.post; ++sw) amg_smooth(ctx, H, l, b, &cur, bufA, bufB, false, false, sw == 0 && ghosts_current);
    return cur;
}

// ---- fused V(1,1) cycle (see k_l0_down / k_level_up) -------------------------------------------------------------------
// Eligible: level 0 on the library's own P (native0), V(1,1) with Chebyshev degree 1, S on every level above the last, no
// distributed level, dense solve on the last level.  Anything else takes amg_vcycle.
static bool fused_eligible(const knp_ctx* ctx, const KnpAmgHier& H) {
    const bool off = getenv("KNP_FUSED") && atoi(getenv("KNP_FUSED")) == 0;   // read at every knp_pc_setup
    if (off || !H.native0 || H.levels < 2 || H.cheby != 1 || H.pre != 1 || H.post != 1 || H.nc <= 0) return false;
    if (!(H.pt || H.pt_f || H.pt_phi || H.pt_phi_f || H.at0_rp)) return false;
    if (ctx->level_comm || ctx->p2p || ctx->halo) return false;
    for (int l = 0; l < H.levels - 1; ++l) {
        const KnpAmgLevel& L = H.lv[l];
        if (!L.S_rp || L.dist || L.repl_n > 0 || L.S_rows != L.n) return false;
    }
    return true;
}

// z (level-0 rows of this hierarchy) = V-cycle applied to b.  Potential-only hierarchies (native0 == 3) take b on compact
// node-indexed vectors and write z[4 node + 3] = cycle + cc[node] * b[node] (Schur term); the others work on 4 unknowns per node.
static void amg_cycle_fused(knp_ctx* ctx, KnpAmgHier& H, const double* b, double* z) {
    hipStream_t st = ctx->stream;
    const int nl = H.levels;
    const bool phi = H.native0 == 3;
    const int fm = H.native0 == 1 ? 0 : H.native0 == 2 ? 1 : 2;
    KnpAmgLevel& L0 = H.lv[0];
    auto cheb_c = [](const KnpAmgLevel& L) { return 1.0 / (0.5 * (1.1 + 0.1) * L.lambda_max); };   // 1/theta of the smoothing interval
    // level 0, down: r0 = b - c Pt b
    const double c0 = cheb_c(L0);
    const int nn = ctx->g.n_nodes_owned;
    if (phi && H.at0_rp) {   // level 0 = the uploaded (coupled) potential block: r = b - (c A Dinv) b on node-indexed vectors
        if (H.at0_vf) launch_spmv_t<1, 0, float>(st, H.at0_lanes, nn, H.at0_rp, H.at0_ci, H.at0_vf, b, b, L0.r);
        else launch_spmv_t<1, 0, double>(st, H.at0_lanes, nn, H.at0_rp, H.at0_ci, H.at0_v, b, b, L0.r);
    } else if (phi) {
        const int Gp = std::max(2, ctx->pc_group / 2);   // four trips per lane in flight: a quarter of the pairs per lane
        if (H.pt_phi_f) launch_l0_down<float>(st, 2, Gp, nn, ctx->d_pair_ptr, ctx->d_pair_col, H.pt_phi_f, b, c0, L0.r);
        else launch_l0_down<double>(st, 2, Gp, nn, ctx->d_pair_ptr, ctx->d_pair_col, H.pt_phi, b, c0, L0.r);
    } else {
        if (H.pt_f) launch_l0_down<float>(st, fm, std::max(2, ctx->pc_group / 2), nn, ctx->d_pair_ptr, ctx->d_pair_col, H.pt_f, b, c0, L0.r);
        else launch_l0_down<double>(st, fm, std::max(2, ctx->pc_group / 2), nn, ctx->d_pair_ptr, ctx->d_pair_col, H.pt, b, c0, L0.r);
    }
    // where level l keeps its right-hand side / where level l's iterate goes when the intermediate levels run as plain products:
    // b_l at the head of the level's `cat`, x_{l+1} behind it (the coarsest right-hand side and x_1 keep their own vectors)
    auto cf_b = [&](int l) { return (l >= 1 && l <= nl - 2) ? H.lv[l].cat : H.lv[l].b; };
    auto cf_x = [&](int l) { return l >= 2 ? H.lv[l - 1].cat + H.lv[l - 1].n : H.lv[l].x; };
    if (H.blocked && H.cfused) {
        const int nf = H.node_nf;
        launch_brestrict(st, nf, 4, L0.bR, L0.r, cf_b(1), 0.0, nullptr, nullptr, nullptr);
        for (int l = 1; l <= nl - 2; ++l) launch_brestrict(st, nf, nf, H.lv[l].bRt, cf_b(l), cf_b(l + 1), 0.0, nullptr, nullptr, nullptr);
        if (H.cinv_f) launch_dense_matvec<float>(st, H.nc, H.cinv_f, cf_b(nl - 1), cf_x(nl - 1));
        else launch_dense_matvec<double>(st, H.nc, H.cinv, cf_b(nl - 1), cf_x(nl - 1));
        for (int l = nl - 2; l >= 1; --l) launch_brestrict(st, nf, nf, H.lv[l].bU, H.lv[l].cat, cf_x(l), 0.0, nullptr, nullptr, nullptr);
        launch_blevel_up(st, nf, 4, L0.bS, cf_x(1), L0.inv_diag, b, L0.r, nullptr, c0, c0, z);
        return;
    }
    if (H.blocked) {   // node-blocked transfer and level operators (node-synchronised hierarchy): same cycle, one row per node
        const int nf = H.node_nf;
        for (int l = 0; l < nl - 1; ++l) {
            KnpAmgLevel& L = H.lv[l];
            KnpAmgLevel& C = H.lv[l + 1];
            if (l + 1 == nl - 1) {
                launch_brestrict(st, nf, l == 0 ? 4 : nf, L.bR, L.r, C.b, 0.0, nullptr, nullptr, nullptr);
                if (H.cinv_f) launch_dense_matvec<float>(st, H.nc, H.cinv_f, C.b, C.x);
                else launch_dense_matvec<double>(st, H.nc, H.cinv, C.b, C.x);
            } else {
                launch_brestrict(st, nf, l == 0 ? 4 : nf, L.bR, L.r, C.b, cheb_c(C), C.inv_diag, C.d, C.x);
                if (nf == 4) launch_bresidual_t<4>(st, C.bA, C.x, C.b, C.r);
                else launch_bresidual_t<3>(st, C.bA, C.x, C.b, C.r);
            }
        }
        for (int l = nl - 2; l >= 0; --l) {
            KnpAmgLevel& L = H.lv[l];
            const double c = cheb_c(L);
            if (l > 0) launch_blevel_up(st, nf, nf, L.bS, H.lv[l + 1].x, L.inv_diag, L.b, L.r, L.x, c, c, L.x);
            else launch_blevel_up(st, nf, 4, L.bS, H.lv[1].x, L.inv_diag, b, L.r, nullptr, c, c, z);
        }
        return;
    }
    auto plain = [&](int lanes, int n_rows, const int32_t* rp, const int32_t* ci, const double* v, const float* vf, const double* x, double* y) {
        if (vf) launch_spmv_t<0, 0, float>(st, lanes, n_rows, rp, ci, vf, x, nullptr, y);
        else launch_spmv_t<0, 0, double>(st, lanes, n_rows, rp, ci, v, x, nullptr, y);
    };
    if (H.cfused) {   // scalar rows, intermediate levels as plain products
        plain(L0.R_lanes, L0.n_coarse, L0.R_rp, phi ? L0.R_ci_c : L0.R_ci, L0.R_v, L0.R_vf, L0.r, cf_b(1));
        for (int l = 1; l <= nl - 2; ++l) plain(H.lv[l].Rt_lanes, H.lv[l].n_coarse, H.lv[l].Rt_rp, H.lv[l].Rt_ci, H.lv[l].Rt_v, H.lv[l].Rt_vf, cf_b(l), cf_b(l + 1));
        if (H.cinv_f) launch_dense_matvec<float>(st, H.nc, H.cinv_f, cf_b(nl - 1), cf_x(nl - 1));
        else launch_dense_matvec<double>(st, H.nc, H.cinv, cf_b(nl - 1), cf_x(nl - 1));
        for (int l = nl - 2; l >= 1; --l) plain(H.lv[l].U_lanes, H.lv[l].n, H.lv[l].U_rp, H.lv[l].U_ci, H.lv[l].U_v, H.lv[l].U_vf, H.lv[l].cat, cf_x(l));
    }
    // down the hierarchy
    for (int l = 0; l < nl - 1 && !H.cfused; ++l) {
        KnpAmgLevel& L = H.lv[l];
        KnpAmgLevel& C = H.lv[l + 1];
        const int nc = L.n_coarse;
        const int32_t* Rci = (l == 0 && phi) ? L.R_ci_c : L.R_ci;
        if (l + 1 == nl - 1) {   // coarsest: b_c = R r ; x_c = Cinv b_c
            if (L.R_vf) launch_spmv_t<0, 0, float>(st, L.R_lanes, nc, L.R_rp, Rci, L.R_vf, L.r, nullptr, C.b);
            else launch_spmv_t<0, 0, double>(st, L.R_lanes, nc, L.R_rp, Rci, L.R_v, L.r, nullptr, C.b);
            if (H.cinv_f) launch_dense_matvec<float>(st, H.nc, H.cinv_f, C.b, C.x);
            else launch_dense_matvec<double>(st, H.nc, H.cinv, C.b, C.x);
        } else {                 // b_c = R r ; x_c = c Dinv b_c (first Chebyshev step, fused) ; r_c = b_c - A_c x_c
            const double cc = cheb_c(C);
            if (L.R_vf) launch_restrict_first_t<float>(st, L.R_lanes, nc, L.R_rp, Rci, L.R_vf, L.r, C.b, cc, C.inv_diag, C.d, C.x);
            else launch_restrict_first_t<double>(st, L.R_lanes, nc, L.R_rp, Rci, L.R_v, L.r, C.b, cc, C.inv_diag, C.d, C.x);
            launch_spmv_mp<1>(st, C.A_lanes, C.n, C.A_rp, C.A_ci, C.A_v, C.A_vf, C.x, C.b, C.r);
        }
    }
    // up: x_l = x_l + c2 Dinv r_l + S x_{l+1}
    for (int l = H.cfused ? 0 : nl - 2; l >= 0; --l) {
        KnpAmgLevel& L = H.lv[l];
        KnpAmgLevel& C = H.lv[l + 1];
        const double c = cheb_c(L);
        const int n_act = L.S_n_act > 0 ? L.S_n_act : L.S_rows;
        const int32_t* rp = L.S_n_act > 0 ? L.S_act_rp : L.S_rp;
        if (l > 0) {
            if (L.S_vf) launch_level_up_t<float, 0>(st, L.S_lanes, n_act, L.S_n_act > 0 ? L.S_act_rows : nullptr, rp, L.S_ci, L.S_vf, C.x, L.inv_diag, L.b, L.r, L.x, c, c, L.x, nullptr);
            else launch_level_up_t<double, 0>(st, L.S_lanes, n_act, L.S_n_act > 0 ? L.S_act_rows : nullptr, rp, L.S_ci, L.S_v, C.x, L.inv_diag, L.b, L.r, L.x, c, c, L.x, nullptr);
        } else if (phi) {
            const int32_t* rows = L.S_n_act > 0 ? L.S_act_rows_c : nullptr;
            const double* ccp = ctx->pc_kind == KNP_PC_AMG_LT ? nullptr : ctx->d_cc;   // no Schur term in the literal lower-triangular form
            if (L.S_vf) launch_level_up_t<float, 1>(st, L.S_lanes, L.S_n_act > 0 ? n_act : nn, rows, rp, L.S_ci, L.S_vf, C.x, L.dinv_c, b, L.r, nullptr, c, c, z, ccp);
            else launch_level_up_t<double, 1>(st, L.S_lanes, L.S_n_act > 0 ? n_act : nn, rows, rp, L.S_ci, L.S_v, C.x, L.dinv_c, b, L.r, nullptr, c, c, z, ccp);
        } else {
            if (L.S_vf) launch_level_up_t<float, 0>(st, L.S_lanes, n_act, L.S_n_act > 0 ? L.S_act_rows : nullptr, rp, L.S_ci, L.S_vf, C.x, L.inv_diag, b, L.r, nullptr, c, c, z, nullptr);
            else launch_level_up_t<double, 0>(st, L.S_lanes, n_act, L.S_n_act > 0 ? L.S_act_rows : nullptr, rp, L.S_ci, L.S_v, C.x, L.inv_diag, b, L.r, nullptr, c, c, z, nullptr);
        }
    }
}

static int check_hier(knp_ctx* ctx, int h) {
    KnpAmgHier& H = ctx->hier[h];
    if (H.levels < 1 || H.lv[0].n != ctx->n_dof_owned) { ctx->err = "AMG hierarchy " + std::to_string(h) + " not supplied"; return KNP_E_STATE; }
    if (H.native0 == 0 && H.lv[0].A_nnz == 0 && H.levels > 1) {   // uploaded with an empty level-0 pattern (the library's own P was to serve as level 0)
        ctx->err = "AMG hierarchy " + std::to_string(h) + ": level 0 has no operator (uploaded for knp_amg_use_native_level0, which is off)";
        return KNP_E_STATE;
    }
    for (int l = 0; l < H.levels - 1; ++l) {
        const int expect = H.lv[l].repl_n > 0 ? H.lv[l + 1].n : H.lv[l + 1].n_loc;   // rows of R = coarse local size
        if (H.lv[l].n_coarse != expect) { ctx->err = "AMG level sizes inconsistent"; return KNP_E_STATE; }
        if (H.lv[l].repl_n > 0 && H.lv[l].repl_n != H.lv[l + 1].n) { ctx->err = "replicated coarse size mismatch"; return KNP_E_STATE; }
    }
    if (H.nc > 0 && H.nc != H.lv[H.levels - 1].n) { ctx->err = "AMG coarse inverse size mismatch"; return KNP_E_STATE; }
    return KNP_OK;
}

// At = c A Dinv of level l (>= 1): the columns of ghost unknowns are scaled with the ghost inverse diagonal, fetched with one
// forward halo of the level (collective: every rank builds the same levels in the same order inside knp_pc_setup)
static int build_level_scaled(knp_ctx* ctx, int hidx, int l) {
    KnpAmgLevel& L = ctx->hier[hidx].lv[l];
    const double c = 1.0 / (0.5 * (1.1 + 0.1) * L.lambda_max);
    double* dloc = nullptr;
    HIPCHK(hipMalloc((void**)&dloc, (size_t)std::max(L.n_loc, 1) * sizeof(double)));
    HIPCHK(hipMemsetAsync(dloc, 0, (size_t)std::max(L.n_loc, 1) * sizeof(double), ctx->stream));
    HIPCHK(hipMemcpyAsync(dloc, L.inv_diag, (size_t)L.n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    if (L.dist && level_comm_on(ctx)) level_exchange(ctx, hidx, l, 0, dloc);
    const int nblk = (int)std::min<int64_t>(nblocks(L.A_nnz), 8192);
    if (L.A_vf) {
        HIPCHK(hipMalloc((void**)&L.At_vf, (size_t)L.A_nnz * sizeof(float)));
        hipLaunchKernelGGL((k_build_at<float, float>), dim3(nblk), dim3(NT), 0, ctx->stream, L.A_nnz, L.A_ci, L.A_vf, dloc, c, L.At_vf);
    } else {
        HIPCHK(hipMalloc((void**)&L.At_v, (size_t)L.A_nnz * sizeof(double)));
        hipLaunchKernelGGL((k_build_at<double, double>), dim3(nblk), dim3(NT), 0, ctx->stream, L.A_nnz, L.A_ci, L.A_v, dloc, c, L.At_v);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    (void)hipFree(dloc);
    if (ctx->comm_rc != KNP_OK) { const int rc = ctx->comm_rc; ctx->comm_rc = KNP_OK; return rc; }
    return KNP_OK;
}

int knp_pc_setup(knp_ctx* ctx, int32_t kind) {
    CHECK_CTX(ctx);
    side_discard(ctx);
    KCHK(join_asm(ctx));
    if (kind != KNP_PC_NONE && kind != KNP_PC_VBJACOBI && kind != KNP_PC_AMG && kind != KNP_PC_AMG_BT && kind != KNP_PC_AMG_LT) { ctx->err = "unknown pc kind"; return KNP_E_ARG; }
    if (kind == KNP_PC_AMG) KCHK(check_hier(ctx, 0));
    if (kind == KNP_PC_AMG_BT || kind == KNP_PC_AMG_LT) {
        KCHK(check_hier(ctx, 0));
        KCHK(check_hier(ctx, 1));
        if (!ctx->d_t2) {
            HIPCHK(hipMalloc((void**)&ctx->d_t2, std::max(ctx->n_dof_local, 1) * sizeof(double)));
            HIPCHK(hipMalloc((void**)&ctx->d_w2, std::max(ctx->n_dof_local, 1) * sizeof(double)));
            HIPCHK(hipMemset(ctx->d_t2, 0, std::max(ctx->n_dof_local, 1) * sizeof(double)));
            HIPCHK(hipMemset(ctx->d_w2, 0, std::max(ctx->n_dof_local, 1) * sizeof(double)));
        }
    }
    ctx->pc_kind = kind;
    for (int h = 0; h < KNP_MAX_HIER; ++h) ctx->hier[h].fused = 0;
    if (kind == KNP_PC_AMG) ctx->hier[0].fused = fused_eligible(ctx, ctx->hier[0]) ? 1 : 0;
    if (kind == KNP_PC_AMG_BT || kind == KNP_PC_AMG_LT) {   // both or none: the potential hierarchy then works on compact vectors
        const bool ok = fused_eligible(ctx, ctx->hier[0]) && fused_eligible(ctx, ctx->hier[1]) && ctx->hier[1].native0 == 3 && ctx->hier[0].native0 == 2 &&
                        ctx->hier[1].lv[0].S_n_act > 0 && ctx->hier[1].lv[0].S_act_rows_c && ctx->hier[1].lv[0].R_ci_c;
        ctx->hier[0].fused = ctx->hier[1].fused = ok ? 1 : 0;
        // a potential hierarchy on its uploaded (coupled) level-0 operator outside the fused cycle: the level-by-level cycle must
        // not smooth with the library's own uncoupled P on level 0 -- it takes the uploaded CSR like any other level
        if (!ok && ctx->hier[1].l0_upload) ctx->hier[1].native0 = 0;
    }
    for (int h = 0; h < KNP_MAX_HIER; ++h) {   // node-blocked operators under the fused cycle: every level must have them
        KnpAmgHier& H = ctx->hier[h];
        const bool off = getenv("KNP_BLOCKED") && atoi(getenv("KNP_BLOCKED")) == 0;
        bool ok = !off && H.fused && H.node_nf > 0 && (H.native0 == 1 || H.native0 == 2) && (H.native0 == 1) == (H.node_nf == 4);
        for (int l = 0; ok && l < H.levels - 1; ++l) {
            const KnpAmgLevel& L = H.lv[l];
            ok = L.bR.rp && L.bS.rp && (l == 0 || L.bA.rp) && L.bR.n_rows * H.node_nf == L.n_coarse &&
                 L.bS.n_rows * (l == 0 ? 4 : H.node_nf) == L.n;
        }
        H.blocked = ok ? 1 : 0;
        // intermediate levels as two plain products each: every one of them must carry Rt and U (node-blocked when the cycle is)
        const bool coff = getenv("KNP_COARSE_FUSED") && atoi(getenv("KNP_COARSE_FUSED")) == 0;
        bool cok = !coff && H.fused && H.levels >= 3;
        for (int l = 1; cok && l < H.levels - 1; ++l) {
            const KnpAmgLevel& L = H.lv[l];
            cok = L.Rt_rp && L.U_rp && (!H.blocked || (L.bRt.rp && L.bU.rp && L.bRt.n_rows * H.node_nf == L.n_coarse && L.bU.n_rows * H.node_nf == L.n));
        }
        H.cfused = cok ? 1 : 0;
        for (int l = 1; cok && l < H.levels - 1; ++l) {
            KnpAmgLevel& L = H.lv[l];
            if (L.cat) continue;
            const size_t nc = (size_t)L.n + (size_t)L.n_coarse;
            HIPCHK(hipMalloc((void**)&L.cat, nc * sizeof(double)));
            HIPCHK(hipMemsetAsync(L.cat, 0, nc * sizeof(double), ctx->stream));
            ctx->side_ws = false;
        }
    }
    for (int h = 0; h < KNP_MAX_HIER; ++h) {   // level 0 in fused form inside the level-by-level cycle (distributed hierarchies)
        KnpAmgHier& H = ctx->hier[h];
        const bool off = getenv("KNP_FUSED") && atoi(getenv("KNP_FUSED")) == 0;
        const KnpAmgLevel& L0 = H.lv[0];
        const bool have_pt = H.native0 == 3 ? (H.pt_phi || H.pt_phi_f) : (H.pt || H.pt_f);
        H.l0_fused = (!off && !H.fused && H.native0 > 0 && H.levels >= 2 && H.cheby == 1 && H.pre == 1 && H.post == 1 && L0.S_rp &&
                      L0.S_rows == L0.n && have_pt) ? 1 : 0;
        // the levels below it likewise (distributed levels and the replicated tail): At = c A Dinv for the down-leg, S for the up-leg
        const bool used = (kind == KNP_PC_AMG && h == 0) || ((kind == KNP_PC_AMG_BT || kind == KNP_PC_AMG_LT) && h < 2);
        // Eligibility is decided GLOBALLY: build_level_scaled and the fused leg exchange a different number of halos than the
        // unfused one, so a rank that owns nothing on a distributed level must still take the decision of its peers.
        // Every rank contributes "1 = my part of this level cannot run fused"; the level is fused where the sum is zero.
        double veto[KNP_MAX_AMG_LEVELS] = {0};
        const bool lvl_off = getenv("KNP_FUSED_LEVELS") && atoi(getenv("KNP_FUSED_LEVELS")) == 0;
        for (int l = 1; l < H.levels - 1; ++l) {
            KnpAmgLevel& L = H.lv[l];
            dev_free(L.At_v); dev_free(L.At_vf);
            L.lfused = 0;
            const bool empty_here = L.dist && L.n == 0;      // owns no rows of a distributed level: follows its peers
            const bool local_ok = !(!L.S_rp || L.S_rows != L.n || L.S_n_act > 0 || L.n_coarse <= 0 || L.A_nnz <= 0 || !L.inv_diag);
            veto[l] = (!used || off || lvl_off || H.fused || H.cheby != 1 || H.pre != 1 || H.post != 1 || !(local_ok || empty_here)) ? 1.0 : 0.0;
        }
        if (used && H.levels > 2) KCHK(global_sum_small(ctx, veto + 1, H.levels - 2));
        for (int l = 1; l < H.levels - 1; ++l) {
            if (veto[l] != 0.0) continue;
            KCHK(build_level_scaled(ctx, h, l));
            H.lv[l].lfused = 1;
        }
    }
    if (kind == KNP_PC_VBJACOBI && ctx->have_A) {
        const KnpHostGraph& g = ctx->g;
        hipLaunchKernelGGL(k_vbj_extract, dim3(nblocks(g.n_nodes_owned)), dim3(NT), 0, ctx->stream, g.n_nodes_owned,
                           ctx->d_pair_ptr, ctx->d_pair_col, ctx->d_ac, ctx->d_at, ctx->d_ax, ctx->d_node_gv, ctx->d_node_side,
                           ctx->d_gdiag, ctx->d_vbj);
        HIPCHK(hipGetLastError());
    }
    return KNP_OK;
}

// z = M^{-1} r, then (optionally) gauge projection.  cnt = global number of potential DoFs.
// Dirichlet rows of A are identity rows: the preconditioner must act as the identity there
__global__ void __launch_bounds__(NT) k_bc_copy(int n_bc, const int32_t* __restrict__ bc_dofs, const double* __restrict__ src,
                                                double* __restrict__ dst, int phi_only) {
    const int i = blockIdx.x * NT + threadIdx.x;
    if (i >= n_bc) return;
    const int d = bc_dofs[i];
    if (phi_only == 2) {   // destination is a node-indexed potential vector
        if ((d & 3) == 3) dst[d >> 2] = src[d];
    } else if (!phi_only || (d & 3) == 3) {
        dst[d] = src[d];
    }
}

static int pc_apply_proj(knp_ctx* ctx, const double* r, double* z, int64_t cnt) {
    const int dm = ctx->defl_m;
    if (dm > 0) {   // coarse sums of the input residual (before r is possibly overwritten)
        ProfScope ps(ctx, 4);
        const int no = ctx->g.n_nodes_owned;
        const int nb = std::min(RED_BLOCKS, nblocks(no));
        for (int c0 = 0; c0 < dm; c0 += 8)
            hipLaunchKernelGGL((k_defl_sums<8>), dim3(nb), dim3(NT), 0, ctx->stream, no, c0, dm, ctx->d_defl_mode, r, ctx->d_partial);
        hipLaunchKernelGGL(k_reduce_partials, dim3(dm), dim3(NT), 0, ctx->stream, nb, ctx->d_partial, ctx->d_red, DEFL_SLOT0, (double*)nullptr);
        KCHK(allreduce_slots(ctx, DEFL_SLOT0, dm));
    }
    {
        ProfScope ps(ctx, 2);
        const KnpHostGraph& g = ctx->g;
        switch (ctx->pc_kind) {
            case KNP_PC_VBJACOBI:
                if (!ctx->have_A) { ctx->err = "vertex-block Jacobi needs an assembled matrix"; return KNP_E_STATE; }
                hipLaunchKernelGGL(k_vbj_apply, dim3(nblocks(g.n_nodes_owned)), dim3(NT), 0, ctx->stream, g.n_nodes_owned,
                                   ctx->d_node_side, ctx->d_node_gv, ctx->d_gv_node_e, ctx->d_vbj, r, z);
                break;
            case KNP_PC_AMG: {
                if (ctx->hier[0].fused) { amg_cycle_fused(ctx, ctx->hier[0], r, z); break; }
                double* out = amg_vcycle(ctx, ctx->hier[0], 0, r, z);
                if (out != z) HIPCHK(hipMemcpyAsync(z, out, (size_t)ctx->n_dof_owned * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                break;
            }
            case KNP_PC_AMG_BT:
            case KNP_PC_AMG_LT: {
                // z_k = V_k r_k ; t_phi = r_phi - A_{phi k} z_k ; z_phi = V_phi t_phi + cc * t_phi   (LT: literal coupling, no cc term)
                const bool lit = ctx->pc_kind == KNP_PC_AMG_LT;
                if (!ctx->have_A || !ctx->have_cc) { ctx->err = "block-triangular preconditioner needs an assembled matrix"; return KNP_E_STATE; }
                if (ctx->hier[0].fused) {
                    amg_cycle_fused(ctx, ctx->hier[0], r, z);                 // ion entries of z
                    if (lit) launch_phi_rhs_literal<true>(ctx, r, z, ctx->d_t2);
                    else launch_phi_rhs<true>(ctx, r, z, ctx->d_t2);          // t_phi on node-indexed vectors
                    if (ctx->n_bc > 0)
                        hipLaunchKernelGGL(k_bc_copy, dim3(nblocks(ctx->n_bc)), dim3(NT), 0, ctx->stream, ctx->n_bc, ctx->d_bc_dofs, r, ctx->d_t2, 2);
                    amg_cycle_fused(ctx, ctx->hier[1], ctx->d_t2, z);         // z_phi = V_phi t + cc t
                    break;
                }
                double* out = amg_vcycle(ctx, ctx->hier[0], 0, r, z);
                if (out != z) HIPCHK(hipMemcpyAsync(z, out, (size_t)ctx->n_dof_owned * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                KCHK(halo_update(ctx, z));   // the mass-matrix row may reach ghost ion unknowns
                if (lit) launch_phi_rhs_literal<false>(ctx, r, z, ctx->d_t2);
                else launch_phi_rhs<false>(ctx, r, z, ctx->d_t2);
                if (ctx->n_bc > 0)   // pinned potentials: no Schur coupling, their right-hand side is the residual itself
                    hipLaunchKernelGGL(k_bc_copy, dim3(nblocks(ctx->n_bc)), dim3(NT), 0, ctx->stream, ctx->n_bc, ctx->d_bc_dofs, r, ctx->d_t2, 1);
                double* w = amg_vcycle(ctx, ctx->hier[1], 0, ctx->d_t2, ctx->d_w2);
                hipLaunchKernelGGL(k_schur_fin, dim3(nblocks(g.n_nodes_owned)), dim3(NT), 0, ctx->stream, g.n_nodes_owned, lit ? (const double*)nullptr : ctx->d_cc, ctx->d_t2, w, z);
                break;
            }
            default:
                HIPCHK(hipMemcpyAsync(z, r, (size_t)ctx->n_dof_owned * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                break;
        }
        if (ctx->n_bc > 0 && (ctx->pc_kind == KNP_PC_AMG || ctx->pc_kind == KNP_PC_AMG_BT || ctx->pc_kind == KNP_PC_AMG_LT))
            hipLaunchKernelGGL(k_bc_copy, dim3(nblocks(ctx->n_bc)), dim3(NT), 0, ctx->stream, ctx->n_bc, ctx->d_bc_dofs, r, z, 0);
        HIPCHK(hipGetLastError());
    }
    if (dm > 0) {
        ProfScope ps(ctx, 4);
        const int no = ctx->g.n_nodes_owned;
        hipLaunchKernelGGL(k_defl_add, dim3(nblocks(no)), dim3(NT), 0, ctx->stream, no, dm, ctx->d_defl_mode, ctx->d_defl_einv,
                           ctx->d_red + DEFL_SLOT0, z);
    }
    if (ctx->ns_on && cnt > 0) {
        ProfScope ps(ctx, 4);
        const int no = ctx->g.n_nodes_owned;
        const int nb = std::min(RED_BLOCKS, nblocks(no));
        hipLaunchKernelGGL(k_phi_sum, dim3(nb), dim3(NT), 0, ctx->stream, no, z, ctx->d_partial);
        hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(NT), 0, ctx->stream, nb, ctx->d_partial, ctx->d_red, 62, (double*)nullptr);
        KCHK(allreduce_slots(ctx, 62, 1));
        hipLaunchKernelGGL(k_phi_sub, dim3(nblocks(no)), dim3(NT), 0, ctx->stream, no, ctx->d_red + 62, 1.0 / (double)cnt, z);
        HIPCHK(hipGetLastError());
    }
    return KNP_OK;
}

int knp_set_deflation(knp_ctx* ctx, int32_t n_modes, const int32_t* node_mode, const double* einv) {
    CHECK_CTX(ctx);
    side_discard(ctx);
    if (n_modes < 0 || n_modes > DEFL_MAX || (n_modes > 0 && (!node_mode || !einv))) { ctx->err = "bad deflation arguments (at most 32 modes)"; return KNP_E_ARG; }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->defl_m = 0;
    if (n_modes == 0) return KNP_OK;
    const int no = ctx->g.n_nodes_owned;
    for (int n = 0; n < no; ++n)
        if (node_mode[n] < -1 || node_mode[n] >= n_modes) { ctx->err = "deflation mode id out of range"; return KNP_E_ARG; }
    dev_free(ctx->d_defl_mode);
    KCHK(dev_upload_raw(ctx, &ctx->d_defl_mode, node_mode, (size_t)no));
    HIPCHK(hipMemcpy(ctx->d_defl_einv, einv, (size_t)n_modes * n_modes * sizeof(double), hipMemcpyHostToDevice));
    ctx->defl_m = n_modes;
    return KNP_OK;
}

int knp_pc_apply(knp_ctx* ctx, const double* r, double* z) {
    CHECK_CTX(ctx);
    side_discard(ctx);
    if (!r || !z) return KNP_E_ARG;
    int rc;
    const int64_t cnt = ctx->ns_on ? global_phi_count(ctx, &rc) : 0;
    return pc_apply_proj(ctx, r, z, cnt);
}

// ---- GMRES(restart), left preconditioning, classical Gram-Schmidt (KSPGMRES semantics) ------
// ---- ||B b|| of the next solve on a side stream ------------------------------------------------------------------------
// The first preconditioner application of a solve only needs the right-hand side, not the matrix: started right after the
// right-hand side is assembled it overlaps the matrix assembly of the same timestep (separate HIP streams; the V-cycle's
// coarse levels are launch-latency bound, the assembly is bandwidth bound).
static void side_discard(knp_ctx* ctx) {   // any call that could touch what the side stream uses joins it first
    if (ctx->prep_b) {
        if (!ctx->prep_deferred) (void)hipEventSynchronize(ctx->ev_join);   // deferred: nothing was enqueued yet
        ctx->prep_b = nullptr;
        ctx->prep_deferred = false;
    }
}

static bool exchanges_all_native(const knp_ctx* ctx) {
    if (!ctx->p2p || ctx->p2p_fine < 0 || ctx->p2p_red < 0 || ctx->defl_m > 0) return false;
    const int nh = ctx->pc_kind == KNP_PC_AMG ? 1 : (ctx->pc_kind == KNP_PC_AMG_BT || ctx->pc_kind == KNP_PC_AMG_LT) ? 2 : 0;
    for (int h = 0; h < nh; ++h)
        for (int l = 0; l < ctx->hier[h].levels; ++l) {
            const KnpAmgLevel& L = ctx->hier[h].lv[l];
            if (L.dist && L.p2p_halo < 0) return false;
            if (L.repl_n > 0 && L.p2p_repl < 0) return false;
        }
    return true;
}

// z = B r and the squared norm of its gauge-projected part with ONE reduction: {sum of the potential entries, z.z} are reduced
// together and |z - ns (ns.z)|^2 = z.z - s^2/cnt (k_proj_norm; slot 60 = the norm, 61 = its cancellation flag, 62 = s).  z itself is
// left UNPROJECTED (*fused = true): the caller subtracts the mean when it normalises (k_scale_rsqrt_proj) or does not need z at
// all (||B b||).  Without a null space: the plain sequence, *fused = false.
// contexts without an exchange between the partial sums and their use: the reduction finishes in one single-block kernel
static bool fin_ok(const knp_ctx* ctx) {
    static const bool off = getenv("KNP_FIN") && atoi(getenv("KNP_FIN")) == 0;
    return !off && !ctx->allreduce && ctx->p2p_red < 0;
}
static bool fused_norm_possible(const knp_ctx* ctx, int64_t cnt) {
    static const bool off = getenv("KNP_NO_FUSED_NORM") != nullptr;
    return ctx->ns_on && cnt > 0 && !off && ctx->defl_m == 0;
}
// side = true: the side-stream form of knp_gmres_prepare's concurrent mode -- partial sums in d_partial_s, reduced values in slots
// 122/123 -> {norm, flag} in 120/121 (+ pinned mirror), and NO sequence word (the solve joins it with an event): nothing the main
// stream uses is touched.  Requires fused_norm_possible.
static int pc_apply_norm(knp_ctx* ctx, const double* r, double* z, int64_t cnt, bool* fused, bool side = false) {
    if (side) {
        *fused = true;
        KCHK(pc_apply_proj(ctx, r, z, 0));
        ProfScope ps(ctx, 1);
        const int nb = ctx->n_red_blocks;
        hipLaunchKernelGGL((k_multi_dot<8, true, true>), dim3(nb), dim3(NT), 0, ctx->stream, ctx->n_dof_owned, (int64_t)ctx->n_dof_local, 0, 0,
                           ctx->d_V, z, ctx->d_partial_s);
        ++ctx->n_allreduce;
        hipLaunchKernelGGL(k_reduce_fin, dim3(1), dim3(NT), 0, ctx->stream, 2, nb, 2, ctx->d_partial_s, ctx->d_red, SIDE_SLOT + 2, GmLayout{1}, 0, 0,
                           1.0 / (double)cnt, (double*)nullptr, SIDE_SLOT, GM_CANCEL, ctx->mirror(), (volatile int64_t*)nullptr, (int64_t)0);
        HIPCHK(hipGetLastError());
        return KNP_OK;
    }
    if (!fused_norm_possible(ctx, cnt)) {
        *fused = false;
        KCHK(pc_apply_proj(ctx, r, z, cnt));
        return dot_to_slot(ctx, z, z, 60);
    }
    *fused = true;
    KCHK(pc_apply_proj(ctx, r, z, 0));
    ProfScope ps(ctx, 1);
    const int nb = ctx->n_red_blocks;
    hipLaunchKernelGGL((k_multi_dot<8, true, true>), dim3(nb), dim3(NT), 0, ctx->stream, ctx->n_dof_owned, (int64_t)ctx->n_dof_local, 0, 0,
                       ctx->d_V, z, ctx->d_partial);
    if (fin_ok(ctx)) {
        ++ctx->n_allreduce;
        hipLaunchKernelGGL(k_reduce_fin, dim3(1), dim3(NT), 0, ctx->stream, 2, nb, 2, ctx->d_partial, ctx->d_red, 62, GmLayout{1}, 0, 0, 1.0 / (double)cnt,
                           (double*)nullptr, 60, GM_CANCEL, ctx->mirror(), ctx->mirror() ? ctx->h_seq_dev : nullptr, ++ctx->seq_counter);
        HIPCHK(hipGetLastError());
        return KNP_OK;
    }
    hipLaunchKernelGGL(k_reduce_partials, dim3(2), dim3(NT), 0, ctx->stream, nb, ctx->d_partial, ctx->d_red, 62, (double*)nullptr);
    KCHK(allreduce_slots(ctx, 62, 2));
    hipLaunchKernelGGL(k_proj_norm, dim3(1), dim3(64), 0, ctx->stream, ctx->d_red, 62, 60, 1.0 / (double)cnt, GM_CANCEL, ctx->mirror(),
                       ctx->mirror() ? ctx->h_seq_dev : nullptr, ++ctx->seq_counter);
    HIPCHK(hipGetLastError());
    return KNP_OK;
}
// the host side of it: read {norm^2, flag}; on cancellation project z explicitly and reduce again (*fused becomes false)
static int pc_norm_read(knp_ctx* ctx, double* z, int64_t cnt, bool* fused) {
    KCHK(read_slots(ctx, 60, *fused ? 2 : 1, ctx->seq_counter));
    if (*fused && ctx->h_red[61] != 0.0) {
        ++ctx->n_norm_fallback;
        const int no = ctx->g.n_nodes_owned;
        hipLaunchKernelGGL(k_phi_sub, dim3(nblocks(no)), dim3(NT), 0, ctx->stream, no, ctx->d_red + 62, 1.0 / (double)cnt, z);
        *fused = false;
        KCHK(dot_to_slot(ctx, z, z, 60));
        KCHK(read_slots(ctx, 60, 1, ctx->seq_counter));
    }
    return KNP_OK;
}

// second set of work vectors for the cycle that runs on the side stream while the main stream applies the same hierarchies
static int ensure_side_ws(knp_ctx* ctx) {
    if (ctx->side_ws) return KNP_OK;
    auto zalloc = [&](double** p, size_t n) -> int {
        HIPCHK(hipMalloc((void**)p, std::max<size_t>(n, 1) * sizeof(double)));
        HIPCHK(hipMemsetAsync(*p, 0, std::max<size_t>(n, 1) * sizeof(double), ctx->stream));
        return KNP_OK;
    };
    for (int h = 0; h < KNP_MAX_HIER; ++h)
        for (int l = 0; l < ctx->hier[h].levels; ++l) {
            KnpAmgLevel& L = ctx->hier[h].lv[l];
            const size_t n = (size_t)std::max(L.n_loc, L.n);
            if (n == 0) continue;
            if (!L.xs) { KCHK(zalloc(&L.xs, n)); KCHK(zalloc(&L.bs, n)); KCHK(zalloc(&L.rs, n)); KCHK(zalloc(&L.ds, n)); KCHK(zalloc(&L.r2s, n)); }
            if (L.cat && !L.cats) KCHK(zalloc(&L.cats, (size_t)L.n + (size_t)L.n_coarse));
        }
    const size_t nl = (size_t)std::max(ctx->n_dof_local, 1);
    if (!ctx->d_t2_s) { KCHK(zalloc(&ctx->d_t2_s, nl)); KCHK(zalloc(&ctx->d_w2_s, nl)); }
    if (!ctx->d_wb) KCHK(zalloc(&ctx->d_wb, nl));
    if (!ctx->d_partial_s) KCHK(zalloc(&ctx->d_partial_s, (size_t)RED_SLOTS * RED_BLOCKS));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->side_ws = true;
    return KNP_OK;
}
static void swap_side_ws(knp_ctx* ctx) {   // host-side pointer swap: kernel arguments are taken at launch
    for (int h = 0; h < KNP_MAX_HIER; ++h)
        for (int l = 0; l < ctx->hier[h].levels; ++l) {
            KnpAmgLevel& L = ctx->hier[h].lv[l];
            if (!L.xs) continue;
            std::swap(L.x, L.xs); std::swap(L.b, L.bs); std::swap(L.r, L.rs); std::swap(L.d, L.ds); std::swap(L.r2, L.r2s);
            if (L.cat && L.cats) std::swap(L.cat, L.cats);
        }
    if (ctx->d_t2 && ctx->d_t2_s) { std::swap(ctx->d_t2, ctx->d_t2_s); std::swap(ctx->d_w2, ctx->d_w2_s); }
}
// the side-stream cycle may run next to the solve's own first preconditioner application: one GPU, fused cycles (their work
// vectors are exactly the per-level sets swapped above), gauge-projected norm from one reduction, pinned mirror for the result
static bool side_concurrent_ok(const knp_ctx* ctx, int64_t cnt) {
    static const bool off = getenv("KNP_SIDE_CONCURRENT") && atoi(getenv("KNP_SIDE_CONCURRENT")) == 0;
    if (off || ctx->halo || ctx->allreduce || ctx->level_comm || ctx->p2p || ctx->n_bc > 0 || !ctx->h_red_dev) return false;
    if (!fused_norm_possible(ctx, cnt)) return false;
    if (ctx->pc_kind == KNP_PC_AMG) return ctx->hier[0].fused != 0;
    if (ctx->pc_kind == KNP_PC_AMG_BT || ctx->pc_kind == KNP_PC_AMG_LT) return ctx->hier[0].fused && ctx->hier[1].fused;
    return false;
}

int knp_gmres_prepare(knp_ctx* ctx, const double* b) {
    CHECK_CTX(ctx);
    if (!b) return KNP_E_ARG;
    side_discard(ctx);
    // single-GPU contexts only (the exchanges of a distributed preconditioner are ordered on the main stream), and only
    // once the Krylov workspace exists (second solve onwards)
    const bool off = getenv("KNP_NO_PREPARE") != nullptr;
    // vertex-block Jacobi takes its blocks from the matrix that knp_assemble_matrix is about to rewrite: nothing to overlap
    if (ctx->pc_kind == KNP_PC_VBJACOBI) return KNP_OK;
    // Distributed contexts: legal when EVERY exchange of the preconditioner runs in the library (native peer-to-peer plans):
    // those kernels are launched on ctx->stream, i.e. on the side stream here, in the same order on every rank, and the main
    // stream does no exchange until the solve joins.  With torch.distributed hooks (ordered on torch's stream) it stays a no-op.
    if (off || ctx->gm_restart <= 0 || (ctx->prof_on & ~1)) return KNP_OK;
    if ((ctx->halo || ctx->allreduce || ctx->level_comm || ctx->p2p) && !exchanges_all_native(ctx)) return KNP_OK;
    if (!ctx->stream2) {
        HIPCHK(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    }
    int rc = KNP_OK;
    const int64_t cnt = ctx->ns_on ? global_phi_count(ctx, &rc) : 0;
    KCHK(rc);
    const bool conc = side_concurrent_ok(ctx, cnt);
    HIPCHK(hipEventRecord(ctx->ev_fork, ctx->stream));
    if (conc) {
        // Concurrent form: only the fork point is fixed here (b is final).  The cycle itself is enqueued by knp_gmres_solve BEHIND the
        // launches of its own first residual chain, so that the host feeds the critical path first and this cycle overlaps it.
        KCHK(ensure_side_ws(ctx));
        ctx->prep_fused = 1;
        ctx->prep_conc = 1;
        ctx->prep_deferred = true;
        ctx->prep_b = b;
        return KNP_OK;
    }
    HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    hipStream_t main_stream = ctx->stream;
    ctx->stream = ctx->stream2;
    bool fused_norm = false;
    rc = pc_apply_norm(ctx, b, ctx->d_w, cnt, &fused_norm);
    ctx->prep_fused = fused_norm ? 1 : 0;
    ctx->prep_conc = 0;
    ctx->prep_deferred = false;
    ctx->stream = main_stream;
    KCHK(rc);
    HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
    ctx->prep_b = b;
    return KNP_OK;
}

int knp_gmres_solve(knp_ctx* ctx, const double* b, double* x, double rtol, double atol, int32_t max_it, int32_t restart,
                    int32_t* its, double* rnorm, int32_t* reason) {
    CHECK_CTX(ctx);
    if (!b || !x || !its || !rnorm || !reason) return KNP_E_ARG;
    if (!ctx->have_A) { ctx->err = "matrix not assembled"; return KNP_E_STATE; }
    // slot layout of d_red: 0 .. restart+1 = the Gram-Schmidt coefficients, the gauge coefficient and w.w of one iteration (nred = j + 2 + ns
    // <= restart + 2 values), GM_EXPL = 57 the explicit norm of the cancellation fallback, 58..63 norms / flags, 64.. deflation, 100.. the mirror
    constexpr int GM_RES = 100, GM_EXPL = 57, GM_MAX_RESTART = GM_EXPL - 2;
    static_assert(GM_RES + 2 <= RED_SLOTS && DEFL_SLOT0 + DEFL_MAX <= GM_RES, "reduction slot layout");
    if (restart < 1 || restart > GM_MAX_RESTART || max_it < 0) {
        ctx->err = "restart must be in [1," + std::to_string(GM_MAX_RESTART) + "] and max_it >= 0";
        return KNP_E_ARG;
    }
    KCHK(ensure_work(ctx, restart));
    prof_collect_ready(ctx);
    const int n = ctx->n_dof_owned;
    const int64_t ldv = ctx->n_dof_local;
    const int nb = ctx->n_red_blocks;
    hipStream_t st = ctx->stream;
    int rc;
    const int64_t cnt = ctx->ns_on ? global_phi_count(ctx, &rc) : 0;
    if (ctx->ns_on) KCHK(rc);
    const int m = restart;
    const GmLayout GL{m};
    if (ctx->gm_cap < GL.size()) {
        HIPCHK(hipStreamSynchronize(st));
        dev_free(ctx->d_gm);
        HIPCHK(hipMalloc((void**)&ctx->d_gm, (size_t)GL.size() * sizeof(double)));
        HIPCHK(hipMemset(ctx->d_gm, 0, (size_t)GL.size() * sizeof(double)));
        ctx->gm_cap = GL.size();
    }
    double* gm = ctx->d_gm;
    // residual estimate and flag of the last k_givens: pinned mirror (slots GM_RES, GM_RES+1) + sequence word, or a copy on the hook path
    auto read_state = [&](double& res_out, int& flag_out) -> int {
        ++ctx->n_readback;
        if (ctx->mirror() && ctx->h_seq_dev) {
            KCHK(read_slots_inner(ctx, GM_RES, 2, ctx->seq_counter));
            if (ctx->p2p) KCHK(knp_p2p_check(ctx));
            res_out = ctx->h_red[GM_RES];
            flag_out = (int)ctx->h_red[GM_RES + 1];
        } else {
            double tmp[3];
            HIPCHK(hipMemcpyAsync(tmp, gm + GL.st(), 3 * sizeof(double), hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            res_out = tmp[2];
            flag_out = (int)tmp[1];
        }
        return KNP_OK;
    };

    // ||M b|| for the relative tolerance (non-zero initial guess, preconditioned norm)
    bool lazy_bnorm = false;
    if (ctx->prep_b == b && restart == ctx->gm_restart && ctx->prep_conc) {
        // concurrent form: the side-stream cycle has its own vectors and slots; it is joined after the first read-back below
        lazy_bnorm = true;
    } else if (ctx->prep_b == b && restart == ctx->gm_restart) {   // already computed on the side stream (knp_gmres_prepare)
        HIPCHK(hipEventSynchronize(ctx->ev_join));
        ctx->prep_b = nullptr;
        if (!ctx->h_red_dev) {
            HIPCHK(hipMemcpy(ctx->h_red + 60, ctx->d_red + 60, 2 * sizeof(double), hipMemcpyDeviceToHost));
        }
        if (ctx->prep_fused && ctx->h_red[61] != 0.0) {   // cancellation in the one-reduction norm: redo it explicitly, in line
            ++ctx->n_norm_fallback;
            KCHK(pc_apply_proj(ctx, b, ctx->d_w, cnt));
            KCHK(dot_to_slot(ctx, ctx->d_w, ctx->d_w, 60));
            KCHK(read_slots(ctx, 60, 1, ctx->seq_counter));
        }
    } else {
        side_discard(ctx);
        bool fused_norm = false;
        KCHK(pc_apply_norm(ctx, b, ctx->d_w, cnt, &fused_norm));
        KCHK(pc_norm_read(ctx, ctx->d_w, cnt, &fused_norm));
    }
    double bnorm = lazy_bnorm ? 0.0 : std::sqrt(ctx->h_red[60]);
    ctx->last_bnorm = bnorm;
    if (!lazy_bnorm && !std::isfinite(bnorm)) { *its = 0; *rnorm = bnorm; *reason = KNP_DIVERGED_NANORINF; return KNP_OK; }
    double ttol = std::max(rtol * bnorm, atol);
    const double dtol = 1e5;
    int it = 0;
    double res = 0.0, res0 = -1.0;
    *reason = 0;
    const bool ns = ctx->ns_on && cnt > 0;
    const int nsi = ns ? 1 : 0;
    const int vec_blocks = std::min(nblocks(n), 2048);
    while (true) {
        // r = M (b - A x)
        KCHK(spmv_A(ctx, x, b, ctx->d_t, true));
        bool fused_norm = false;   // the gauge projection of r rides on the norm's reduction and on the normalisation pass
        KCHK(pc_apply_norm(ctx, ctx->d_t, ctx->d_w, cnt, &fused_norm));
        if (lazy_bnorm && ctx->prep_deferred) {   // ||B b||: its cycle goes to the side stream now, behind the launches above
            ctx->prep_deferred = false;
            HIPCHK(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
            ctx->stream = ctx->stream2;
            bool fn = false;
            swap_side_ws(ctx);
            const int rcs = pc_apply_norm(ctx, b, ctx->d_wb, cnt, &fn, true);
            swap_side_ws(ctx);
            ctx->stream = st;
            KCHK(rcs);
            HIPCHK(hipEventRecord(ctx->ev_join, ctx->stream2));
        }
        KCHK(pc_norm_read(ctx, ctx->d_w, cnt, &fused_norm));
        if (lazy_bnorm) {   // join the side stream now: its cycle ran next to the SpMV and the cycle above
            lazy_bnorm = false;
            HIPCHK(hipEventSynchronize(ctx->ev_join));
            ctx->prep_b = nullptr;
            double nb2 = ctx->h_red[SIDE_SLOT];
            if (ctx->h_red[SIDE_SLOT + 1] != 0.0) {   // cancellation in the one-reduction norm: explicit projection and norm, in line
                ++ctx->n_norm_fallback;
                KCHK(pc_apply_proj(ctx, b, ctx->d_wb, cnt));
                KCHK(dot_to_slot(ctx, ctx->d_wb, ctx->d_wb, 60));
                KCHK(read_slots(ctx, 60, 1, ctx->seq_counter));
                nb2 = ctx->h_red[60];
                // ... which used the slots and vectors of the residual norm above: redo that one (rare)
                KCHK(pc_apply_norm(ctx, ctx->d_t, ctx->d_w, cnt, &fused_norm));
                KCHK(pc_norm_read(ctx, ctx->d_w, cnt, &fused_norm));
            }
            bnorm = std::sqrt(nb2);
            ctx->last_bnorm = bnorm;
            if (!std::isfinite(bnorm)) { *its = 0; *rnorm = bnorm; *reason = KNP_DIVERGED_NANORINF; return KNP_OK; }
            ttol = std::max(rtol * bnorm, atol);
        }
        const double beta = std::sqrt(ctx->h_red[60]);
        res = beta;
        if (res0 < 0) res0 = beta;
        if (!std::isfinite(beta)) { *reason = KNP_DIVERGED_NANORINF; break; }
        if (beta <= ttol) { *reason = (beta <= atol) ? KNP_CONVERGED_ATOL : KNP_CONVERGED_RTOL; break; }
        if (it >= max_it) { *reason = KNP_DIVERGED_ITS; break; }
        if (fused_norm) {
            hipLaunchKernelGGL(k_scale_rsqrt_proj, dim3(vec_blocks), dim3(NT), 0, st, n, ctx->d_w, ctx->d_red + 60, ctx->d_red + 62,
                               1.0 / (double)cnt, ctx->d_V, gm + GL.g(), m);
        } else {
            hipLaunchKernelGGL(k_scale_rsqrt, dim3(vec_blocks), dim3(NT), 0, st, n, ctx->d_w, ctx->d_red + 60, ctx->d_V);
            hipLaunchKernelGGL(k_gm_init, dim3(1), dim3(64), 0, st, GL, gm, ctx->d_red + 60);
        }
        int jd = 0;
        bool stop = false;
        for (int j = 0; j < m; ++j) {
            double* vj = ctx->d_V + (size_t)j * ldv;
            double* vn = ctx->d_V + (size_t)(j + 1) * ldv;
            KCHK(spmv_A(ctx, vj, nullptr, ctx->d_t, false));
            // The null-space removal that follows the preconditioner (KSP_RemoveNullSpace) is folded into the
            // Gram-Schmidt pass: the basis vectors are orthogonal to ns, so h_i = V_i.(w - ns ns.w) = V_i.w, and the
            // projection itself is one more "basis vector" in the update (same reduction, no extra all-reduce).
            KCHK(pc_apply_proj(ctx, ctx->d_t, ctx->d_w, ns ? 0 : cnt));
            int flag = 0;
            {
                ProfScope ps(ctx, 1);
                // ONE reduction per iteration: the j+1 Gram-Schmidt coefficients, the gauge coefficient and w.w; the norm of the
                // orthogonalised vector follows by Pythagoras (explicit norm only when that would cancel, see k_givens)
                for (int i0 = 0; i0 <= j; i0 += 8) {
                    if (i0 == 0 && ns)
                        hipLaunchKernelGGL((k_multi_dot<8, true, true>), dim3(nb), dim3(NT), 0, st, n, ldv, i0, j + 1, ctx->d_V, ctx->d_w, ctx->d_partial);
                    else if (i0 == 0)
                        hipLaunchKernelGGL((k_multi_dot<8, false, true>), dim3(nb), dim3(NT), 0, st, n, ldv, i0, j + 1, ctx->d_V, ctx->d_w, ctx->d_partial);
                    else
                        hipLaunchKernelGGL((k_multi_dot<8, false, false>), dim3(nb), dim3(NT), 0, st, n, ldv, i0, j + 1, ctx->d_V, ctx->d_w, ctx->d_partial);
                }
                const int nred = j + 2 + nsi;
                if (fin_ok(ctx)) {   // one GPU: second reduction stage + Givens step in one single-block kernel
                    ++ctx->n_allreduce;
                    hipLaunchKernelGGL(k_reduce_fin, dim3(1), dim3(NT), 0, st, 1, nb, nred, ctx->d_partial, ctx->d_red, 0, GL, j, nsi, ns ? 1.0 / (double)cnt : 0.0, gm,
                                       0, 0.0, ctx->mirror() ? ctx->h_red_dev + GM_RES : nullptr, ctx->mirror() ? ctx->h_seq_dev : nullptr, ++ctx->seq_counter);
                } else {
                    hipLaunchKernelGGL(k_reduce_partials, dim3(nred), dim3(NT), 0, st, nb, ctx->d_partial, ctx->d_red, 0, (double*)nullptr);
                    KCHK(allreduce_slots(ctx, 0, nred));
                    hipLaunchKernelGGL(k_givens, dim3(1), dim3(64), 0, st, GL, j, nsi, ns ? 1.0 / (double)cnt : 0.0, ctx->d_red, -1, gm,
                                       ctx->mirror() ? ctx->h_red_dev + GM_RES : nullptr, ctx->mirror() ? ctx->h_seq_dev : nullptr, ++ctx->seq_counter);
                }
                hipLaunchKernelGGL(k_update_scale, dim3(nb), dim3(NT), 0, st, n, ldv, j + 1, ctx->d_V, ctx->d_red, ctx->d_w, gm + GL.st(),
                                   ns ? 1.0 / (double)cnt : 0.0, vn);
                HIPCHK(hipGetLastError());
            }
            KCHK(read_state(res, flag));
            if (flag == 1) {   // cancellation: explicit norm of the (unnormalised) vector, second reduction of this iteration
                ProfScope ps(ctx, 1);
                ++ctx->n_norm_fallback;
                hipLaunchKernelGGL(k_dot, dim3(nb), dim3(NT), 0, st, n, vn, vn, ctx->d_partial);
                hipLaunchKernelGGL(k_reduce_partials, dim3(1), dim3(NT), 0, st, nb, ctx->d_partial, ctx->d_red, GM_EXPL, (double*)nullptr);
                KCHK(allreduce_slots(ctx, GM_EXPL, 1));
                hipLaunchKernelGGL(k_givens, dim3(1), dim3(64), 0, st, GL, j, nsi, ns ? 1.0 / (double)cnt : 0.0, ctx->d_red, GM_EXPL, gm,
                                   ctx->mirror() ? ctx->h_red_dev + GM_RES : nullptr, ctx->mirror() ? ctx->h_seq_dev : nullptr, ++ctx->seq_counter);
                hipLaunchKernelGGL(k_scale_inplace_rsqrt, dim3(vec_blocks), dim3(NT), 0, st, n, gm + GL.st(), vn);
                HIPCHK(hipGetLastError());
                KCHK(read_state(res, flag));
            }
            if (flag != 0 || !std::isfinite(res)) { *reason = KNP_DIVERGED_NANORINF; stop = true; jd = j; break; }
            ++it;
            jd = j + 1;
            if (res <= ttol) { *reason = (res <= atol) ? KNP_CONVERGED_ATOL : KNP_CONVERGED_RTOL; stop = true; break; }
            if (it >= max_it) { *reason = KNP_DIVERGED_ITS; stop = true; break; }
            if (res > dtol * res0) { *reason = KNP_DIVERGED_DTOL; stop = true; break; }
        }
        if (jd > 0) {
            ProfScope ps(ctx, 1);
            if (jd <= 8) {
                hipLaunchKernelGGL(k_lincomb_solve, dim3(vec_blocks), dim3(NT), 0, st, GL, jd, gm, n, ldv, ctx->d_V, x);
            } else {
                hipLaunchKernelGGL(k_gm_solve_y, dim3(1), dim3(64), 0, st, GL, jd, gm);
                hipLaunchKernelGGL(k_lincomb, dim3(vec_blocks), dim3(NT), 0, st, n, ldv, jd, ctx->d_V, gm + GL.y(), x);
            }
        }
        if (stop) break;
    }
    *its = it;
    *rnorm = res;
    KCHK(halo_update(ctx, x));
    HIPCHK(hipGetLastError());
    if (ctx->p2p) {   // the final halo is two asynchronous kernels: make its outcome (and any timeout) known before returning
        HIPCHK(hipStreamSynchronize(st));
        KCHK(knp_p2p_check(ctx));
    }
    if (ctx->comm_rc != KNP_OK) { const int rc2 = ctx->comm_rc; ctx->comm_rc = KNP_OK; return rc2; }
    return KNP_OK;
}

// ---- state transfer -------------------------------------------------------------------------
static int out_ptrs(knp_ctx* ctx, const knp_fields_out* f, OutPtrs& o, bool need_phim) {
    if (!f) { ctx->err = "null fields"; return KNP_E_ARG; }
    for (int j = 0; j < 3; ++j) {
        if (!f->k_i[j] || !f->k_e[j]) { ctx->err = "null concentration field"; return KNP_E_ARG; }
        o.ki[j] = f->k_i[j]; o.ke[j] = f->k_e[j];
    }
    if (!f->phi_i || !f->phi_e || (need_phim && !f->phi_m)) { ctx->err = "null potential field"; return KNP_E_ARG; }
    o.phi_i = f->phi_i; o.phi_e = f->phi_e; o.phi_m = f->phi_m;
    return KNP_OK;
}
int knp_pack(knp_ctx* ctx, const knp_fields_out* f, double* x) {
    CHECK_CTX(ctx);
    if (!x) return KNP_E_ARG;
    OutPtrs o;
    KCHK(out_ptrs(ctx, f, o, false));
    hipLaunchKernelGGL(k_pack, dim3(nblocks(ctx->g.n_nodes)), dim3(NT), 0, ctx->stream, ctx->g.n_nodes, ctx->d_node_vertex,
                       ctx->d_node_side, o, x);
    HIPCHK(hipGetLastError());
    return KNP_OK;
}
int knp_unpack(knp_ctx* ctx, const double* x, const knp_fields_out* f) {
    CHECK_CTX(ctx);
    if (!x) return KNP_E_ARG;
    OutPtrs o;
    KCHK(out_ptrs(ctx, f, o, true));
    KCHK(join_asm(ctx));   // the fields written here are what an assembly still in flight reads
    KCHK(halo_update(ctx, const_cast<double*>(x)));
    hipLaunchKernelGGL(k_unpack, dim3(nblocks(ctx->g.n_v)), dim3(NT), 0, ctx->stream, ctx->g.n_v, ctx->d_node_i, ctx->d_node_e, x, o);
    HIPCHK(hipGetLastError());
    return KNP_OK;
}
int knp_hh_update(knp_ctx* ctx, const double* phi_m, double* n, double* m, double* h, int32_t count, double dt, double phi_rest,
                  int32_t rush_larsen, int32_t substeps) {
    CHECK_CTX(ctx);
    if (!phi_m || !n || !m || !h || count < 0 || substeps < 1 || !(dt > 0)) { ctx->err = "bad hh_update arguments"; return KNP_E_ARG; }
    if (count == 0) return KNP_OK;
    ProfScope ps(ctx, 4);
    hipLaunchKernelGGL(k_hh_update, dim3(nblocks(count)), dim3(NT), 0, ctx->stream, count, phi_m, n, m, h, dt, phi_rest,
                       rush_larsen, substeps);
    HIPCHK(hipGetLastError());
    return KNP_OK;
}
int knp_l2_norms(knp_ctx* ctx, const double* phi_i, const double* phi_e, double* out) {
    CHECK_CTX(ctx);
    side_discard(ctx);
    if (!phi_i || !phi_e || !out) return KNP_E_ARG;
    const KnpHostGraph& g = ctx->g;
    const int nb = std::min(RED_BLOCKS, nblocks(g.n_c_owned));
    if (g.dim == 2)
        hipLaunchKernelGGL((k_l2<2>), dim3(nb), dim3(NT), 0, ctx->stream, g.n_c_owned, ctx->d_cells, ctx->d_cell_side, ctx->d_coords, phi_i, phi_e, ctx->d_partial);
    else
        hipLaunchKernelGGL((k_l2<3>), dim3(nb), dim3(NT), 0, ctx->stream, g.n_c_owned, ctx->d_cells, ctx->d_cell_side, ctx->d_coords, phi_i, phi_e, ctx->d_partial);
    hipLaunchKernelGGL(k_reduce_partials, dim3(2), dim3(NT), 0, ctx->stream, nb, ctx->d_partial, ctx->d_red, 58, ctx->mirror());
    HIPCHK(hipGetLastError());
    KCHK(read_slots(ctx, 58, 2));
    out[0] = ctx->h_red[58];
    out[1] = ctx->h_red[59];
    return KNP_OK;
}

// ---- instrumentation --------------------------------------------------------------------------
// Step timers of the host loop (the reference brackets assembly and solve with perf_counter + allreduce(MAX),
// KNPEMIx_solver.py:402-413,434-449): a mark is one hipEventRecord on the main stream from a recycled pool -- after joining the
// matrix assembly that may run on its own stream, so that a mark after the assembly phase covers both chains -- and
// knp_timer_read returns the elapsed seconds between consecutive marks with ONE synchronisation.
int knp_timer_mark(knp_ctx* ctx, int32_t join_assembly) {
    CHECK_CTX(ctx);
    if (join_assembly) KCHK(join_asm(ctx));
    if (ctx->tm_used == ctx->tm_events.size()) {
        hipEvent_t e = nullptr;
        HIPCHK(hipEventCreate(&e));
        ctx->tm_events.push_back(e);
    }
    HIPCHK(hipEventRecord(ctx->tm_events[ctx->tm_used++], ctx->stream));
    return KNP_OK;
}
int knp_timer_read(knp_ctx* ctx, int32_t capacity, double* seconds, int32_t* n_intervals) {
    CHECK_CTX(ctx);
    if (!seconds || !n_intervals || capacity < 0) return KNP_E_ARG;
    const size_t n = ctx->tm_used;
    *n_intervals = 0;
    if (n == 0) return KNP_OK;
    if ((size_t)capacity + 1 < n) { ctx->err = "knp_timer_read: capacity below the number of marks - 1"; return KNP_E_ARG; }
    HIPCHK(hipEventSynchronize(ctx->tm_events[n - 1]));
    for (size_t i = 0; i + 1 < n; ++i) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ctx->tm_events[i], ctx->tm_events[i + 1]));
        seconds[i] = 1e-3 * (double)ms;
    }
    *n_intervals = (int32_t)(n - 1);
    ctx->tm_used = 0;
    return KNP_OK;
}
int knp_timer_pending(const knp_ctx* ctx) { return ctx ? (int)ctx->tm_used : 0; }

int knp_profile_enable(knp_ctx* ctx, int32_t on) {
    CHECK_CTX(ctx);
    KCHK(prof_collect(ctx));
    ctx->prof_on = on;   // bit k enables class k
    return KNP_OK;
}
int knp_profile_get(knp_ctx* ctx, int32_t cls, double* ms, int64_t* launches) {
    CHECK_CTX(ctx);
    if (cls < 0 || cls >= KNP_NPROF || !ms || !launches) return KNP_E_ARG;
    KCHK(prof_collect(ctx));
    *ms = ctx->prof_ms[cls];
    *launches = ctx->prof_n[cls];
    return KNP_OK;
}
int knp_profile_reset(knp_ctx* ctx) {
    CHECK_CTX(ctx);
    KCHK(prof_collect(ctx));
    for (int i = 0; i < KNP_NPROF; ++i) { ctx->prof_ms[i] = 0; ctx->prof_n[i] = 0; }
    ctx->n_allreduce = ctx->n_halo = ctx->n_readback = ctx->n_norm_fallback = 0;
    return KNP_OK;
}
// Bytes the kernels of one application must move, from the sizes of the arrays they read and write (the "algorithmic bytes" of the
// per-class roofline): [0] SpMV on A, [1] one preconditioner application (all hierarchies of the current kind, the path
// knp_pc_setup selected), [2] matrix assembly of one step (entries that depend on the previous solution), [3] right-hand side
// assembly, [4] bytes of ONE owned vector (the orthogonalisation of iteration j moves (2 (j + 1) + 3) of these).
static double cycle_bytes(const knp_ctx* ctx, const KnpAmgHier& H) {
    if (H.levels < 1) return 0.0;
    const double vs = H.cinv_f || ctx->amg_fp32 ? 4.0 : 8.0;
    double b = 0.0;
    const int nl = H.levels;
    const int nf0 = H.native0 == 1 ? 4 : H.native0 == 2 ? 3 : 1;
    const double n0 = (double)ctx->g.n_nodes_owned;
    auto blocked = [&](const KnpBlockedCsr& M, int nf) { return (double)M.nnz * (nf == 4 ? 20.0 : 16.0) + 4.0 * (M.n_rows + 1); };
    auto scalar = [&](int64_t nnz, int rows) { return (double)nnz * (vs + 4.0) + 4.0 * (rows + 1); };
    // level 0, down: P Dinv of the library's own P (values per pair and field + neighbour index) or the uploaded compact operator
    if (H.at0_rp) b += scalar(H.lv[0].A_nnz, ctx->g.n_nodes_owned);
    else if (H.native0) b += (double)ctx->n_pairs * (vs * nf0 + 4.0) + 4.0 * (n0 + 1);
    else b += scalar(H.lv[0].A_nnz, H.lv[0].n);
    b += 8.0 * n0 * nf0 * 3.0;                                     // b gathered, b read, r written
    for (int l = 0; l < nl - 1; ++l) {
        const KnpAmgLevel& L = H.lv[l];
        const double nl_ = (double)(l == 0 ? n0 * nf0 : L.n), nc_ = (double)L.n_coarse;
        const bool mid = l >= 1 && H.cfused;
        if (mid) {                                                  // Rt down, U up: operator + input gathered + output written
            b += (H.blocked ? blocked(L.bRt, H.node_nf) : scalar(L.Rt_nnz, L.n_coarse)) + 8.0 * (nl_ + nc_);
            b += (H.blocked ? blocked(L.bU, H.node_nf) : scalar(L.U_nnz, L.n)) + 8.0 * (nl_ + nc_ + nl_);
        } else {                                                    // R down; S up with b, r, dinv and the output
            b += (H.blocked ? blocked(L.bR, H.node_nf) : scalar(L.R_nnz, L.n_coarse)) + 8.0 * (nl_ + nc_);
            b += (H.blocked ? blocked(L.bS, H.node_nf) : scalar(L.S_nnz, L.n)) + 8.0 * (nc_ + 4.0 * nl_);
            if (l >= 1) b += (H.blocked ? blocked(L.bA, H.node_nf) : scalar(L.A_nnz, L.n)) + 8.0 * 3.0 * nl_;   // residual of the level
        }
    }
    if (H.nc > 0) b += (double)H.nc * H.nc * (H.cinv_f ? 4.0 : 8.0) + 16.0 * H.nc;
    return b;
}
int knp_get_traffic_model(const knp_ctx* ctx, double* out) {
    if (!ctx || !out) return KNP_E_ARG;
    const KnpHostGraph& g = ctx->g;
    const double np_ = (double)ctx->n_pairs, no = (double)g.n_nodes_owned, ngp = (double)ctx->n_gp;
    static const bool mf_off = getenv("KNP_SPMV_MF") && atoi(getenv("KNP_SPMV_MF")) == 0;
    const bool mf = !mf_off && ctx->n_bc == 0 && ctx->d_pair_MK != nullptr;
    // SpMV on A: per pair a_t (32 B), {M, K} (16 B) or a_c (48 B), the neighbour index; per node pointer, membrane index, side;
    // per membrane pair 32 B + a column; x gathered through the caches is counted once (8 n_local), y written
    out[0] = np_ * (32.0 + (mf ? 16.0 : 48.0) + 4.0) + no * (4.0 + 4.0 + 1.0) + 2.0 * ngp * (32.0 + 4.0) + 8.0 * ctx->n_dof_owned + 8.0 * ctx->n_dof_local;
    double pc = 0.0;
    if (ctx->pc_kind == KNP_PC_AMG) pc = cycle_bytes(ctx, ctx->hier[0]);
    else if (ctx->pc_kind == KNP_PC_AMG_BT || ctx->pc_kind == KNP_PC_AMG_LT)
        pc = cycle_bytes(ctx, ctx->hier[0]) + cycle_bytes(ctx, ctx->hier[1]) + np_ * 12.0 + no * 8.0 * 9.0;   // + the potential right-hand side (k_phi_rhs)
    else if (ctx->pc_kind == KNP_PC_VBJACOBI) pc = no * 8.0 * (16.0 + 8.0);
    out[1] = pc;
    // matrix assembly per step (SURVEY B_asm_step in this layout): contribution lists (8 B value + 1 B slot), the cell means staged per node,
    // a_t written (32 B per pair); + the membrane entries (facet matrices read, a_t slots updated, a_x written)
    // the previous concentrations: node-indexed copy + one 32-byte record per neighbour + the byte table of the cells' vertices
    // (fused cell means), or the per-cell pass (connectivity, 3 (d+1) gathers, a 32-byte record written) + a record read per cell of a node
    const double conc = ctx->asm_dmax > 0 ? (double)g.n_nodes * 56.0 + np_ * 36.0 + (double)ctx->n_node_cells * g.nv1
                                          : (double)g.n_c * (4.0 * g.nv1 + 32.0 + 24.0 * g.nv1) + (double)ctx->n_node_cells * 36.0;
    out[2] = (double)(ctx->n_tc > 0 ? ctx->n_tc : ctx->n_contrib) * 9.0 + conc + np_ * 32.0 + ngp * (64.0 + 48.0);
    out[3] = no * (8.0 * 4.0 + 9.0) + np_ * (8.0 + 4.0 + 24.0) + (double)g.n_g * g.dim * 7.0 * 8.0 * 2.0;
    out[4] = 8.0 * ctx->n_dof_owned;
    return KNP_OK;
}
int knp_get_stats(const knp_ctx* ctx, double* out) {
    if (!ctx || !out) return KNP_E_ARG;
    for (int i = 0; i < KNP_ST_COUNT; ++i) out[i] = 0.0;
    out[KNP_ST_BNORM] = ctx->last_bnorm;
    out[KNP_ST_ALLREDUCE] = (double)ctx->n_allreduce;
    out[KNP_ST_HALO] = (double)ctx->n_halo;
    out[KNP_ST_READBACK] = (double)ctx->n_readback;
    out[KNP_ST_FUSED] = (double)(ctx->hier[0].fused + 2 * ctx->hier[1].fused + 4 * ctx->hier[0].l0_fused + 8 * ctx->hier[1].l0_fused);
    out[KNP_ST_NORM_FALLBACK] = (double)ctx->n_norm_fallback;
    out[KNP_ST_BLOCKED] = (double)(ctx->hier[0].blocked + 2 * ctx->hier[1].blocked);
    int nlf = 0;
    for (int h = 0; h < KNP_MAX_HIER; ++h)
        for (int l = 1; l < ctx->hier[h].levels; ++l) nlf += ctx->hier[h].lv[l].lfused;
    out[KNP_ST_FUSED_LEVELS] = (double)nlf;
    return KNP_OK;
}

}  // extern "C"
