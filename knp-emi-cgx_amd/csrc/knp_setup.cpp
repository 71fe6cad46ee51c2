// Host-side graph builder for libknpemi_hip: DoF layout of the restricted intra/extra spaces,
// same-side node graph with per-pair cell contribution lists, membrane (Gamma) vertex graph, and
// the CSR pattern of the block system.  Replaces what DOLFINx/multiphenicsx do inside
// DofMapRestriction + create_matrix_block (reference: src/CGx/KNPEMI/KNPEMIx_problem.py:75-94,
// src/CGx/KNPEMI/KNPEMIx_solver.py:157-161) -- but built so that assembly on the GPU is a
// deterministic gather (no hashing, no atomics).
#include <cstdlib>
#include <cstdio>
#include <sched.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "knp_internal.hpp"

namespace {

struct Tup {
    int32_t nb, cell;
    int8_t la, lb;
    int32_t slot;   // position of the cell in the row node's cell list
};

inline int node_of(const KnpHostGraph& g, int v, int side) { return side == 0 ? g.node_i[v] : g.node_e[v]; }

// volume and barycentric gradients of one P1 simplex
inline bool cell_geom(int dim, const double* X /* (dim+1) x dim */, double& vol, double G[4][3]) {
    if (dim == 2) {
        double a = X[2] - X[0], b = X[3] - X[1];  // v1 - v0
        double c = X[4] - X[0], d = X[5] - X[1];  // v2 - v0
        double det = a * d - b * c;
        if (det == 0.0) return false;
        vol = std::fabs(det) * 0.5;
        double inv = 1.0 / det;
        // grad lambda_1, lambda_2 = rows of J^{-1} where J = [[a,b],[c,d]] maps ref->phys rows
        G[1][0] = d * inv;  G[1][1] = -c * inv;
        G[2][0] = -b * inv; G[2][1] = a * inv;
        G[0][0] = -G[1][0] - G[2][0];
        G[0][1] = -G[1][1] - G[2][1];
        return true;
    }
    double e1[3], e2[3], e3[3];
    for (int k = 0; k < 3; ++k) {
        e1[k] = X[3 + k] - X[k];
        e2[k] = X[6 + k] - X[k];
        e3[k] = X[9 + k] - X[k];
    }
    double c23[3] = {e2[1] * e3[2] - e2[2] * e3[1], e2[2] * e3[0] - e2[0] * e3[2], e2[0] * e3[1] - e2[1] * e3[0]};
    double c31[3] = {e3[1] * e1[2] - e3[2] * e1[1], e3[2] * e1[0] - e3[0] * e1[2], e3[0] * e1[1] - e3[1] * e1[0]};
    double c12[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    double det = e1[0] * c23[0] + e1[1] * c23[1] + e1[2] * c23[2];
    if (det == 0.0) return false;
    vol = std::fabs(det) / 6.0;
    double inv = 1.0 / det;
    for (int k = 0; k < 3; ++k) {
        G[1][k] = c23[k] * inv;
        G[2][k] = c31[k] * inv;
        G[3][k] = c12[k] * inv;
        G[0][k] = -G[1][k] - G[2][k] - G[3][k];
    }
    return true;
}

}  // namespace

int knp_build_graph(const knp_mesh_desc* m, KnpHostGraph& g) {
    if (!m || (m->dim != 2 && m->dim != 3)) { g.error = "dim must be 2 or 3"; return KNP_E_ARG; }
    if (m->n_vertices <= 0 || m->n_cells <= 0 || !m->cells || !m->coords || !m->cell_side) {
        g.error = "empty mesh or null pointers"; return KNP_E_ARG;
    }
    if (m->n_vertices_owned < 0 || m->n_vertices_owned > m->n_vertices) { g.error = "bad n_vertices_owned"; return KNP_E_ARG; }
    if (m->n_gamma > 0 && (!m->gamma || !m->gamma_prog)) { g.error = "gamma arrays missing"; return KNP_E_ARG; }
    if (m->n_q <= 0 || !m->q_pts || !m->q_w) { g.error = "facet quadrature missing"; return KNP_E_ARG; }
    const int dim = m->dim, nv1 = dim + 1;
    g.dim = dim; g.nv1 = nv1;
    g.n_v = m->n_vertices; g.n_v_owned = m->n_vertices_owned;
    g.n_c = m->n_cells; g.n_c_owned = (m->n_cells_owned > 0 && m->n_cells_owned <= m->n_cells) ? m->n_cells_owned : m->n_cells;
    g.n_g = m->n_gamma; g.n_q = m->n_q;
    const int32_t* cells = m->cells;
    const uint8_t* side = m->cell_side;

    // ---- 1. layout --------------------------------------------------------------------
    std::vector<uint8_t> in_i(g.n_v, 0), in_e(g.n_v, 0);
    for (int c = 0; c < g.n_c; ++c) {
        if (side[c] > 1) { g.error = "cell_side must be 0 or 1"; return KNP_E_MESH; }
        for (int a = 0; a < nv1; ++a) {
            int v = cells[(size_t)c * nv1 + a];
            if (v < 0 || v >= g.n_v) { g.error = "cell vertex index out of range"; return KNP_E_MESH; }
            (side[c] == 0 ? in_i : in_e)[v] = 1;
        }
    }
    g.node_i.assign(g.n_v, -1);
    g.node_e.assign(g.n_v, -1);
    int nn = 0;
    g.n_nodes_owned = 0;
    for (int v = 0; v < g.n_v; ++v) {
        if (v == g.n_v_owned) g.n_nodes_owned = nn;
        if (in_i[v]) g.node_i[v] = nn++;
        if (in_e[v]) g.node_e[v] = nn++;
    }
    if (g.n_v_owned == g.n_v) g.n_nodes_owned = nn;
    g.n_nodes = nn;
    g.node_vertex.resize(nn);
    g.node_side.resize(nn);
    for (int v = 0; v < g.n_v; ++v) {
        if (g.node_i[v] >= 0) { g.node_vertex[g.node_i[v]] = v; g.node_side[g.node_i[v]] = 0; }
        if (g.node_e[v] >= 0) { g.node_vertex[g.node_e[v]] = v; g.node_side[g.node_e[v]] = 1; }
    }
    const int no = g.n_nodes_owned;

    // ---- 2. node -> cells (same side), owned nodes only --------------------------------
    std::vector<int32_t> nc_ptr(no + 1, 0);
    for (int c = 0; c < g.n_c; ++c)
        for (int a = 0; a < nv1; ++a) {
            int n = node_of(g, cells[(size_t)c * nv1 + a], side[c]);
            if (n < no) nc_ptr[n + 1]++;
        }
    for (int n = 0; n < no; ++n) nc_ptr[n + 1] += nc_ptr[n];
    std::vector<int32_t> nc(nc_ptr[no]);
    {
        std::vector<int32_t> fill(nc_ptr.begin(), nc_ptr.end() - 1);
        for (int c = 0; c < g.n_c; ++c)
            for (int a = 0; a < nv1; ++a) {
                int n = node_of(g, cells[(size_t)c * nv1 + a], side[c]);
                if (n < no) nc[fill[n]++] = c * 4 + a;  // cell*4 + local index of the node's vertex
            }
    }

    // ---- 3. pairs + contributions: count pass then fill pass -----------------------------
    std::vector<int32_t> npair(no + 1, 0);
    std::vector<int64_t> ncon(no + 1, 0);
#pragma omp parallel num_threads(knp_host_threads())
    {
        std::vector<int32_t> nbs;
#pragma omp for schedule(dynamic, 4096)
        for (int n = 0; n < no; ++n) {
            nbs.clear();
            for (int k = nc_ptr[n]; k < nc_ptr[n + 1]; ++k) {
                int c = nc[k] >> 2;
                for (int b = 0; b < nv1; ++b) nbs.push_back(node_of(g, cells[(size_t)c * nv1 + b], side[c]));
            }
            // contributions of the self pair are not stored: sum_b K_ab(T) = 0 on every simplex, so the self entry of the
            // concentration-weighted stiffness is minus the sum over the node's other pairs (k_assemble_nodes)
            ncon[n + 1] = (int64_t)nbs.size() - (nc_ptr[n + 1] - nc_ptr[n]);
            std::sort(nbs.begin(), nbs.end());
            npair[n + 1] = (int32_t)(std::unique(nbs.begin(), nbs.end()) - nbs.begin());
        }
    }
    int64_t tot_pairs = 0, tot_con = 0;
    for (int n = 0; n < no; ++n) { tot_pairs += npair[n + 1]; tot_con += ncon[n + 1]; }
    if (tot_pairs > 0x7fffffffLL / 10 * 9 || tot_con > 0x7fffffffLL) { g.error = "local problem too large for int32 indices"; return KNP_E_MESH; }
    g.pair_ptr.assign(no + 1, 0);
    for (int n = 0; n < no; ++n) g.pair_ptr[n + 1] = g.pair_ptr[n] + npair[n + 1];
    std::vector<int64_t> con_start(no + 1, 0);
    for (int n = 0; n < no; ++n) con_start[n + 1] = con_start[n] + ncon[n + 1];
    const int64_t np = tot_pairs;
    g.pair_col.resize(np); g.pair_row.resize(np); g.pair_M.assign(np, 0.0); g.pair_K.assign(np, 0.0);
    g.contrib_ptr.assign(np + 1, 0);
    g.contrib_cell.resize(tot_con); g.contrib_k.resize(tot_con); g.contrib_slot.resize(tot_con);
    g.node_cell_ptr.assign(nc_ptr.begin(), nc_ptr.end());
    g.node_cell.resize(nc.size());
    g.max_node_cells = 0;
    for (int n = 0; n < no; ++n) g.max_node_cells = std::max(g.max_node_cells, nc_ptr[n + 1] - nc_ptr[n]);
    for (size_t k = 0; k < nc.size(); ++k) g.node_cell[k] = nc[k] >> 2;
    const double mfac = 1.0 / ((dim + 1.0) * (dim + 2.0));
    bool degenerate = false;
#pragma omp parallel num_threads(knp_host_threads())
    {
        std::vector<Tup> tups;
#pragma omp for schedule(dynamic, 4096)
        for (int n = 0; n < no; ++n) {
            tups.clear();
            for (int k = nc_ptr[n]; k < nc_ptr[n + 1]; ++k) {
                int c = nc[k] >> 2, la = nc[k] & 3;
                for (int b = 0; b < nv1; ++b)
                    tups.push_back({node_of(g, cells[(size_t)c * nv1 + b], side[c]), c, (int8_t)la, (int8_t)b, (int32_t)(k - nc_ptr[n])});
            }
            std::stable_sort(tups.begin(), tups.end(), [](const Tup& x, const Tup& y) { return x.nb < y.nb; });
            int64_t p = g.pair_ptr[n] - 1;
            int64_t cc = con_start[n];
            int prev = -1;
            for (const Tup& t : tups) {
                if (t.nb != prev) {
                    ++p;
                    prev = t.nb;
                    g.pair_col[p] = t.nb;
                    g.pair_row[p] = n;
                    g.contrib_ptr[p] = (int32_t)cc;
                }
                double X[12], vol, G[4][3];
                for (int a = 0; a < nv1; ++a)
                    for (int k = 0; k < dim; ++k) X[a * dim + k] = m->coords[(size_t)cells[(size_t)t.cell * nv1 + a] * dim + k];
                if (!cell_geom(dim, X, vol, G)) { degenerate = true; vol = 0; std::memset(G, 0, sizeof(G)); }
                double dot = 0;
                for (int k = 0; k < dim; ++k) dot += G[t.la][k] * G[t.lb][k];
                double kab = vol * dot;
                if (t.nb != n) {
                    g.contrib_cell[cc] = t.cell;
                    g.contrib_k[cc] = kab;
                    g.contrib_slot[cc] = (uint8_t)std::min(t.slot, 255);
                    ++cc;
                }
                g.pair_K[p] += kab;
                g.pair_M[p] += vol * mfac * (t.la == t.lb ? 2.0 : 1.0);
            }
        }
    }
    g.contrib_ptr[np] = (int32_t)tot_con;
    if (degenerate) { g.error = "degenerate (zero-volume) cell in mesh"; return KNP_E_MESH; }

    // ---- 4. membrane graph ---------------------------------------------------------------
    const int ng = g.n_g;
    g.fv.resize((size_t)ng * dim);
    g.fmeas.resize(ng);
    for (int f = 0; f < ng; ++f) {
        int cp = m->gamma[4 * f], lp = m->gamma[4 * f + 1], cm = m->gamma[4 * f + 2];
        if (cp < 0 || cp >= g.n_c || cm < 0 || cm >= g.n_c || lp < 0 || lp >= nv1) { g.error = "gamma entry out of range"; return KNP_E_MESH; }
        if (side[cp] != 0 || side[cm] != 1) { g.error = "gamma facet: '+' cell must be intracellular and '-' extracellular"; return KNP_E_MESH; }
        int k = 0;
        for (int a = 0; a < nv1; ++a)
            if (a != lp) g.fv[(size_t)f * dim + k++] = cells[(size_t)cp * nv1 + a];
        const double* x0 = &m->coords[(size_t)g.fv[(size_t)f * dim] * dim];
        const double* x1 = &m->coords[(size_t)g.fv[(size_t)f * dim + 1] * dim];
        if (dim == 2) {
            g.fmeas[f] = std::hypot(x1[0] - x0[0], x1[1] - x0[1]);
        } else {
            const double* x2 = &m->coords[(size_t)g.fv[(size_t)f * dim + 2] * dim];
            double u[3] = {x1[0] - x0[0], x1[1] - x0[1], x1[2] - x0[2]}, w[3] = {x2[0] - x0[0], x2[1] - x0[1], x2[2] - x0[2]};
            double cx = u[1] * w[2] - u[2] * w[1], cy = u[2] * w[0] - u[0] * w[2], cz = u[0] * w[1] - u[1] * w[0];
            g.fmeas[f] = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
        }
        for (int a = 0; a < dim; ++a) {
            int v = g.fv[(size_t)f * dim + a];
            if (g.node_i[v] < 0 || g.node_e[v] < 0) { g.error = "gamma facet vertex lacks an intra or extra node"; return KNP_E_MESH; }
        }
    }
    // owned membrane vertices
    std::vector<int32_t> gv_of_vertex(g.n_v, -1);
    {
        std::vector<uint8_t> on_g(g.n_v, 0);
        for (size_t k = 0; k < g.fv.size(); ++k) on_g[g.fv[k]] = 1;
        g.n_gv = 0;
        for (int v = 0; v < g.n_v_owned; ++v)
            if (on_g[v]) {
                gv_of_vertex[v] = g.n_gv++;
                g.gv_vertex.push_back(v);
                g.gv_node_i.push_back(g.node_i[v]);
                g.gv_node_e.push_back(g.node_e[v]);
            }
    }
    g.node_gv.assign(no, -1);
    for (int A = 0; A < g.n_gv; ++A) { g.node_gv[g.gv_node_i[A]] = A; g.node_gv[g.gv_node_e[A]] = A; }
    // membrane vertex -> facets
    std::vector<int32_t> vf_ptr(g.n_gv + 1, 0);
    for (int f = 0; f < ng; ++f)
        for (int a = 0; a < dim; ++a) {
            int A = gv_of_vertex[g.fv[(size_t)f * dim + a]];
            if (A >= 0) vf_ptr[A + 1]++;
        }
    for (int A = 0; A < g.n_gv; ++A) vf_ptr[A + 1] += vf_ptr[A];
    std::vector<int32_t> vf(vf_ptr[g.n_gv]);
    {
        std::vector<int32_t> fill(vf_ptr.begin(), vf_ptr.end() - 1);
        for (int f = 0; f < ng; ++f)
            for (int a = 0; a < dim; ++a) {
                int A = gv_of_vertex[g.fv[(size_t)f * dim + a]];
                if (A >= 0) vf[fill[A]++] = f * 4 + a;
            }
    }
    g.gptr.assign(g.n_gv + 1, 0);
    g.gdiag.assign(g.n_gv, -1);
    struct GT { int32_t vb, facet; int8_t la, lb; };
    std::vector<GT> gt;
    for (int A = 0; A < g.n_gv; ++A) {
        gt.clear();
        for (int k = vf_ptr[A]; k < vf_ptr[A + 1]; ++k) {
            int f = vf[k] >> 2, la = vf[k] & 3;
            for (int b = 0; b < dim; ++b) gt.push_back({g.fv[(size_t)f * dim + b], f, (int8_t)la, (int8_t)b});
        }
        std::stable_sort(gt.begin(), gt.end(), [](const GT& x, const GT& y) { return x.vb < y.vb; });
        int prev = -1;
        for (const GT& t : gt) {
            if (t.vb != prev) {
                prev = t.vb;
                if (t.vb == g.gv_vertex[A]) g.gdiag[A] = (int32_t)g.gcol.size();
                g.gcol.push_back(t.vb);
                g.grow.push_back(A);
                g.gcptr.push_back((int32_t)g.gc_facet.size());
            }
            g.gc_facet.push_back(t.facet);
            g.gc_lab.push_back(t.la * 4 + t.lb);
        }
        g.gptr[A + 1] = (int32_t)g.gcol.size();
    }
    g.gcptr.push_back((int32_t)g.gc_facet.size());
    const int64_t ngp = (int64_t)g.gcol.size();
    g.gq_i.resize(ngp); g.gq_e.resize(ngp); g.gx_i.resize(ngp); g.gx_e.resize(ngp);
    for (int64_t s = 0; s < ngp; ++s) {
        int A = g.grow[s], vb = g.gcol[s];
        int ni = g.gv_node_i[A], ne = g.gv_node_e[A];
        int nbi = g.node_i[vb], nbe = g.node_e[vb];
        auto find = [&](int n, int nb) -> int {
            const int32_t* b = &g.pair_col[g.pair_ptr[n]];
            const int32_t* e = &g.pair_col[g.pair_ptr[n + 1]];
            const int32_t* it = std::lower_bound(b, e, nb);
            return (it != e && *it == nb) ? (int)(it - b) : -1;
        };
        g.gq_i[s] = find(ni, nbi);
        g.gq_e[s] = find(ne, nbe);
        if (g.gq_i[s] < 0 || g.gq_e[s] < 0) { g.error = "membrane pair without a same-side edge (inconsistent gamma data)"; return KNP_E_MESH; }
        g.gx_i[s] = nbe;  // cross column of the intra row  = extra node of the neighbour
        g.gx_e[s] = nbi;  // cross column of the extra row  = intra node of the neighbour
    }

    // ---- 5. size of A (the CSR pattern itself is built on demand: knp_build_csr_pattern) ----------------------------
    {
        int64_t nnz = 10 * np;
        for (int n = 0; n < no; ++n) {
            int A = g.node_gv[n];
            if (A >= 0) nnz += 4 * (int64_t)(g.gptr[A + 1] - g.gptr[A]);
        }
        if (nnz > 0x7fffffffLL) { g.error = "nnz exceeds int32"; return KNP_E_MESH; }
        g.nnz = nnz;
    }
    return KNP_OK;
}

// CSR pattern of A in the documented row order (export / parity hook):
//   row (n,j<3): [ (nb,j),(nb,3) for nb in pairs ] ++ [ (cross,3) ]   length 2*deg + x
//   row (n,3)  : [ (nb,0..3) for nb in pairs ]     ++ [ (cross,3) ]   length 4*deg + x
int knp_build_csr_pattern(KnpHostGraph& g) {
    const int no = g.n_nodes_owned;
    if (!g.rowptr.empty()) return KNP_OK;
    g.rowptr.assign((size_t)4 * no + 1, 0);
    int64_t nnz = 0;
    for (int n = 0; n < no; ++n) {
        int deg = g.pair_ptr[n + 1] - g.pair_ptr[n];
        int A = g.node_gv[n];
        int x = A >= 0 ? g.gptr[A + 1] - g.gptr[A] : 0;
        for (int f = 0; f < 4; ++f) {
            g.rowptr[(size_t)4 * n + f] = (int32_t)nnz;
            nnz += (f < 3 ? 2 : 4) * (int64_t)deg + x;
        }
    }
    g.rowptr[(size_t)4 * no] = (int32_t)nnz;
    g.colind.resize(nnz);
#pragma omp parallel for schedule(static) num_threads(knp_host_threads())
    for (int n = 0; n < no; ++n) {
        int p0 = g.pair_ptr[n], deg = g.pair_ptr[n + 1] - p0;
        int A = g.node_gv[n];
        int x = A >= 0 ? g.gptr[A + 1] - g.gptr[A] : 0;
        const int32_t* cross = nullptr;
        if (A >= 0) cross = (g.node_side[n] == 0 ? g.gx_i.data() : g.gx_e.data()) + g.gptr[A];
        for (int f = 0; f < 3; ++f) {
            int32_t* ci = &g.colind[g.rowptr[(size_t)4 * n + f]];
            for (int q = 0; q < deg; ++q) { ci[2 * q] = 4 * g.pair_col[p0 + q] + f; ci[2 * q + 1] = 4 * g.pair_col[p0 + q] + 3; }
            for (int r = 0; r < x; ++r) ci[2 * deg + r] = 4 * cross[r] + 3;
        }
        int32_t* ci = &g.colind[g.rowptr[(size_t)4 * n + 3]];
        for (int q = 0; q < deg; ++q)
            for (int f = 0; f < 4; ++f) ci[4 * q + f] = 4 * g.pair_col[p0 + q] + f;
        for (int r = 0; r < x; ++r) ci[4 * deg + r] = 4 * cross[r] + 3;
    }
    return KNP_OK;
}

int knp_host_threads() {
    static const int n = [] {
        if (const char* e = getenv("KNP_HOST_THREADS")) { const int v = atoi(e); if (v > 0) return v; }
        int cpus = 0;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
        if (cpus <= 0) cpus = 1;
        double quota = 0.0;      // cgroup v2: "<quota> <period>" or "max <period>"; v1: two files
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[64] = {0};
            long long period = 0;
            if (fscanf(f, "%63s %lld", q, &period) == 2 && period > 0 && strcmp(q, "max") != 0) quota = atof(q) / (double)period;
            fclose(f);
        } else {
            long long q = -1, per = 0;
            if (FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(fq, "%lld", &q) != 1) q = -1; fclose(fq); }
            if (FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(fp, "%lld", &per) != 1) per = 0; fclose(fp); }
            if (q > 0 && per > 0) quota = (double)q / (double)per;
        }
        if (quota >= 1.0) cpus = std::min(cpus, (int)(quota + 0.5));
        // one process per GPU: the ranks of a node share that budget (torchrun / bench.py export LOCAL_WORLD_SIZE)
        if (const char* lws = getenv("LOCAL_WORLD_SIZE")) { const int k = atoi(lws); if (k > 1) cpus = std::max(1, cpus / k); }
        return std::max(1, std::min(cpus, quota >= 1.0 ? 32 : 16));
    }();
    return n;
}
