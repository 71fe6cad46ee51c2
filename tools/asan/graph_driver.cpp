#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
#include "knp_internal.hpp"
int main(int argc, char** argv) {
    for (int a = 1; a < argc; ++a) {
        FILE* f = fopen(argv[a], "rb"); int32_t h[5]; if (fread(h, 4, 5, f) != 5) return 2;
        const int dim = h[0], nv = h[1], nc = h[2], ng = h[3], nq = h[4];
        std::vector<int32_t> cells((size_t)nc * (dim + 1)), gamma((size_t)ng * 4), gprog(ng);
        std::vector<double> coords((size_t)nv * dim), qp((size_t)nq * dim), qw(nq);
        std::vector<uint8_t> side(nc);
        size_t r = 0;
        r += fread(cells.data(), 4, cells.size(), f); r += fread(coords.data(), 8, coords.size(), f); r += fread(side.data(), 1, nc, f);
        r += fread(gamma.data(), 4, gamma.size(), f); r += fread(gprog.data(), 4, ng, f); r += fread(qp.data(), 8, qp.size(), f); r += fread(qw.data(), 8, nq, f);
        fclose(f);
        knp_mesh_desc m{}; m.dim = dim; m.n_vertices = nv; m.n_vertices_owned = nv; m.n_cells = nc; m.n_cells_owned = nc;
        m.cells = cells.data(); m.coords = coords.data(); m.cell_side = side.data(); m.n_gamma = ng; m.gamma = gamma.data(); m.gamma_prog = gprog.data();
        m.n_q = nq; m.q_pts = qp.data(); m.q_w = qw.data();
        KnpHostGraph g; int rc = knp_build_graph(&m, g);
        printf("%s rc=%d nodes=%d pairs=%zu contrib=%zu nnz=%d err=%s\n", argv[a], rc, g.n_nodes, g.pair_col.size(), g.contrib_cell.size(), g.rowptr.empty() ? -1 : g.rowptr.back(), g.error.c_str());
        // partial ownership variant: half the vertices/cells owned (ghost layer semantics are the caller's business; sizes only)
    }
    return 0;
}
