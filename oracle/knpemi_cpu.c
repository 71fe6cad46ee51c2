/* knpemi_cpu.c -- TEST INFRASTRUCTURE / CPU BASELINE ONLY (never linked into, loaded by or called from the product).
 *
 * C/OpenMP twin of the three kernels that carry a KNP-EMI timestep on the host, so that the CPU baseline beside the GPU
 * number can be quoted at 1 core and at all cores of the box:
 *   knp_cpu_assemble_volume   element-local P1 block assembly, cell by cell, scatter-added into CSR -- what
 *                             multiphenicsx.fem.petsc.assemble_matrix_block does behind KNPEMIx_solver.py:110-115
 *                             with the forms of KNPEMIx_problem.py:586-591,598,603,633-634
 *   knp_cpu_spmv              CSR SpMV = MatMult inside KSPSolve (KNPEMIx_solver.py:435) and inside every V-cycle level
 *   knp_cpu_dense_matvec      the coarsest-level solve of the V-cycle
 * The driver around them is oracle/knpemi_cpu_twin.py (same GMRES / V-cycle restatement as the NumPy oracle).
 * Built by __graft_entry__.build():  gcc -O3 -march=x86-64-v3 -fopenmp -fPIC -shared oracle/knpemi_cpu.c -o oracle/libknpemi_cpu.so
 */
#include <omp.h>
#include <stdint.h>
#include <string.h>

int knp_cpu_max_threads(void) { return omp_get_num_procs(); }
void knp_cpu_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }

void knp_cpu_spmv(int n, const int32_t* rp, const int32_t* ci, const double* v, const double* x, double* y) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = rp[i]; k < rp[i + 1]; ++k) s += v[k] * x[ci[k]];
        y[i] = s;
    }
}

void knp_cpu_dense_matvec(int n, const double* M, const double* x, double* y) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        const double* m = M + (size_t)i * n;
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += m[k] * x[k];
        y[i] = s;
    }
}

/* Volume blocks of A.  For every cell c and local pair (a,b), slots[((c*nl + a)*nl + b)*10 + t] is the CSR position of
 *   t = j     : A[(ra,j),(cb,j)]     += M_ab + dt D_j K_ab
 *   t = 3 + j : A[(ra,j),(cb,phi)]   += dt D_j z_j / psi * cbar_j(c) * K_ab
 *   t = 6 + j : A[(ra,phi),(cb,j)]   += dt z_j D_j K_ab
 *   t = 9     : A[(ra,phi),(cb,phi)] += sum_j dt D_j z_j^2 / psi * cbar_j(c) * K_ab
 * cbar_j(c) = mean of the previous concentration j over the cell's vertices (exact for P1 x constant gradients). */
void knp_cpu_assemble_volume(int nc, int nl, const int32_t* slots, const double* Mloc, const double* Kloc, const double* cbar,
                             double dt, double psi, const double* D, const double* z, int64_t nnz, double* vals) {
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nnz; ++k) vals[k] = 0.0;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < nc; ++c) {
        const double cb0 = cbar[c], cb1 = cbar[(size_t)nc + c], cb2 = cbar[(size_t)2 * nc + c];
        const double cb[3] = {cb0, cb1, cb2};
        for (int ab = 0; ab < nl * nl; ++ab) {
            const double M = Mloc[(size_t)c * nl * nl + ab], K = Kloc[(size_t)c * nl * nl + ab];
            const int32_t* s = slots + ((size_t)c * nl * nl + ab) * 10;
            double pp = 0.0;
            for (int j = 0; j < 3; ++j) {
                const double kk = M + dt * D[j] * K;
                const double kp = dt * D[j] * z[j] / psi * cb[j] * K;
                const double pk = dt * z[j] * D[j] * K;
                pp += dt * D[j] * z[j] * z[j] / psi * cb[j] * K;
#pragma omp atomic
                vals[s[j]] += kk;
#pragma omp atomic
                vals[s[3 + j]] += kp;
#pragma omp atomic
                vals[s[6 + j]] += pk;
            }
#pragma omp atomic
            vals[s[9]] += pp;
        }
    }
}

/* vals[slot[k]] += add[k] (membrane terms, computed by the driver) */
void knp_cpu_scatter_add(int64_t n, const int32_t* slot, const double* add, double* vals) {
    for (int64_t k = 0; k < n; ++k) vals[slot[k]] += add[k];
}
