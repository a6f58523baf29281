// Host build of centerpose_amd/csrc/pnp_linalg.h for tests/test_pnp_linalg_cpu.py (the same source the device compiles).
#include "../../centerpose_amd/csrc/pnp_linalg.h"

extern "C" {
// packed lower triangle (n (n + 1) / 2 doubles, destroyed) -> unit eigenvector of the smallest eigenvalue
void pnp_host_smallest_eigvec12(double* A, double* out) { smallest_eigvec<12>(A, out); }
void pnp_host_smallest_eigvec9(double* A, double* out) { smallest_eigvec<9>(A, out); }
// rotation vector -> R (row-major 9) and dR/dr (3 x 9, row-major)
void pnp_host_rodrigues(const double* r, double* R, double* J) {
    double Rm[9], Jm[27];
    rodrigues(r, Rm, Jm);
    for (int i = 0; i < 9; ++i) R[i] = Rm[i];
    for (int i = 0; i < 27; ++i) J[i] = Jm[i];
}
void pnp_host_rodrigues_nojac(const double* r, double* R) {
    double Rm[9];
    rodrigues(r, Rm, nullptr);
    for (int i = 0; i < 9; ++i) R[i] = Rm[i];
}
void pnp_host_polar3(const double* A, double* R) { polar3(A, R); }
void pnp_host_rot_to_rvec(const double* R, double* r) { rot_to_rvec(R, r); }
// row-major 6 x 6 (destroyed), b (destroyed) -> x
void pnp_host_solve6(double* A, double* b, double* x) {
    double Am[36], bm[6], xm[6];
    for (int i = 0; i < 36; ++i) Am[i] = A[i];
    for (int i = 0; i < 6; ++i) bm[i] = b[i];
    solve6(Am, bm, xm);
    for (int i = 0; i < 6; ++i) x[i] = xm[i];
}
// symmetric n x n row-major (destroyed: eigenvalues on the diagonal), V: eigenvectors as columns; floor_exit = the product's form
void pnp_host_jacobi_eig(double* A, int n, double* V, int floor_exit) {
    if (floor_exit) jacobi_eig<true>(A, n, V);
    else jacobi_eig<false>(A, n, V);
}
}
