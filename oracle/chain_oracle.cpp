// chain_oracle.cpp -- CPU restatement of forward evaluation and Jacobian of a FROG transform
// chain.  TEST INFRASTRUCTURE (see frog_oracle.h): never linked into the product.
//
// PARITY UNPINNED: the reference delegates this to VTK (vtkGeneralTransform /
// vtkMatrixToLinearTransform / vtkBSplineTransform: tools/PointsTransform.cxx:70-80,
// tools/CheckDiffeomorphism.cxx:67-85), which is absent from this image.  The restatement
// follows VTK's documented semantics (double path): see include/frog_chain.h.  Checked by
// closed-form cases in tests/test_chain.py (identity, pure matrix, a lattice whose displacement
// is an exact quadratic/linear function of position).
#include "../include/frog_chain.h"

#include <cmath>
#include <cstring>
#include <omp.h>

namespace {

// uniform cubic B-spline basis at fraction f (registration/imageGroup.cxx:221-232) and its derivative
inline void basis(double f, double F[4], double G[4])
{
    F[3] = f * f * f / 6;
    F[0] = (f * f - f) / 2 - F[3] + 1.0 / 6;
    F[2] = f + F[0] - F[3] * 2;
    F[1] = 1 - F[0] - F[2] - F[3];
    G[0] = -(1 - f) * (1 - f) / 2;
    G[1] = 1.5 * f * f - 2 * f;
    G[2] = -1.5 * f * f + f + 0.5;
    G[3] = f * f / 2;
}

// y = T(x) and J = dT/dx for one link
void link_apply(const frog_chain_link &t, const double x[3], double y[3], double J[3][3])
{
    if (t.type == FROG_T_LINEAR) {
        for (int r = 0; r < 3; r++) {
            y[r] = t.matrix[4 * r] * x[0] + t.matrix[4 * r + 1] * x[1] + t.matrix[4 * r + 2] * x[2] + t.matrix[4 * r + 3];
            for (int c = 0; c < 3; c++) J[r][c] = t.matrix[4 * r + c];
        }
        return;
    }
    double F[3][4], G[3][4];
    int i0[3];
    for (int k = 0; k < 3; k++) {
        const double p = (x[k] - t.origin[k]) / t.spacing[k];
        const double fl = std::floor(p);
        i0[k] = (int)fl - 1;
        basis(p - fl, F[k], G[k]);
    }
    double d[3] = { 0, 0, 0 }, dd[3][3] = { { 0 } };        // displacement and d(displacement_r)/d(u_c)
    const int dx = (int)t.dims[0], dy = (int)t.dims[1], dz = (int)t.dims[2];
    for (int k = 0; k < 4; k++) {
        const int z = i0[2] + k;
        if (z < 0 || z >= dz) continue;
        for (int j = 0; j < 4; j++) {
            const int yy = i0[1] + j;
            if (yy < 0 || yy >= dy) continue;
            for (int i = 0; i < 4; i++) {
                const int xx = i0[0] + i;
                if (xx < 0 || xx >= dx) continue;
                const float *c = t.coeffs + 3 * ((size_t)xx + (size_t)dx * ((size_t)yy + (size_t)dy * (size_t)z));
                const double w = F[0][i] * F[1][j] * F[2][k];
                const double wx = G[0][i] * F[1][j] * F[2][k], wy = F[0][i] * G[1][j] * F[2][k], wz = F[0][i] * F[1][j] * G[2][k];
                for (int r = 0; r < 3; r++) {
                    d[r] += w * c[r];
                    dd[r][0] += wx * c[r]; dd[r][1] += wy * c[r]; dd[r][2] += wz * c[r];
                }
            }
        }
    }
    for (int r = 0; r < 3; r++) {
        y[r] = x[r] + d[r];
        for (int c = 0; c < 3; c++) J[r][c] = (r == c ? 1.0 : 0.0) + dd[r][c] / t.spacing[c];
    }
}

void chain_apply(const frog_chain_link *links, uint32_t n, const double x[3], double y[3], double J[3][3])
{
    double p[3] = { x[0], x[1], x[2] };
    double A[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (uint32_t l = 0; l < n; l++) {
        double q[3], Jl[3][3], B[3][3];
        link_apply(links[l], p, q, Jl);
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) B[r][c] = Jl[r][0] * A[0][c] + Jl[r][1] * A[1][c] + Jl[r][2] * A[2][c];
        std::memcpy(A, B, sizeof A);
        std::memcpy(p, q, sizeof p);
    }
    std::memcpy(y, p, sizeof p);
    std::memcpy(J, A, sizeof A);
}

inline double det3(const double J[3][3])
{
    return J[0][0] * (J[1][1] * J[2][2] - J[1][2] * J[2][1]) - J[0][1] * (J[1][0] * J[2][2] - J[1][2] * J[2][0])
         + J[0][2] * (J[1][0] * J[2][1] - J[1][1] * J[2][0]);
}

} // namespace

extern "C" {

void frogo_chain_apply(const frog_chain_link *links, uint32_t n_links, const double *in, double *out, double *jac9, size_t n)
{
    #pragma omp parallel for
    for (long i = 0; i < (long)n; i++) {
        double J[3][3];
        chain_apply(links, n_links, in + 3 * i, out + 3 * i, J);
        if (jac9) std::memcpy(jac9 + 9 * i, J, sizeof J);
    }
}

// CheckDiffeomorphism.cxx:67-85
void frogo_chain_check(const frog_chain_link *links, uint32_t n_links, const double origin[3], const double spacing[3],
                       const uint32_t dims[3], uint64_t *n_negative, double *min_det)
{
    uint64_t neg = 0;
    double mn = INFINITY;
    #pragma omp parallel for reduction(+ : neg) reduction(min : mn)
    for (long k = 0; k < (long)dims[2]; k++)
        for (uint32_t j = 0; j < dims[1]; j++)
            for (uint32_t i = 0; i < dims[0]; i++) {
                const double in[3] = { origin[0] + i * spacing[0], origin[1] + j * spacing[1], origin[2] + k * spacing[2] };
                double out[3], J[3][3];
                chain_apply(links, n_links, in, out, J);
                const double d = det3(J);
                if (d < 0) neg++;
                mn = std::min(mn, d);
            }
    *n_negative = neg;
    *min_det = mn;
}

}
