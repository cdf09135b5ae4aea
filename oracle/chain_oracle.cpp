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

#include <algorithm>
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

// 3x3 linear solve by Gaussian elimination with partial pivoting (the role of vtkMath::LinearSolve3x3)
void solve3x3(const double A[3][3], const double b[3], double x[3])
{
    double M[3][4];
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) M[r][c] = A[r][c]; M[r][3] = b[r]; }
    for (int col = 0; col < 3; col++) {
        int piv = col;
        for (int r = col + 1; r < 3; r++) if (std::fabs(M[r][col]) > std::fabs(M[piv][col])) piv = r;
        if (piv != col) for (int c = 0; c < 4; c++) std::swap(M[piv][c], M[col][c]);
        for (int r = col + 1; r < 3; r++) {
            const double f = M[r][col] / M[col][col];
            for (int c = col; c < 4; c++) M[r][c] -= f * M[col][c];
        }
    }
    for (int r = 2; r >= 0; r--) {
        double v = M[r][3];
        for (int c = r + 1; c < 3; c++) v -= M[r][c] * x[c];
        x[r] = v / M[r][r];
    }
}

// vtkWarpTransform's inverse (the scheme vtkBSplineTransform inherits): Newton iterations on
// T(x) = p from the first guess x = p - (T(p) - p); when |T(x) - p|^2 grows, the last step is
// shortened by the factor a parabola through the last two values suggests, clamped to [0.1, 0.5];
// stops when step and residual are both under InverseTolerance (0.001) or after InverseIterations
// (500), then keeps the last good point.  VTK absent: restated from its documented behaviour.
void bspline_inverse(const frog_chain_link &t, const double point[3], double inverse[3], double Jinv[3][3])
{
    frog_chain_link fwd = t;
    fwd.type = FROG_T_BSPLINE;
    const double toleranceSquared = 1e-3 * 1e-3;
    double deltaP[3], deltaI[3] = { 0, 0, 0 }, derivative[3][3], lastInverse[3];
    double functionValue = 0, functionDerivative = 0, lastFunctionValue = 1e300, f = 1.0;
    link_apply(fwd, point, inverse, derivative);
    for (int k = 0; k < 3; k++) { inverse[k] = point[k] - (inverse[k] - point[k]); lastInverse[k] = inverse[k]; }
    int i;
    const int n = 500;
    for (i = 0; i < n; i++) {
        link_apply(fwd, inverse, deltaP, derivative);
        for (int k = 0; k < 3; k++) deltaP[k] -= point[k];
        functionValue = deltaP[0] * deltaP[0] + deltaP[1] * deltaP[1] + deltaP[2] * deltaP[2];
        if (i == 0 || functionValue < lastFunctionValue) {
            solve3x3(derivative, deltaP, deltaI);
            const double errorSquared = deltaI[0] * deltaI[0] + deltaI[1] * deltaI[1] + deltaI[2] * deltaI[2];
            if (errorSquared < toleranceSquared && functionValue < toleranceSquared) break;
            for (int k = 0; k < 3; k++) lastInverse[k] = inverse[k];
            lastFunctionValue = functionValue;
            functionDerivative = 0;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) functionDerivative -= 2 * deltaP[r] * derivative[r][c] * deltaI[c];
            for (int k = 0; k < 3; k++) inverse[k] -= deltaI[k];
            f = 1.0;
            continue;
        }
        double a = -functionDerivative / (2 * (functionValue - lastFunctionValue - functionDerivative));
        if (a < 0.1) a = 0.1;
        if (a > 0.5) a = 0.5;
        f *= a;
        for (int k = 0; k < 3; k++) inverse[k] = lastInverse[k] - f * deltaI[k];
    }
    if (i >= n) for (int k = 0; k < 3; k++) inverse[k] = lastInverse[k];
    double out[3];
    link_apply(fwd, inverse, out, derivative);
    for (int c = 0; c < 3; c++) {
        const double e[3] = { c == 0 ? 1.0 : 0.0, c == 1 ? 1.0 : 0.0, c == 2 ? 1.0 : 0.0 };
        double col[3];
        solve3x3(derivative, e, col);
        for (int r = 0; r < 3; r++) Jinv[r][c] = col[r];
    }
}

void chain_apply(const frog_chain_link *links, uint32_t n, const double x[3], double y[3], double J[3][3])
{
    double p[3] = { x[0], x[1], x[2] };
    double A[3][3] = { { 1, 0, 0 }, { 0, 1, 0 }, { 0, 0, 1 } };
    for (uint32_t l = 0; l < n; l++) {
        double q[3], Jl[3][3], B[3][3];
        if (links[l].type == FROG_T_BSPLINE_INVERSE) bspline_inverse(links[l], p, q, Jl);
        else link_apply(links[l], p, q, Jl);
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


// vtkImageReslice as tools/VolumeTransform.cxx:119-136 sets it up: every voxel of the output grid goes
// through the chain into the source's frame; nearest or trilinear sampling; default border (half a voxel
// beyond the outermost voxel centres still reads the edge), background elsewhere.  `src` and `out`
// are doubles here (the product converts to and from the file's scalar type around the same arithmetic).
void frogo_chain_reslice(const frog_chain_link *links, uint32_t n_links, const double *src, const uint32_t sdims[3],
                         const double sorigin[3], const double sspacing[3], const uint32_t odims[3], const double oorigin[3],
                         const double ospacing[3], int interpolation, double background, double *out)
{
    const long sx = sdims[0], sy = sdims[1], sz = sdims[2];
    auto voxel = [&](long x, long y, long z) {
        x = std::min(std::max(x, 0L), sx - 1); y = std::min(std::max(y, 0L), sy - 1); z = std::min(std::max(z, 0L), sz - 1);
        return src[x + sx * (y + sy * z)];
    };
    #pragma omp parallel for
    for (long k = 0; k < (long)odims[2]; k++)
        for (uint32_t j = 0; j < odims[1]; j++)
            for (uint32_t i = 0; i < odims[0]; i++) {
                const double in[3] = { oorigin[0] + i * ospacing[0], oorigin[1] + j * ospacing[1], oorigin[2] + k * ospacing[2] };
                double p[3], J[3][3];
                chain_apply(links, n_links, in, p, J);
                double c[3];
                bool inside = true;
                for (int a = 0; a < 3; a++) {
                    c[a] = (p[a] - sorigin[a]) / sspacing[a];
                    inside = inside && c[a] >= -0.5 && c[a] <= (double)sdims[a] - 0.5;
                }
                double v = background;
                if (inside && !interpolation) {
                    v = voxel((long)std::floor(c[0] + 0.5), (long)std::floor(c[1] + 0.5), (long)std::floor(c[2] + 0.5));
                } else if (inside) {
                    const long x0 = (long)std::floor(c[0]), y0 = (long)std::floor(c[1]), z0 = (long)std::floor(c[2]);
                    const double fx = c[0] - x0, fy = c[1] - y0, fz = c[2] - z0;
                    v = 0;
                    for (int dz = 0; dz < 2; dz++)
                        for (int dy = 0; dy < 2; dy++)
                            for (int dx = 0; dx < 2; dx++)
                                v += (dx ? fx : 1 - fx) * (dy ? fy : 1 - fy) * (dz ? fz : 1 - fz) * voxel(x0 + dx, y0 + dy, z0 + dz);
                }
                out[i + (size_t)odims[0] * (j + (size_t)odims[1] * k)] = v;
            }
}

}
