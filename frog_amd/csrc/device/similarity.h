// similarity.h -- closed-form least-squares similarity transform between two point lists
// (rotation + isotropic scale + translation), host side, f64.
//
// This is what the reference gets from vtkLandmarkTransform in SetModeToSimilarity()
// (imageGroup.cxx:669-700, :724-758).  VTK is a third-party dependency that is absent from
// /root/reference; the published algorithm is Horn, "Closed-form solution of absolute
// orientation using unit quaternions", JOSA A 4(4), 1987: centroids, the 3x3 correlation
// matrix of the centred points, the symmetric 4x4 matrix whose dominant eigenvector is the
// rotation quaternion, scale = sqrt(sum |b'|^2 / sum |a'|^2), translation from the centroids.
// The eigenvector comes from a cyclic Jacobi iteration (the family vtkMath::JacobiN belongs to).
// Parity with VTK's bits is unpinned; the fit is unique, so any correct solver agrees to rounding.
#pragma once

#include <cmath>
#include <cstddef>

namespace frog {

// Eigen-decomposition of a symmetric 4x4 matrix by cyclic Jacobi rotations.
// On return `a` is (nearly) diagonal and the columns of `v` are the eigenvectors.
inline void jacobi4(double a[4][4], double v[4][4])
{
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) v[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = 0;
        for (int p = 0; p < 4; p++)
            for (int q = p + 1; q < 4; q++) off += a[p][q] * a[p][q];
        if (off == 0.0) break;
        for (int p = 0; p < 4; p++)
            for (int q = p + 1; q < 4; q++) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 4; k++) {                    // A <- A J
                    const double akp = a[k][p], akq = a[k][q];
                    a[k][p] = c * akp - s * akq;
                    a[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 4; k++) {                    // A <- J^T A
                    const double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = c * apk - s * aqk;
                    a[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 4; k++) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

// Fits target ~ s R source + t.  `point(i, a, b)` yields the i-th correspondence.
// Returns false (and the identity) when the rotation is not determined (fewer than 3 points,
// collinear points: the two largest eigenvalues coincide).  out: row-major 4x4.
template <class Points> bool similarity_fit(size_t n, Points point, double out[16])
{
    for (int k = 0; k < 16; k++) out[k] = (k % 5 == 0) ? 1.0 : 0.0;
    if (n == 0) return true;                                     // vtkLandmarkTransform: identity for no points
    double ca[3] = { 0, 0, 0 }, cb[3] = { 0, 0, 0 }, a[3], b[3];
    for (size_t i = 0; i < n; i++) {
        point(i, a, b);
        for (int k = 0; k < 3; k++) { ca[k] += a[k]; cb[k] += b[k]; }
    }
    for (int k = 0; k < 3; k++) { ca[k] /= (double)n; cb[k] /= (double)n; }
    if (n == 1) {
        for (int k = 0; k < 3; k++) out[4 * k + 3] = cb[k] - ca[k];
        return true;
    }
    double M[3][3] = { { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 } }, sa = 0, sb = 0;
    for (size_t i = 0; i < n; i++) {
        point(i, a, b);
        for (int k = 0; k < 3; k++) { a[k] -= ca[k]; b[k] -= cb[k]; }
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) M[r][c] += a[r] * b[c];
        sa += a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
        sb += b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    }
    const double scale = std::sqrt(sb / sa);
    double N[4][4], V[4][4];
    N[0][0] = M[0][0] + M[1][1] + M[2][2];
    N[1][1] = M[0][0] - M[1][1] - M[2][2];
    N[2][2] = -M[0][0] + M[1][1] - M[2][2];
    N[3][3] = -M[0][0] - M[1][1] + M[2][2];
    N[0][1] = N[1][0] = M[1][2] - M[2][1];
    N[0][2] = N[2][0] = M[2][0] - M[0][2];
    N[0][3] = N[3][0] = M[0][1] - M[1][0];
    N[1][2] = N[2][1] = M[0][1] + M[1][0];
    N[1][3] = N[3][1] = M[2][0] + M[0][2];
    N[2][3] = N[3][2] = M[1][2] + M[2][1];
    jacobi4(N, V);
    int best = 0, second = -1;
    for (int k = 1; k < 4; k++) if (N[k][k] > N[best][best]) best = k;
    for (int k = 0; k < 4; k++) if (k != best && (second < 0 || N[k][k] > N[second][second])) second = k;
    if (n == 2 || N[best][best] == N[second][second]) return false;
    const double w = V[0][best], x = V[1][best], y = V[2][best], z = V[3][best];
    const double ww = w * w, wx = w * x, wy = w * y, wz = w * z, xx = x * x, yy = y * y, zz = z * z,
                 xy = x * y, xz = x * z, yz = y * z;
    double R[3][3];
    R[0][0] = ww + xx - yy - zz; R[1][0] = 2.0 * (wz + xy);    R[2][0] = 2.0 * (-wy + xz);
    R[0][1] = 2.0 * (-wz + xy);  R[1][1] = ww - xx + yy - zz;  R[2][1] = 2.0 * (wx + yz);
    R[0][2] = 2.0 * (wy + xz);   R[1][2] = 2.0 * (-wx + yz);   R[2][2] = ww - xx - yy + zz;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) out[4 * r + c] = R[r][c] * scale;
        out[4 * r + 3] = cb[r] - (out[4 * r] * ca[0] + out[4 * r + 1] * ca[1] + out[4 * r + 2] * ca[2]);
    }
    return true;
}

inline double det3_of_4x4(const double m[16])
{
    // vtkMatrix4x4::Determinant of an affine matrix (last row 0 0 0 1) = determinant of the 3x3 block
    return m[0] * (m[5] * m[10] - m[6] * m[9]) - m[1] * (m[4] * m[10] - m[6] * m[8]) + m[2] * (m[4] * m[9] - m[5] * m[8]);
}

} // namespace frog
