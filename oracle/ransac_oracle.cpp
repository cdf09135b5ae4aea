// ransac_oracle.cpp -- TEST INFRASTRUCTURE: CPU restatement of ImageGroup::RANSAC and
// ImageGroup::RANSACBatch (/root/reference/registration/imageGroup.cxx:629-804), the stage
// that replaces the linear iterations when fixed images are present (run(), :40-49).
// Not part of the product: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use it.
//
// PARITY UNPINNED.  The reference gets each candidate from vtkLandmarkTransform (similarity
// mode) -- VTK is a third-party dependency absent from /root/reference, version unpinned
// (CMakeLists.txt:15; CI matrix 9.0 .. latest).  What is restated here is the published
// algorithm VTK implements: Horn's closed-form absolute orientation with unit quaternions
// (JOSA A 4(4), 1987) -- centroids, correlation matrix M = sum a' b'^T, the symmetric 4x4 matrix
// N(M), its dominant eigenvector (vtkMath::JacobiN: the classic threshold Jacobi of Numerical
// Recipes 11.1, eigenvalues sorted in decreasing order), scale = sqrt(sum|b'|^2 / sum|a'|^2),
// translation = target centroid - s R source centroid.  The reference's tests hold no vector
// for this path.  The fit is unique where it is defined, so two correct solvers agree to
// rounding; the census (float compare against 2500) and the draws (std::mt19937(batch*1000),
// `rng() % nPoints`, `rng() % links.size()`) follow the reference line by line.
//
// One documented deviation: upstream runs the batches as OpenMP threads and takes the best of
// them in the order the threads finished (:640-664), so ties between batches resolve at random
// and the batch count is the machine's core count.  Here: batch order, explicit batch count.

#include "frog_oracle.h"

#include <cmath>
#include <cstring>
#include <random>
#include <vector>

namespace {

// vtkMath::JacobiN for n = 4 (Numerical Recipes `jacobi` + descending sort).
void jacobi_n4(double a[4][4], double w[4], double v[4][4])
{
    const int n = 4;
    double b[4], z[4];
    for (int ip = 0; ip < n; ip++) {
        for (int iq = 0; iq < n; iq++) v[ip][iq] = 0.0;
        v[ip][ip] = 1.0;
        b[ip] = w[ip] = a[ip][ip];
        z[ip] = 0.0;
    }
    for (int i = 0; i < 20; i++) {
        double sm = 0.0;
        for (int ip = 0; ip < n - 1; ip++)
            for (int iq = ip + 1; iq < n; iq++) sm += std::fabs(a[ip][iq]);
        if (sm == 0.0) break;
        const double tresh = i < 3 ? 0.2 * sm / (n * n) : 0.0;
        for (int ip = 0; ip < n - 1; ip++) {
            for (int iq = ip + 1; iq < n; iq++) {
                const double g = 100.0 * std::fabs(a[ip][iq]);
                if (i > 3 && (std::fabs(w[ip]) + g) == std::fabs(w[ip]) && (std::fabs(w[iq]) + g) == std::fabs(w[iq])) {
                    a[ip][iq] = 0.0;
                } else if (std::fabs(a[ip][iq]) > tresh) {
                    double h = w[iq] - w[ip], t;
                    if ((std::fabs(h) + g) == std::fabs(h)) {
                        t = a[ip][iq] / h;
                    } else {
                        const double theta = 0.5 * h / a[ip][iq];
                        t = 1.0 / (std::fabs(theta) + std::sqrt(1.0 + theta * theta));
                        if (theta < 0.0) t = -t;
                    }
                    const double c = 1.0 / std::sqrt(1 + t * t), s = t * c, tau = s / (1.0 + c);
                    h = t * a[ip][iq];
                    z[ip] -= h; z[iq] += h; w[ip] -= h; w[iq] += h;
                    a[ip][iq] = 0.0;
                    auto rot = [&](double m[4][4], int i1, int j1, int i2, int j2) {
                        const double gg = m[i1][j1], hh = m[i2][j2];
                        m[i1][j1] = gg - s * (hh + gg * tau);
                        m[i2][j2] = hh + s * (gg - hh * tau);
                    };
                    for (int j = 0; j <= ip - 1; j++) rot(a, j, ip, j, iq);
                    for (int j = ip + 1; j <= iq - 1; j++) rot(a, ip, j, j, iq);
                    for (int j = iq + 1; j < n; j++) rot(a, ip, j, iq, j);
                    for (int j = 0; j < n; j++) rot(v, j, ip, j, iq);
                }
            }
        }
        for (int ip = 0; ip < n; ip++) { b[ip] += z[ip]; w[ip] = b[ip]; z[ip] = 0.0; }
    }
    for (int j = 0; j < n - 1; j++) {               // eigenvalues in decreasing order, vectors follow
        int k = j;
        double tmp = w[k];
        for (int i = j + 1; i < n; i++) if (w[i] >= tmp) { k = i; tmp = w[k]; }
        if (k != j) {
            w[k] = w[j]; w[j] = tmp;
            for (int i = 0; i < n; i++) { const double t = v[i][j]; v[i][j] = v[i][k]; v[i][k] = t; }
        }
    }
}

struct Correspondences {
    std::vector<const float *> source, target;
    void reset() { source.clear(); target.clear(); }
    void add(const float *s, const float *t) { source.push_back(s); target.push_back(t); }
};

// vtkLandmarkTransform::InternalUpdate, similarity mode.  Returns false where VTK takes its
// collinear-points branch (not restated: such a candidate is skipped on both sides).
bool landmark_similarity(const Correspondences &c, double matrix[4][4])
{
    const size_t N = c.source.size();
    for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) matrix[i][j] = i == j;
    if (N == 0) return true;
    double sc[3] = { 0, 0, 0 }, tc[3] = { 0, 0, 0 };
    for (size_t i = 0; i < N; i++)
        for (int k = 0; k < 3; k++) { sc[k] += c.source[i][k]; tc[k] += c.target[i][k]; }
    for (int k = 0; k < 3; k++) { sc[k] /= N; tc[k] /= N; }
    if (N == 1) { for (int k = 0; k < 3; k++) matrix[k][3] = tc[k] - sc[k]; return true; }
    double M[3][3] = { { 0 } }, sa = 0, sb = 0;
    for (size_t pt = 0; pt < N; pt++) {
        double a[3], b[3];
        for (int k = 0; k < 3; k++) { a[k] = c.source[pt][k] - sc[k]; b[k] = c.target[pt][k] - tc[k]; }
        for (int i = 0; i < 3; i++) { M[i][0] += a[i] * b[0]; M[i][1] += a[i] * b[1]; M[i][2] += a[i] * b[2]; }
        sa += a[0] * a[0] + a[1] * a[1] + a[2] * a[2];
        sb += b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    }
    const double scale = std::sqrt(sb / sa);
    double Nm[4][4], ev[4][4], ew[4];
    Nm[0][0] = M[0][0] + M[1][1] + M[2][2];
    Nm[1][1] = M[0][0] - M[1][1] - M[2][2];
    Nm[2][2] = -M[0][0] + M[1][1] - M[2][2];
    Nm[3][3] = -M[0][0] - M[1][1] + M[2][2];
    Nm[0][1] = Nm[1][0] = M[1][2] - M[2][1];
    Nm[0][2] = Nm[2][0] = M[2][0] - M[0][2];
    Nm[0][3] = Nm[3][0] = M[0][1] - M[1][0];
    Nm[1][2] = Nm[2][1] = M[0][1] + M[1][0];
    Nm[1][3] = Nm[3][1] = M[2][0] + M[0][2];
    Nm[2][3] = Nm[3][2] = M[1][2] + M[2][1];
    jacobi_n4(Nm, ew, ev);
    if (ew[0] == ew[1] || N == 2) return false;
    const double w = ev[0][0], x = ev[1][0], y = ev[2][0], z = ev[3][0];
    const double ww = w * w, wx = w * x, wy = w * y, wz = w * z, xx = x * x, yy = y * y, zz = z * z, xy = x * y, xz = x * z, yz = y * z;
    matrix[0][0] = ww + xx - yy - zz; matrix[1][0] = 2.0 * (wz + xy); matrix[2][0] = 2.0 * (-wy + xz);
    matrix[0][1] = 2.0 * (-wz + xy); matrix[1][1] = ww - xx + yy - zz; matrix[2][1] = 2.0 * (wx + yz);
    matrix[0][2] = 2.0 * (wy + xz); matrix[1][2] = 2.0 * (-wx + yz); matrix[2][2] = ww - xx - yy + zz;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) matrix[i][j] *= scale;
    for (int i = 0; i < 3; i++)
        matrix[i][3] = tc[i] - (matrix[i][0] * sc[0] + matrix[i][1] * sc[1] + matrix[i][2] * sc[2]);
    return true;
}

// vtkLinearTransform::TransformPoint(const float in[3], float out[3]) with a double matrix
inline void linear_point(const double m[4][4], const float in[3], float out[3])
{
    const float x = (float)(m[0][0] * in[0] + m[0][1] * in[1] + m[0][2] * in[2] + m[0][3]);
    const float y = (float)(m[1][0] * in[0] + m[1][1] * in[1] + m[1][2] * in[2] + m[1][3]);
    const float z = (float)(m[2][0] * in[0] + m[2][1] * in[1] + m[2][2] * in[2] + m[2][3]);
    out[0] = x; out[1] = y; out[2] = z;
}

inline float distance2(const float a[3], const float b[3])
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}

inline double determinant3(const double m[4][4])
{
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0])
         + m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

} // namespace

extern "C" long frogo_ransac(frogo_group *g, uint32_t image, int iterations, int batches,
                             float inlier_distance, float max_scale)
{
    const float *xyz = frogo_xyz_ptr(g), *xyz2 = frogo_xyz2_ptr(g);
    const uint32_t *poff = frogo_point_offset_ptr(g);
    const uint64_t *rowp = frogo_row_ptr(g);
    const uint16_t *limg = frogo_link_image_ptr(g);
    const uint32_t *lpt = frogo_link_point_ptr(g);
    const uint32_t pb = poff[image];
    const int nPoints = (int)(poff[image + 1] - pb);
    const int batchIterations = iterations / batches;                     // :636
    const float maxDistance2 = std::pow(inlier_distance, 2);              // :677, :730
    auto position = [&](const float *table, uint16_t img, uint32_t pt) { return table + 3 * ((size_t)poff[img] + pt); };

    long maxNumberOfInliers = 0;
    double best[4][4];
    frogo_get_matrix(g, image, &best[0][0]);
    std::vector<long> batchMaxima(batches, 0);
    std::vector<double> batchMatrices((size_t)batches * 16, 0.0);
    #pragma omp parallel for schedule(dynamic, 1)                         // :639-647, one batch per thread
    for (int batch = 0; batch < batches; batch++) {                       // RANSACBatch, :718-804
        std::mt19937 rng(batch * 1000);
        long batchMax = 0;
        double batchMatrix[4][4];
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) batchMatrix[i][j] = i == j;
        Correspondences c;
        for (int i = 0; i < batchIterations; i++) {
            c.reset();
            for (int j = 0; j < 4; j++) {
                while (true) {
                    int pt = rng() % nPoints;
                    const uint64_t first = rowp[pb + pt], size = rowp[pb + pt + 1] - first;
                    if (size == 0) continue;
                    int linkId = rng() % size;
                    c.add(position(xyz, image, pt), position(xyz, limg[first + linkId], lpt[first + linkId]));
                    break;
                }
            }
            double matrix2[4][4];
            const bool defined = landmark_similarity(c, matrix2);
            long nInliers = 0;
            float transformed[3];
            for (int pt = 0; pt < nPoints; pt++) {
                linear_point(matrix2, position(xyz, image, pt), transformed);
                for (uint64_t l = rowp[pb + pt]; l < rowp[pb + pt + 1]; l++)
                    if (distance2(transformed, position(xyz2, limg[l], lpt[l])) < maxDistance2) nInliers++;
            }
            float determinant = std::fabs(determinant3(matrix2));
            if (!defined) continue;
            if ((determinant > max_scale) || (determinant < 1.0 / max_scale)) continue;
            if (batchMax < nInliers) { batchMax = nInliers; std::memcpy(batchMatrix, matrix2, sizeof batchMatrix); }
        }
        batchMaxima[batch] = batchMax;
        std::memcpy(&batchMatrices[(size_t)batch * 16], batchMatrix, sizeof batchMatrix);
    }
    for (int batch = 0; batch < batches; batch++)                         // :651-664, in batch order (see header)
        if (batchMaxima[batch] > maxNumberOfInliers) {
            maxNumberOfInliers = batchMaxima[batch];
            std::memcpy(best, &batchMatrices[(size_t)batch * 16], sizeof best);
        }
    // refit on the inlier half-links of the best candidate, :666-700
    Correspondences all;
    float transformed[3];
    for (int pt = 0; pt < nPoints; pt++) {
        linear_point(best, position(xyz, image, pt), transformed);
        for (uint64_t l = rowp[pb + pt]; l < rowp[pb + pt + 1]; l++) {
            const float *pB = position(xyz2, limg[l], lpt[l]);
            if (distance2(transformed, pB) < maxDistance2) all.add(position(xyz, image, pt), pB);
        }
    }
    double fit[4][4];
    if (!landmark_similarity(all, fit)) std::memcpy(fit, best, sizeof fit);
    frogo_set_matrix(g, image, &fit[0][0]);
    return maxNumberOfInliers;
}
