// frog_oracle.cpp -- CPU oracle for the FROG groupwise-registration hot path.
//
// TEST INFRASTRUCTURE (see frog_oracle.h): checker for the HIP path and timed
// CPU baseline.  Not linked, imported or executed by anything under frog_amd/.
//
// Restates, over flat SoA/CSR arrays, the loops of
//   /root/reference/registration/imageGroup.cxx, stats.h, stats.cxx, image.cxx
// and the pieces of VTK (absent from the reference tree: vtkOpenSURF3D submodule
// empty, VTK >= 9.0 found by find_package, version unpinned, CMakeLists.txt:15)
// that those loops call:
//   vtkMath::Distance2BetweenPoints(float*,float*)   f32 sum of squared diffs
//   vtkLinearTransform point transform               f64 M*[x y z 1] -> f32
//   vtkBSplineTransform forward transform (cubic)    f64, separable 4x4x4
//   vtkBoundingBox AddPoint/ScaleAboutCenter/GetLength/GetBound
// Parity: Stats part pinned by oracle/_ref (the reference's own stats.cxx);
// solver loops and VTK arithmetic "parity unpinned" (nothing in the reference
// to pin them with, imageGroup.cxx unbuildable here).
//
// Build: g++ -O2 -std=c++17 -fopenmp -ffp-contract=off (no -ffast-math, no
// -march: the reference's x86-64 build has no FMA contraction).

#include "frog_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <random>
#include <vector>

#include <omp.h>

namespace {

// ---------------------------------------------------------------------------
// Stats (stats.h:18-101, stats.cxx:14-131)
// ---------------------------------------------------------------------------

// stats.h:10-16.  c and x2 are f32, the exponential and the final product are
// f64, the result is rounded to f32.
inline float chi_pdf(float x)
{
    float c = 0.797884560802865;
    float x2 = x * x;
    float cx2 = c * x2;                               // f32 * f32 (left-to-right)
    return (float)((double)cx2 * std::exp(-0.5 * (double)x2));
}

struct EmStats {
    int max_size = 10000;          // Stats::maxSize       stats.cxx:10
    int max_iterations = 10000;    // Stats::maxIterations stats.cxx:11
    float epsilon = 1e-6f;         // Stats::epsilon       stats.cxx:12

    std::vector<float> samples, weights;
    std::vector<uint32_t> ordinals;   // oracle-only bookkeeping: source ordinal of each sample
    uint32_t next_ordinal = 0;
    int size = 0;
    int virtual_size = 0;
    float c1 = 10, c2 = 300, ratio = 0.5f;     // stats.h:94
    std::mt19937 rng;                         // seeded once, never reseeded (stats.h:97)
    bool needs_random = false;
    std::vector<float> histogram;

    EmStats() { rng.seed(0); }

    // stats.h:36-50
    void add_slot()
    {
        virtual_size++;
        if (virtual_size > max_size) { needs_random = true; return; }
        samples.push_back(-1);
        weights.push_back(-1);
        ordinals.push_back(0);
    }

    // stats.h:52-56
    void reset() { size = 0; next_ordinal = 0; }

    // stats.h:58-76.  The generator is consulted only while the buffer is not full.
    void add_sample(float sample, float weight = 1)
    {
        uint32_t ord = next_ordinal++;
        if (!needs_random) {
            samples[size] = sample; weights[size] = weight; ordinals[size] = ord;
            size++;
            return;
        }
        if ((size_t)size == samples.size()) return;
        // (float) rng() / rng.max(): u32 draw -> f32, divided by (float)4294967295 = 2^32
        float random = (float)rng() / (float)rng.max();
        if (random > (float)samples.size() / (float)virtual_size) return;
        samples[size] = sample; weights[size] = weight; ordinals[size] = ord;
        size++;
    }

    // stats.h:84-92
    float inlier_probability(float d) const
    {
        const float eps = 1e-10;
        if ((double)d < 0.1) return 1;
        float a1 = c1 + eps;
        float x1 = ratio * chi_pdf(d / a1) / a1;                       // all f32
        float a2 = c2 + eps;
        float x2 = (float)((1.0 - (double)ratio) * (double)chi_pdf(d / a2) / (double)a2);
        return x1 / (x1 + x2 + eps);
    }

    // stats.cxx:14-70
    void estimate()
    {
        const float esp = 1.59576912160573;
        int iteration = 0;
        const float eps = epsilon;
        while (iteration++ < max_iterations) {
            float sum1 = 0, sum2 = 0, sum3 = 0, sum4 = 0, sum5 = 0;
            for (int i = 0; i < size; i++) {
                float f1 = ratio * chi_pdf(samples[i] / c1) / c1;
                float f2 = (float)((1.0 - (double)ratio) * (double)chi_pdf(samples[i] / c2) / (double)c2);
                float t = (float)((double)f1 / ((double)(f1 + f2) + 1e-16));
                float p = samples[i] * weights[i];
                sum1 += t * p;
                sum2 += t * weights[i];
                sum3 = (float)((double)sum3 + (1.0 - (double)t) * (double)p);
                sum4 = (float)((double)sum4 + (1.0 - (double)t) * (double)weights[i]);
                sum5 += weights[i];
            }
            sum2 = std::max(sum2, eps);
            sum3 = std::max(sum3, eps);
            sum5 = std::max(sum5, eps);                 // sum4 is not floored (stats.cxx:42-44)
            float nc1 = std::max(eps, sum1 / sum2 / esp);
            float nc2 = std::max(eps, sum3 / sum4 / esp);
            float nratio = std::max(eps, sum2 / sum5);
            bool done = (double)std::fabs((c1 - nc1) / nc1) < 0.001
                     && (double)std::fabs((c2 - nc2) / nc2) < 0.001
                     && (double)std::fabs((nratio - ratio) / nratio) < 0.001;
            c1 = nc1; c2 = nc2; ratio = nratio;
            if (done) break;
        }
    }

    // stats.cxx:121-131
    void make_histogram(float bin = 1)
    {
        float mx = *std::max_element(samples.begin(), samples.begin() + size);
        int n = (int)std::round(mx / bin) + 1;
        histogram.resize(n);
        std::fill(histogram.begin(), histogram.begin() + n, 0.f);
        for (int i = 0; i < size; i++) histogram[(size_t)std::round(samples[i] / bin)]++;
    }
};

// ---------------------------------------------------------------------------
// VTK pieces
// ---------------------------------------------------------------------------

// vtkMath::Distance2BetweenPoints(const float[3], const float[3]): f32.
inline float dist2_f32(const float *a, const float *b)
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1])
         + (a[2] - b[2]) * (a[2] - b[2]);
}

// vtkBoundingBox with double min/max.
struct BBox {
    double mn[3], mx[3];
    BBox() { reset(); }
    void reset()
    {
        for (int k = 0; k < 3; k++) { mn[k] = std::numeric_limits<double>::max(); mx[k] = -std::numeric_limits<double>::max(); }
    }
    void add_point(double x, double y, double z)
    {
        const double p[3] = { x, y, z };
        for (int k = 0; k < 3; k++) { if (p[k] < mn[k]) mn[k] = p[k]; if (p[k] > mx[k]) mx[k] = p[k]; }
    }
    void add_box(const BBox &b)
    {
        for (int k = 0; k < 3; k++) { if (b.mn[k] < mn[k]) mn[k] = b.mn[k]; if (b.mx[k] > mx[k]) mx[k] = b.mx[k]; }
    }
    bool valid() const { return mn[0] <= mx[0] && mn[1] <= mx[1] && mn[2] <= mx[2]; }
    // vtkBoundingBox::ScaleAboutCenter(s)
    void scale_about_center(double s)
    {
        if (!valid()) return;
        for (int k = 0; k < 3; k++) {
            double c = 0.5 * (mn[k] + mx[k]);
            double lo = c + s * (mn[k] - c);
            double hi = c + s * (mx[k] - c);
            mn[k] = lo; mx[k] = hi;
        }
    }
    double length(int k) const { return mx[k] - mn[k]; }
    double bound(int i) const { return (i & 1) ? mx[i >> 1] : mn[i >> 1]; }
};

// imageGroup.cxx:221-232 (same basis as vtkBSplineTransform's cubic weights)
inline void bspline_weights(double F[4], double f)
{
    const double sixth = 1.0 / 6.0;
    const double half = 0.5;
    const double f2 = f * f;
    F[3] = f2 * f * sixth;
    F[0] = (f2 - f) * half - F[3] + sixth;
    F[2] = f + F[0] - F[3] * 2;
    F[1] = 1 - F[0] - F[2] - F[3];
}

struct Grid {
    int dims[3];
    double origin[3], spacing[3], bbox[6];
    // per image: coeffs[3*G]
    std::vector<std::vector<float>> coeffs;
};

// vtkBSplineTransform::ForwardTransformPoint<float> with cubic interpolation,
// BorderModeZero, DisplacementScale 1: point -> f64 lattice coordinates, floor,
// cubic weights, separable accumulation x (inner) -> y -> z in f64 of the f32
// coefficients, taps outside the lattice contribute zero, out = in + disp -> f32.
inline void bspline_apply(const Grid &g, const float *coeffs, const float in[3], float out[3])
{
    double F[3][4];
    int i0[3];
    for (int k = 0; k < 3; k++) {
        double p = ((double)in[k] - g.origin[k]) / g.spacing[k];
        double fl = std::floor(p);
        i0[k] = (int)fl - 1;
        bspline_weights(F[k], p - fl);
    }
    double disp[3] = { 0, 0, 0 };
    const int dx = g.dims[0], dy = g.dims[1], dz = g.dims[2];
    for (int k = 0; k < 4; k++) {
        int z = i0[2] + k;
        if (z < 0 || z >= dz) continue;
        double vz[3] = { 0, 0, 0 };
        for (int j = 0; j < 4; j++) {
            int y = i0[1] + j;
            if (y < 0 || y >= dy) continue;
            double vy[3] = { 0, 0, 0 };
            for (int i = 0; i < 4; i++) {
                int x = i0[0] + i;
                if (x < 0 || x >= dx) continue;
                const float *c = coeffs + 3 * ((size_t)x + (size_t)dx * ((size_t)y + (size_t)dy * (size_t)z));
                double f = F[0][i];
                vy[0] += c[0] * f; vy[1] += c[1] * f; vy[2] += c[2] * f;
            }
            double f = F[1][j];
            vz[0] += vy[0] * f; vz[1] += vy[1] * f; vz[2] += vy[2] * f;
        }
        double f = F[2][k];
        disp[0] += vz[0] * f; disp[1] += vz[1] * f; disp[2] += vz[2] * f;
    }
    for (int k = 0; k < 3; k++) out[k] = (float)((double)in[k] + disp[k] * 1.0);
}

} // namespace

// ---------------------------------------------------------------------------
// Group
// ---------------------------------------------------------------------------

struct frogo_stats { EmStats s; };

struct frogo_group {
    frog_options opt;
    uint32_t nI = 0;
    uint32_t nf = 0;                          // numberOfFixedImages
    uint32_t ib = 0, ie = 0;              // images this instance updates (all by default); see frogo_set_range
    uint64_t P = 0, L = 0;
    std::vector<uint32_t> poff;
    std::vector<float> xyz, xyz2;
    std::vector<uint64_t> rowp;
    std::vector<uint16_t> limg;
    std::vector<uint32_t> lpt;
    std::vector<EmStats> stats;
    std::vector<double> matrix;           // nI * 16, row-major (vtkMatrix4x4::Element)
    bool deformable = false;              // image.transform is the newest B-spline
    std::vector<Grid> grids;              // allTransforms[1..]
    std::vector<std::vector<float>> gradient;   // per image 4*G (image.gradient)
    std::vector<float> point_sums;        // 4*P, oracle-only: sDisp xyz + sWeight of the last step
    bool keep_raw = false;                // oracle-only (frogo_keep_raw_gradient): copy the gradient image before the
    std::vector<std::vector<float>> gradient_raw;   // control-point step overwrites it with the proposals (:346-375)
    // Point::hardLinks (landmark constraints, imageGroup.cxx:1210-1225): CSR over all points, partner = global point
    std::vector<uint64_t> hard_rowp;      // P + 1 (empty: none)
    std::vector<uint64_t> hard_partner;
    float hard_weight2 = 0;               // (nImages * landmarksConstraintsWeight)^2, :237, :287
};

namespace {

inline const float *pos2(const frogo_group *g, uint16_t image, uint32_t point)
{
    return &g->xyz2[3 * ((size_t)g->poff[image] + point)];
}

// getBoundingBox (imageGroup.cxx:1513-1527) over the owned images (no -fi: all are moving).
void group_bbox(const frogo_group *g, BBox &box)
{
    box.reset();
    for (uint32_t i = g->ib; i < g->ie; i++) {
        BBox local;
        for (uint32_t p = g->poff[i]; p < g->poff[i + 1]; p++)
            local.add_point(g->xyz[3 * (size_t)p], g->xyz[3 * (size_t)p + 1], g->xyz[3 * (size_t)p + 2]);
        box.add_box(local);
    }
}

} // namespace

extern "C" {

frogo_group *frogo_create(const frog_model *m, const frog_options *o)
{
    frogo_group *g = new frogo_group;
    g->opt = *o;
    g->nI = m->n_images;
    g->nf = o->n_fixed_images > 0 ? (uint32_t)o->n_fixed_images : 0;   // -fi, imageGroup.h:40,69
    g->ib = g->nf; g->ie = m->n_images;
    g->poff.assign(m->point_offset, m->point_offset + m->n_images + 1);
    g->P = g->poff[g->nI];
    g->xyz.assign(m->xyz, m->xyz + 3 * g->P);
    g->xyz2.assign(3 * g->P, 0.f);
    // readAndApplyFixedImagesTransforms (:1445-1450): a fixed image's xyz2 is its registered position (the
    // caller passes xyz already transformed) and transformPoints never touches it again
    std::copy(g->xyz.begin(), g->xyz.begin() + 3 * (size_t)g->poff[g->nf], g->xyz2.begin());
    g->rowp.assign(m->row_ptr, m->row_ptr + g->P + 1);
    g->L = g->rowp[g->P];
    g->limg.assign(m->link_image, m->link_image + g->L);
    g->lpt.assign(m->link_point, m->link_point + g->L);
    g->stats.resize(g->nI);
    for (auto &s : g->stats) {
        s.max_size = o->stats_max_size;
        s.max_iterations = o->stats_max_iterations;
        s.epsilon = o->stats_epsilon;
    }
    g->matrix.assign((size_t)g->nI * 16, 0.0);
    g->gradient.resize(g->nI);
    return g;
}

void frogo_destroy(frogo_group *g) { delete g; }
void frogo_set_threads(int n) { omp_set_num_threads(n); }
int frogo_get_max_threads(void) { return omp_get_max_threads(); }

// setupStats, imageGroup.cxx:1151-1159: one slot per half-link of the image.
void frogo_setup_stats(frogo_group *g)
{
    for (uint32_t i = 0; i < g->nI; i++) {
        uint64_t n = g->rowp[g->poff[i + 1]] - g->rowp[g->poff[i]];
        for (uint64_t k = 0; k < n; k++) g->stats[i].add_slot();
    }
}

// setupLinearTransforms, imageGroup.cxx:806-848.
void frogo_linear_init(frogo_group *g, const float anchor_pos[3])
{
    std::vector<float> anchors((size_t)g->nI * 3);
    float average[3] = { 0, 0, 0 };
    for (uint32_t i = 0; i < g->nI; i++) {
        BBox box;
        for (uint32_t p = g->poff[i]; p < g->poff[i + 1]; p++)
            box.add_point(g->xyz[3 * (size_t)p], g->xyz[3 * (size_t)p + 1], g->xyz[3 * (size_t)p + 2]);
        for (int j = 0; j < 3; j++) {
            float c = anchor_pos[j];
            float a = (float)((double)(1 - c) * box.bound(2 * j) + (double)c * box.bound(1 + 2 * j));
            anchors[3 * (size_t)i + j] = a;
            if (i < g->nI - g->nf) average[j] += a / (float)(g->nI - g->nf);      // :823-824
        }
    }
    for (uint32_t i = 0; i < g->nI; i++) {
        double *M = &g->matrix[(size_t)i * 16];
        for (int k = 0; k < 16; k++) M[k] = (k % 5 == 0) ? 1.0 : 0.0;
        for (int j = 0; j < 3; j++) M[4 * j + 3] = (double)(average[j] - anchors[3 * (size_t)i + j]);
    }
    g->deformable = false;
}

// transformPoints, imageGroup.cxx:910-916 -> Image::transformPoints, image.cxx:3-13.
// image.transform is the matrix during the linear stage and the newest B-spline
// during the deformable stage (never the concatenation).
void frogo_transform_points(frogo_group *g, int apply)
{
    #pragma omp parallel for
    for (int i = (int)g->ib; i < (int)g->ie; i++) {
        const double *M = &g->matrix[(size_t)i * 16];
        const Grid *grid = g->deformable ? &g->grids.back() : nullptr;
        const float *coeffs = grid ? grid->coeffs[i].data() : nullptr;
        for (uint32_t p = g->poff[i]; p < g->poff[i + 1]; p++) {
            float *in = &g->xyz[3 * (size_t)p];
            float *out = &g->xyz2[3 * (size_t)p];
            if (grid) {
                bspline_apply(*grid, coeffs, in, out);
            } else {
                // vtkLinearTransformPoint(double M[4][4], const float in[3], float out[3])
                float x = (float)(M[0] * in[0] + M[1] * in[1] + M[2] * in[2] + M[3]);
                float y = (float)(M[4] * in[0] + M[5] * in[1] + M[6] * in[2] + M[7]);
                float z = (float)(M[8] * in[0] + M[9] * in[1] + M[10] * in[2] + M[11]);
                out[0] = x; out[1] = y; out[2] = z;
            }
            if (!apply) continue;
            for (int k = 0; k < 3; k++) in[k] = out[k];
        }
    }
}

// updateStats, imageGroup.cxx:569-598.
void frogo_update_stats(frogo_group *g)
{
    // :573 loops over every image, fixed ones included
    #pragma omp parallel for
    for (int i = (int)(g->nf ? 0 : g->ib); i < (int)g->ie; i++) {
        EmStats &st = g->stats[i];
        st.reset();
        for (uint32_t p = g->poff[i]; p < g->poff[i + 1]; p++) {
            const float *pA = &g->xyz2[3 * (size_t)p];
            for (uint64_t l = g->rowp[p]; l < g->rowp[p + 1]; l++) {
                const float *pB = pos2(g, g->limg[l], g->lpt[l]);
                float d2 = dist2_f32(pA, pB);
                st.add_sample(std::sqrt(d2));
            }
        }
        st.estimate();
    }
}

// updateLinearTransforms, imageGroup.cxx:1063-1149, over the owned images; the two
// energy sums are returned for the caller to combine across instances.
void frogo_linear_step_local(frogo_group *g, double out2[2])
{
    double sDistances = 0, sWeights = 0;
    const float la = g->opt.linear_alpha;

    #pragma omp parallel for reduction(+ : sDistances, sWeights)
    for (int image1 = (int)g->ib; image1 < (int)g->ie; image1++) {
        float diff[3];
        double sDisp[3] = { 0, 0, 0 }, sPosA[3] = { 0, 0, 0 }, sPosA2[3] = { 0, 0, 0 };
        double sPosB[3] = { 0, 0, 0 }, sPosB2[3] = { 0, 0, 0 };
        double sWeight = 0;
        const EmStats &stA = g->stats[image1];

        for (uint32_t p = g->poff[image1]; p < g->poff[image1 + 1]; p++) {
            const float *pA = &g->xyz2[3 * (size_t)p];
            for (uint64_t l = g->rowp[p]; l < g->rowp[p + 1]; l++) {
                const uint16_t image2 = g->limg[l];
                const float *pB = pos2(g, image2, g->lpt[l]);
                float dist = 0;
                for (int k = 0; k < 3; k++) { diff[k] = pB[k] - pA[k]; dist += diff[k] * diff[k]; }
                dist = std::sqrt(dist);
                float probA = stA.inlier_probability(dist);
                float probB = g->stats[image2].inlier_probability(dist);
                float w = std::min(probA, probB);
                sDistances += w * w * dist * dist;      // f32 products, f64 sum
                sWeights += w * w;
                for (int k = 0; k < 3; k++) {
                    float a = pA[k], b = pB[k];
                    sDisp[k] += w * diff[k];
                    sPosA[k] += w * a;
                    sPosB[k] += w * b;
                    sPosA2[k] += w * a * a;
                    sPosB2[k] += w * b * b;
                }
                sWeight += w;
            }
        }

        double *M = &g->matrix[(size_t)image1 * 16];
        for (int k = 0; k < 3; k++) {
            float scale = (float)M[5 * k];
            float newScale = g->opt.use_scale
                ? (float)std::pow((sWeight * sPosB2[k] - sPosB[k] * sPosB[k]) /
                                  (sWeight * sPosA2[k] - sPosA[k] * sPosA[k]), 0.5 * (double)la)
                : 1.0f;
            if (std::isnan(newScale)) continue;
            M[5 * k] = (double)(scale * newScale);
            float translation = (float)M[4 * k + 3];
            if (std::isnan(translation)) continue;
            M[4 * k + 3] = (double)translation + (double)la * sDisp[k] / sWeight
                         + sPosA[k] * (double)(1 - newScale) / sWeight;
        }
    }
    out2[0] = sDistances; out2[1] = sWeights;
}

double frogo_linear_step(frogo_group *g)
{
    double s[2];
    frogo_linear_step_local(g, s);
    return std::sqrt(s[0] / s[1]);
}

void frogo_bounds_local(frogo_group *g, double mins[3], double maxs[3])
{
    BBox box;
    group_bbox(g, box);
    for (int k = 0; k < 3; k++) { mins[k] = box.mn[k]; maxs[k] = box.mx[k]; }
}

// setupDeformableTransforms, imageGroup.cxx:159-218, from the group-wide bounding box.
void frogo_deformable_setup_bounds(frogo_group *g, int level, const double mins[3], const double maxs[3], frog_grid_info *out)
{
    Grid grid;
    double size = (double)g->opt.initial_grid_size / std::pow(2, level);
    BBox box;
    for (int k = 0; k < 3; k++) { box.mn[k] = mins[k]; box.mx[k] = maxs[k]; }
    float s = 1 + 2 * g->opt.bounding_box_margin;      // int + int*float -> float
    box.scale_about_center((double)s);
    for (int k = 0; k < 3; k++) {
        double length = box.length(k);
        int d = (int)std::round(length / size);
        if (d < 1) d = 1;
        grid.spacing[k] = length / d;
        grid.origin[k] = box.bound(2 * k) - grid.spacing[k];
        grid.dims[k] = d + 3;
        grid.bbox[2 * k] = box.mn[k]; grid.bbox[2 * k + 1] = box.mx[k];
    }
    const size_t G = (size_t)grid.dims[0] * grid.dims[1] * grid.dims[2];
    grid.coeffs.resize(g->nI);
    for (uint32_t i = 0; i < g->nI; i++) {
        grid.coeffs[i].assign(3 * G, 0.f);
        g->gradient[i].assign(4 * G, 0.f);      // AllocateScalars leaves it uninitialised; zeroed at step start
    }
    g->grids.push_back(std::move(grid));
    g->deformable = true;
    if (out) {
        const Grid &gr = g->grids.back();
        for (int k = 0; k < 3; k++) { out->dims[k] = gr.dims[k]; out->origin[k] = gr.origin[k]; out->spacing[k] = gr.spacing[k]; }
        for (int k = 0; k < 6; k++) out->bbox[k] = gr.bbox[k];
        out->n_grid = (int)g->grids.size() - 1;
    }
}

void frogo_deformable_setup(frogo_group *g, int level, frog_grid_info *out)
{
    double mn[3], mx[3];
    frogo_bounds_local(g, mn, mx);
    frogo_deformable_setup_bounds(g, level, mn, mx, out);
}

// updateDeformableTransforms, imageGroup.cxx:234-472 (no hardLinks: -lc is out of scope), in
// three phases so that several instances owning disjoint image ranges can be combined:
//   A  :239-377  per owned image: point sums, scatter, control-point step; returns the sum of
//                the proposals over the owned images (3G doubles, image order) and the energy sums
//   B  :400-432  given the sum over ALL images: subtract the mean, count oversize coefficients
//   C  :441-468  commit
void frogo_deformable_phase_a(frogo_group *g, const float alpha, double *gridsum, double out2[2])
{
    double sDistances = 0, sWeights = 0;
    Grid &grid = g->grids.back();
    const int *dims = grid.dims;
    const long inc[3] = { 4, 4L * dims[0], 4L * dims[0] * dims[1] };   // vtkImageData increments, 4 comps
    const size_t G = (size_t)dims[0] * dims[1] * dims[2];
    const float thr = g->opt.inlier_threshold;
    if (g->point_sums.size() != 4 * g->P) g->point_sums.assign(4 * g->P, 0.f);

    #pragma omp parallel for reduction(+ : sDistances, sWeights)
    for (int image1 = (int)g->ib; image1 < (int)g->ie; image1++) {
        float *gradient = g->gradient[image1].data();
        double weights[3][4];
        std::fill(gradient, gradient + 4 * G, 0.f);
        const EmStats &stA = g->stats[image1];

        for (uint32_t p = g->poff[image1]; p < g->poff[image1 + 1]; p++) {
            const float *pos = &g->xyz[3 * (size_t)p];
            const float *pA = &g->xyz2[3 * (size_t)p];
            float sWeight = 0;
            float sDisp[3] = { 0, 0, 0 };
            for (uint64_t l = g->rowp[p]; l < g->rowp[p + 1]; l++) {
                const uint16_t image2 = g->limg[l];
                const float *pB = pos2(g, image2, g->lpt[l]);
                float d2 = dist2_f32(pA, pB);
                float dist = std::sqrt(d2);
                float probA = stA.inlier_probability(dist);
                float probB = g->stats[image2].inlier_probability(dist);
                float w = std::min(probA, probB);
                float w2 = w * w;
                if (w < thr) continue;
                sWeights += w2;
                sDistances += w2 * d2;
                for (int k = 0; k < 3; k++) sDisp[k] += w2 * (pB[k] - pA[k]);
                sWeight += w2;
            }
            if (!g->hard_rowp.empty())                                   // hardLinks, :280-295
                for (uint64_t l = g->hard_rowp[p]; l < g->hard_rowp[p + 1]; l++) {
                    const float *pB = &g->xyz2[3 * g->hard_partner[l]];
                    float d2 = dist2_f32(pA, pB);
                    float w2 = g->hard_weight2;
                    sDistances += w2 * d2;
                    sWeights += w2;
                    for (int k = 0; k < 3; k++) sDisp[k] += w2 * (pB[k] - pA[k]);
                    sWeight += w2;
                }
            float *ps = &g->point_sums[4 * (size_t)p];
            ps[0] = sDisp[0]; ps[1] = sDisp[1]; ps[2] = sDisp[2]; ps[3] = sWeight;
            if (sWeight == 0) continue;

            long idZ = 0;
            for (int k = 0; k < 3; k++) {
                float coord = (float)(((double)pos[k] - grid.origin[k]) / grid.spacing[k]);
                int ic = (int)std::floor(coord);
                bspline_weights(weights[k], (double)(coord - (float)ic));
                idZ += (long)(ic - 1) * inc[k];
            }
            for (int k = 0; k < 4; k++) {
                long idY = idZ;
                for (int j = 0; j < 4; j++) {
                    long idX = idY;
                    for (int i = 0; i < 4; i++) {
                        double w = weights[0][i] * weights[1][j] * weights[2][k];
                        for (int l = 0; l < 3; l++)
                            gradient[idX + l] = (float)((double)gradient[idX + l] + w * (double)sDisp[l]);
                        gradient[idX + 3] = (float)((double)gradient[idX + 3] + w * (double)sWeight);
                        idX += inc[0];
                    }
                    idY += inc[1];
                }
                idZ += inc[2];
            }
        }

        if (g->keep_raw) g->gradient_raw[image1].assign(gradient, gradient + 4 * G);
        // control-point step, :346-375 (writes the proposal into gradient[0..2])
        const float *coeffs = grid.coeffs[image1].data();
        for (size_t i = 0; i < G; i++) {
            const float gw = gradient[4 * i + 3];
            if (gw > 0) {
                for (int j = 0; j < 3; j++)
                    gradient[4 * i + j] = coeffs[3 * i + j] + alpha * gradient[4 * i + j] / gw;
            } else {
                for (int j = 0; j < 3; j++) gradient[4 * i + j] = coeffs[3 * i + j];
            }
        }
    }

    // sum of the proposals over the owned images, :411-415 (before the division)
    #pragma omp parallel for
    for (long i = 0; i < (long)G; i++)
        for (int j = 0; j < 3; j++) {
            double sum = 0;
            for (int im = (int)g->ib; im < (int)g->ie; im++) sum += g->gradient[im][4 * i + j];
            gridsum[3 * i + j] = sum;
        }
    out2[0] = sDistances; out2[1] = sWeights;
}

// subtract the cross-image mean, count oversize coefficients, :417-428
long frogo_deformable_phase_b(frogo_group *g, const double *gridsum_all)
{
    Grid &grid = g->grids.back();
    const size_t G = (size_t)grid.dims[0] * grid.dims[1] * grid.dims[2];
    const int nImages = (int)g->nI;
    long nBig = 0;
    const float maxD = g->opt.max_displacement_ratio;
    #pragma omp parallel for reduction(+ : nBig)
    for (long i = 0; i < (long)G; i++) {
        for (int j = 0; j < 3; j++) {
            double sum = 0;
            if (g->nf == 0) {                     // :398 `apply`
                sum = gridsum_all[3 * i + j];
                sum /= nImages;
            }
            for (int im = (int)g->ib; im < (int)g->ie; im++) {
                float &v = g->gradient[im][4 * i + j];
                v = (float)((double)v - sum);
                if ((double)std::fabs(v) > (double)maxD * grid.spacing[j]) nBig++;
            }
        }
    }
    return nBig;
}

// commit, :441-468
void frogo_deformable_phase_c(frogo_group *g)
{
    Grid &grid = g->grids.back();
    const size_t G = (size_t)grid.dims[0] * grid.dims[1] * grid.dims[2];
    #pragma omp parallel for
    for (int image1 = (int)g->ib; image1 < (int)g->ie; image1++) {
        float *disp = grid.coeffs[image1].data();
        const float *nd = g->gradient[image1].data();
        for (size_t i = 0; i < G; i++)
            for (int j = 0; j < 3; j++) disp[3 * i + j] = nd[4 * i + j];
    }
}

double frogo_deformable_step(frogo_group *g, const float alpha)
{
    const Grid &grid = g->grids.back();
    std::vector<double> gridsum((size_t)3 * grid.dims[0] * grid.dims[1] * grid.dims[2]);
    double s[2];
    frogo_deformable_phase_a(g, alpha, gridsum.data(), s);
    const long nBig = frogo_deformable_phase_b(g, gridsum.data());
    if (g->opt.guarantee_diffeomorphism && nBig > 0) return -1;      // :434-439
    frogo_deformable_phase_c(g);
    return std::sqrt(s[0] / s[1]);
}

// saveErrorMaps, imageGroup.cxx:475-567 (without hardLinks: no landmarks on this path): the
// per-point residual sums of the current xyz2 (inlier links only), each added at the lattice
// node floor((xyz - origin) / spacing) of the image's gradient image, then divided by the
// accumulated weight.  Returns 4*G floats (mean residual x, y, z, weight); the reference
// leaves the result in image.gradient and writes it as errorMaps/<image>.nii.gz.
int frogo_error_map(frogo_group *g, uint32_t image1, float *out, size_t cap)
{
    if (g->grids.empty() || image1 >= g->nI) return -1;
    const Grid &grid = g->grids.back();
    const int *dims = grid.dims;
    const long inc[3] = { 4, 4L * dims[0], 4L * dims[0] * dims[1] };
    const size_t G = (size_t)dims[0] * dims[1] * dims[2];
    if (cap < 4 * G) return -1;
    std::fill(out, out + 4 * G, 0.f);                                   // :489
    const EmStats &stA = g->stats[image1];
    const float thr = g->opt.inlier_threshold;
    for (uint32_t p = g->poff[image1]; p < g->poff[image1 + 1]; p++) {
        const float *pos = &g->xyz[3 * (size_t)p];
        const float *pA = &g->xyz2[3 * (size_t)p];
        float sWeight = 0;
        float sDisp[3] = { 0, 0, 0 };
        for (uint64_t l = g->rowp[p]; l < g->rowp[p + 1]; l++) {       // :500-518
            const uint16_t image2 = g->limg[l];
            const float *pB = pos2(g, image2, g->lpt[l]);
            float d2 = dist2_f32(pA, pB);
            float dist = std::sqrt(d2);
            float probA = stA.inlier_probability(dist);
            float probB = g->stats[image2].inlier_probability(dist);
            float w = std::min(probA, probB);
            float w2 = w * w;
            if (w < thr) continue;
            for (int k = 0; k < 3; k++) sDisp[k] += w2 * (pB[k] - pA[k]);
            sWeight += w2;
        }
        if (!g->hard_rowp.empty())                                      // hardLinks, :520-533
            for (uint64_t l = g->hard_rowp[p]; l < g->hard_rowp[p + 1]; l++) {
                const float *pB = &g->xyz2[3 * g->hard_partner[l]];
                float w2 = g->hard_weight2;
                for (int k = 0; k < 3; k++) sDisp[k] += w2 * (pB[k] - pA[k]);
                sWeight += w2;
            }
        if (sWeight == 0) continue;                                     // :535
        long id = 0;
        for (int k = 0; k < 3; k++) {                                   // :539-544
            float coord = (float)(((double)pos[k] - grid.origin[k]) / grid.spacing[k]);
            id += (long)std::floor(coord) * inc[k];
        }
        if (id < 0 || (size_t)id + 3 >= 4 * G) continue;                // outside the image: UB upstream
        out[id + 3] += sWeight;
        for (int k = 0; k < 3; k++) out[id + k] += sDisp[k];
    }
    for (size_t i = 0; i < G; i++)                                      // :551-556
        if (out[4 * i + 3] > 0)
            for (int k = 0; k < 3; k++) out[4 * i + k] /= out[4 * i + 3];
    return 0;
}

// hardLinks of the landmark constraints (-lc, imageGroup.cxx:1210-1225): n directed links
// (point <- partner, global point indices), grouped by point in the order they are given.
int frogo_set_hard_links(frogo_group *g, const uint64_t *point, const uint64_t *partner, size_t n, float weight2)
{
    g->hard_rowp.assign(g->P + 1, 0);
    g->hard_partner.resize(n);
    for (size_t k = 0; k < n; k++) { if (point[k] >= g->P || partner[k] >= g->P) return -1; g->hard_rowp[point[k] + 1]++; }
    for (uint64_t p = 0; p < g->P; p++) g->hard_rowp[p + 1] += g->hard_rowp[p];
    std::vector<uint64_t> cur(g->hard_rowp.begin(), g->hard_rowp.end() - 1);
    for (size_t k = 0; k < n; k++) g->hard_partner[cur[point[k]]++] = partner[k];
    g->hard_weight2 = weight2;
    return 0;
}

void frogo_set_range(frogo_group *g, uint32_t image_begin, uint32_t image_end) { g->ib = image_begin; g->ie = image_end; }

// countInliers, imageGroup.cxx:988-1060.
void frogo_count_inliers(frogo_group *g, frog_counts *out)
{
    const float thr = g->opt.inlier_threshold;
    #pragma omp parallel for
    for (int image1 = (int)g->nf; image1 < (int)g->nI; image1++) {      // :1003
        const EmStats &stA = g->stats[image1];
        int64_t nPairs = 0, nOut = 0, nIn = 0;
        for (uint32_t p = g->poff[image1]; p < g->poff[image1 + 1]; p++) {
            const float *pA = &g->xyz2[3 * (size_t)p];
            for (uint64_t l = g->rowp[p]; l < g->rowp[p + 1]; l++) {
                const uint16_t image2 = g->limg[l];
                const float *pB = pos2(g, image2, g->lpt[l]);
                float dist = std::sqrt(dist2_f32(pA, pB));
                float w = std::min(stA.inlier_probability(dist), g->stats[image2].inlier_probability(dist));
                nPairs++;
                if (w < thr) nOut++; else nIn++;
            }
        }
        frog_counts &c = out[image1];
        c.points = g->poff[image1 + 1] - g->poff[image1];
        c.pairs = nPairs; c.inliers = nIn; c.outliers = nOut;
        c.c1 = stA.c1; c.c2 = stA.c2; c.ratio = stA.ratio; c.pad_ = 0;
    }
}

// run(), imageGroup.cxx:31-157, no fixed images, no landmarks, no file output.
int frogo_run(frogo_group *g, int linear_iterations, int deformable_levels,
              int deformable_iterations, float deformable_alpha, int stat_interval,
              const float anchor[3], double *E_out, int cap, int *n_grids_out)
{
    int nE = 0;
    auto push = [&](double e) { if (nE < cap) E_out[nE] = (double)(float)e; nE++; };
    frogo_setup_stats(g);
    frogo_linear_init(g, anchor);
    frogo_transform_points(g, 0);
    for (int it = 0; it < linear_iterations; it++) {
        if (!(it % stat_interval)) frogo_update_stats(g);
        double e = frogo_linear_step(g);
        frogo_transform_points(g, 0);
        push(e);
    }
    frogo_transform_points(g, 1);
    for (int level = 0; level < deformable_levels; level++) {
        frogo_deformable_setup(g, level, nullptr);
        frogo_transform_points(g, 0);
        int nGrids = 1;
        float alpha = deformable_alpha;
        int nDiffeo = 0;
        for (int it = 0; it < deformable_iterations; it++) {
            if (!(it % stat_interval)) frogo_update_stats(g);
            float e = (float)frogo_deformable_step(g, alpha);
            if (e < 0) {
                if (nDiffeo == 0) alpha /= 2;
                nGrids++;
                it--;
                frogo_transform_points(g, 1);
                frogo_deformable_setup(g, level, nullptr);
                frogo_transform_points(g, 0);
                nDiffeo = 0;
                continue;
            }
            nDiffeo++;
            frogo_transform_points(g, 0);
            push(e);
        }
        if (n_grids_out) n_grids_out[level] = nGrids;
        frogo_transform_points(g, 1);
    }
    return nE;
}

uint64_t frogo_num_points(const frogo_group *g) { return g->P; }
// raw views for the oracle's other translation units (ransac_oracle.cpp)
const float *frogo_xyz_ptr(const frogo_group *g) { return g->xyz.data(); }
const float *frogo_xyz2_ptr(const frogo_group *g) { return g->xyz2.data(); }
const uint32_t *frogo_point_offset_ptr(const frogo_group *g) { return g->poff.data(); }
const uint64_t *frogo_row_ptr(const frogo_group *g) { return g->rowp.data(); }
const uint16_t *frogo_link_image_ptr(const frogo_group *g) { return g->limg.data(); }
const uint32_t *frogo_link_point_ptr(const frogo_group *g) { return g->lpt.data(); }
void frogo_set_matrix(frogo_group *g, uint32_t image, const double in16[16]) { std::memcpy(&g->matrix[(size_t)image * 16], in16, 16 * sizeof(double)); }
void frogo_get_xyz(const frogo_group *g, float *out) { std::memcpy(out, g->xyz.data(), g->xyz.size() * sizeof(float)); }
void frogo_get_xyz2(const frogo_group *g, float *out) { std::memcpy(out, g->xyz2.data(), g->xyz2.size() * sizeof(float)); }
void frogo_set_xyz2(frogo_group *g, const float *in) { std::memcpy(g->xyz2.data(), in, g->xyz2.size() * sizeof(float)); }
void frogo_get_matrix(const frogo_group *g, uint32_t image, double out16[16])
{
    std::memcpy(out16, &g->matrix[(size_t)image * 16], 16 * sizeof(double));
}
void frogo_get_em(const frogo_group *g, uint32_t image, float out3[3])
{
    out3[0] = g->stats[image].c1; out3[1] = g->stats[image].c2; out3[2] = g->stats[image].ratio;
}
void frogo_set_em(frogo_group *g, uint32_t image, const float in3[3])
{
    g->stats[image].c1 = in3[0]; g->stats[image].c2 = in3[1]; g->stats[image].ratio = in3[2];
}
int frogo_get_samples(const frogo_group *g, uint32_t image, float *out, int cap)
{
    const EmStats &s = g->stats[image];
    int n = std::min(cap, s.size);
    std::memcpy(out, s.samples.data(), (size_t)n * sizeof(float));
    return s.size;
}
int frogo_get_sample_ordinals(const frogo_group *g, uint32_t image, uint32_t *out, int cap)
{
    const EmStats &s = g->stats[image];
    int n = std::min(cap, s.size);
    std::memcpy(out, s.ordinals.data(), (size_t)n * sizeof(uint32_t));
    return s.size;
}
int frogo_get_histogram(frogo_group *g, uint32_t image, float *out, int cap)
{
    EmStats &s = g->stats[image];
    s.make_histogram();
    int n = std::min<int>(cap, (int)s.histogram.size());
    std::memcpy(out, s.histogram.data(), (size_t)n * sizeof(float));
    return (int)s.histogram.size();
}
int frogo_num_grids(const frogo_group *g) { return (int)g->grids.size(); }
int frogo_get_grid(const frogo_group *g, uint32_t image, int k, frog_grid_info *info, float *coeffs, size_t cap)
{
    if (k < 0 || k >= (int)g->grids.size()) return -1;
    const Grid &gr = g->grids[k];
    if (info) {
        for (int a = 0; a < 3; a++) { info->dims[a] = gr.dims[a]; info->origin[a] = gr.origin[a]; info->spacing[a] = gr.spacing[a]; }
        for (int a = 0; a < 6; a++) info->bbox[a] = gr.bbox[a];
        info->n_grid = k;
    }
    if (coeffs) {
        size_t n = std::min(cap, gr.coeffs[image].size());
        std::memcpy(coeffs, gr.coeffs[image].data(), n * sizeof(float));
    }
    return 0;
}
void frogo_get_point_sums(const frogo_group *g, float *out)
{
    std::memcpy(out, g->point_sums.data(), g->point_sums.size() * sizeof(float));
}
// oracle-only: keep (and read back) the gradient image as the scatter left it, :301-338
void frogo_keep_raw_gradient(frogo_group *g, int on)
{
    g->keep_raw = on != 0;
    g->gradient_raw.resize(g->keep_raw ? g->nI : 0);
}

int frogo_get_gradient_raw(const frogo_group *g, uint32_t image, float *out, size_t cap)
{
    if (!g->keep_raw || image >= g->gradient_raw.size()) return -1;
    size_t n = std::min(cap, g->gradient_raw[image].size());
    std::memcpy(out, g->gradient_raw[image].data(), n * sizeof(float));
    return (int)g->gradient_raw[image].size();
}

int frogo_get_gradient(const frogo_group *g, uint32_t image, float *out, size_t cap)
{
    size_t n = std::min(cap, g->gradient[image].size());
    std::memcpy(out, g->gradient[image].data(), n * sizeof(float));
    return (int)g->gradient[image].size();
}

// ---- stand-alone Stats --------------------------------------------------------
frogo_stats *frogo_stats_new(int max_size, int max_iterations, float epsilon)
{
    frogo_stats *s = new frogo_stats;
    s->s.max_size = max_size; s->s.max_iterations = max_iterations; s->s.epsilon = epsilon;
    return s;
}
void frogo_stats_free(frogo_stats *s) { delete s; }
void frogo_stats_add_slots(frogo_stats *s, int n) { for (int i = 0; i < n; i++) s->s.add_slot(); }
void frogo_stats_reset(frogo_stats *s) { s->s.reset(); }
void frogo_stats_add_samples(frogo_stats *s, const float *v, int n) { for (int i = 0; i < n; i++) s->s.add_sample(v[i]); }
void frogo_stats_estimate(frogo_stats *s) { s->s.estimate(); }
float frogo_stats_inlier_probability(const frogo_stats *s, float d) { return s->s.inlier_probability(d); }
void frogo_stats_get_params(const frogo_stats *s, float o[3]) { o[0] = s->s.c1; o[1] = s->s.c2; o[2] = s->s.ratio; }
void frogo_stats_set_params(frogo_stats *s, const float i[3]) { s->s.c1 = i[0]; s->s.c2 = i[1]; s->s.ratio = i[2]; }
int frogo_stats_size(const frogo_stats *s) { return s->s.size; }
int frogo_stats_get_samples(const frogo_stats *s, float *out, int cap)
{
    int n = std::min(cap, s->s.size);
    std::memcpy(out, s->s.samples.data(), (size_t)n * sizeof(float));
    return s->s.size;
}
int frogo_stats_histogram(frogo_stats *s, float bin, float *out, int cap)
{
    s->s.make_histogram(bin);
    int n = std::min<int>(cap, (int)s->s.histogram.size());
    std::memcpy(out, s->s.histogram.data(), (size_t)n * sizeof(float));
    return (int)s->s.histogram.size();
}
float frogo_chipdf(float x) { return chi_pdf(x); }
void frogo_bspline_weights_n(const double *f, int n, double *out4n) { for (int i = 0; i < n; i++) bspline_weights(out4n + 4 * (long)i, f[i]); }

} // extern "C"
