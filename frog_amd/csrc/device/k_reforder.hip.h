// k_reforder.hip.h -- the solver loops once more, in the REFERENCE'S OWN ORDER AND ARITHMETIC (test hook,
// FROG_REFERENCE_ORDER=1 at frog_create).
//
// The product kernels (k_links.hip.h, k_grid.hip.h) re-associate the reference's sums: per-point f32 sums per partner
// group, tree-reduced f64 sums in the linear step, a tiled scatter with the cubic weights rounded once, a fast f32 inlier
// weight.  Each of those moves a result by ~1e-7 per step, and weakly determined control points amplify that to 1e-4 of
// a lattice's range.  To show that re-association is ALL that separates the product path from the
// reference, this file runs the same loops with nothing re-associated:
//   * every inlier weight through inlier_probability_exact (stats.h:84-92 with its own promotions);
//   * a point's f32 sums as ONE chain over its half-links in readPairs order (imageGroup.cxx:252-299);
//   * the linear step's 18 f64 sums as one chain per image over points and links in order (:1080-1121);
//   * the B-spline scatter image by image, point by point in index order, f64 tap product and f32
//     read-modify-write (:301-338); the control-point step and the image-order f64 proposal sums (:346-375, :400-419);
//   * the B-spline transform without fused multiply-adds (vtkBSplineTransform as restated in DESIGN.md section 2).
// Its results are compared with the tests' CPU restatement of the reference by np.array_equal (tests/test_gpu_reference_order.py); the product path
// is then compared with THIS mode on the device, at sizes the CPU restatement cannot reach in a test.
// The kernels of this file are the LITERAL forms (rounds 4-5: a chain walked by one lane, the scatter by one wavefront per image
// with a barrier per point; 22 iterations/s on the benchmark group).  Since round 6 the mode runs through k_refchain.hip.h --
// only what the reference's order really binds is kept serial: 265 iterations/s, the same bits -- and these stay behind
// FROG_REF_LITERAL=1 as what the new forms are held against (tests/test_gpu_round6.py); the control-point step, the B-spline
// transform and the update of an image's matrix from its sums are used by both.
#pragma once

#include "ctx.h"
#include "k_links.hip.h"
#include "k_grid.hip.h"

namespace frog {

__device__ __forceinline__ float ref_min(float a, float b) { return (b < a) ? b : a; }     // std::min(a, b)

// correctly rounded f32 square root, denormals included: the f64 root of an f32 value rounds to it (53 >= 2 * 24 + 2)
__device__ __forceinline__ float ref_sqrt(float x) { return (float)sqrt((double)x); }

// Per half-link of the owned rows, in reference order (ref_rowptr / ref_link, prep.h): the own point (internal
// numbering), dist = sqrt(|pB - pA|^2) and w = min(probA(dist), probB(dist)) (imageGroup.cxx:1086-1099).  Thread per row.
__global__ __launch_bounds__(256) void ref_link_terms_kernel(const uint64_t *rowptr, const uint32_t *link, const uint32_t *new_of_old,
                                                             uint32_t n_rows, const float4 *pos, const P3 *pos2, const float4 *em,
                                                             uint32_t *own, float *w_out, float *d_out)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t a = new_of_old[r];
    const P3 pA = pos2[a];
    const float4 emA = em[__float_as_int(pos[a].w)];
    for (uint64_t l = rowptr[r]; l < rowptr[r + 1]; l++) {
        const uint32_t b = link[l];
        const P3 pB = pos2[b];
        const float4 emB = em[__float_as_int(pos[b].w)];
        const float dx = pB.x - pA.x, dy = pB.y - pA.y, dz = pB.z - pA.z;
        const float dist = ref_sqrt(dx * dx + dy * dy + dz * dz);
        const float w = ref_min(inlier_probability_exact(dist, emA), inlier_probability_exact(dist, emB));
        own[l] = a; w_out[l] = w; d_out[l] = dist;
    }
}

// imageGroup.cxx:1123-1143 for one axis of one image, from the image's 16 sums (shared with linear_update_kernel's arithmetic)
__device__ __forceinline__ void ref_linear_update_axis(double *M, int k, const double *sums, float linear_alpha, int use_scale)
{
    const double sDisp = sums[k], sPosA = sums[3 + k], sPosB = sums[6 + k];
    const double sPosA2 = sums[9 + k], sPosB2 = sums[12 + k], sWeight = sums[15];
    const float scale = (float)M[5 * k];
    float newScale = 1.0f;
    if (use_scale)
        newScale = (float)pow((sWeight * sPosB2 - sPosB * sPosB) / (sWeight * sPosA2 - sPosA * sPosA), 0.5 * (double)linear_alpha);
    if (isnan(newScale)) return;
    M[5 * k] = (double)(scale * newScale);
    const float translation = (float)M[4 * k + 3];
    if (isnan(translation)) return;
    M[4 * k + 3] = (double)translation + (double)linear_alpha * sDisp / sWeight + sPosA * (double)(1 - newScale) / sWeight;
}

// updateLinearTransforms for one image (imageGroup.cxx:1080-1143): block = one wavefront.  64 half-links at a time: lane j
// forms the 18 f32 terms of half-link base + j (f32 products left to right, as :1102-1117) and parks them in LDS; lane s < 18
// then adds term s of the 64 half-links, IN ORDER, to its f64 running sum.  Sum layout = LINEAR_SUMS of k_links.hip.h.
__global__ __launch_bounds__(64) void ref_linear_chain_kernel(const uint64_t *img_link, const uint32_t *link, const uint32_t *own,
                                                              const float *w_in, const float *d_in, const P3 *pos2,
                                                              uint32_t image_begin, double *mat, float linear_alpha, int use_scale,
                                                              double *img_energy)
{
    __shared__ float terms[64][LINEAR_SUMS + 1];
    __shared__ double sums[LINEAR_SUMS];
    const int lane = threadIdx.x;
    const uint64_t l0 = img_link[blockIdx.x], l1 = img_link[blockIdx.x + 1];
    double acc = 0.0;
    for (uint64_t base = l0; base < l1; base += 64) {
        const uint64_t l = base + lane;
        if (l < l1) {
            const P3 pA = pos2[own[l]], pB = pos2[link[l]];
            const float w = w_in[l], dist = d_in[l];
            const float a[3] = { pA.x, pA.y, pA.z }, b[3] = { pB.x, pB.y, pB.z };
            #pragma unroll
            for (int k = 0; k < 3; k++) {
                const float diff = b[k] - a[k];
                terms[lane][k] = w * diff;
                terms[lane][3 + k] = w * a[k];
                terms[lane][6 + k] = w * b[k];
                terms[lane][9 + k] = w * a[k] * a[k];
                terms[lane][12 + k] = w * b[k] * b[k];
            }
            terms[lane][15] = w;
            terms[lane][16] = w * w * dist * dist;
            terms[lane][17] = w * w;
        }
        __syncthreads();
        const int n = (int)min((uint64_t)64, l1 - base);
        if (lane < LINEAR_SUMS)
            for (int j = 0; j < n; j++) acc += (double)terms[j][lane];
        __syncthreads();
    }
    if (lane < LINEAR_SUMS) sums[lane] = acc;
    __syncthreads();
    if (lane < 3) ref_linear_update_axis(mat + (size_t)(image_begin + blockIdx.x) * 16, lane, sums, linear_alpha, use_scale);
    if (lane == 0) { img_energy[2 * blockIdx.x] = sums[16]; img_energy[2 * blockIdx.x + 1] = sums[17]; }
}

// (sDistances, sWeights) of the owned images added in image order -> energy[0..1]; [2] (oversize count) and [3] (list flag) = 0.
// (The reference adds them with an omp reduction, imageGroup.cxx:239, :1067: its own order depends on the thread count.)
__global__ void ref_energy_total_kernel(const double *img_energy, uint32_t n_owned, double *energy)
{
    if (blockIdx.x || threadIdx.x) return;
    double a = 0, b = 0;
    for (uint32_t i = 0; i < n_owned; i++) { a += img_energy[2 * i]; b += img_energy[2 * i + 1]; }
    energy[0] = a; energy[1] = b; energy[2] = 0.0; energy[3] = 0.0;
}

// imageGroup.cxx:252-278 for one point: sDisp / sWeight as ONE f32 chain over the half-links in readPairs order, links with
// w < threshold skipped; the point's energy terms (f64) in pt_energy[2 r], [2 r + 1].  Thread per owned row.
__global__ __launch_bounds__(256) void ref_point_sums_kernel(const uint64_t *rowptr, const uint32_t *link, const uint32_t *new_of_old,
                                                             uint32_t n_rows, const float4 *pos, const P3 *pos2, const float4 *em,
                                                             float threshold, float4 *point_sums, double *pt_energy)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t a = new_of_old[r];
    const P3 pA = pos2[a];
    const float4 emA = em[__float_as_int(pos[a].w)];
    float sx = 0, sy = 0, sz = 0, sw = 0;
    double ed = 0, ew = 0;
    for (uint64_t l = rowptr[r]; l < rowptr[r + 1]; l++) {
        const uint32_t b = link[l];
        const P3 pB = pos2[b];
        const float4 emB = em[__float_as_int(pos[b].w)];
        // vtkMath::Distance2BetweenPoints(pA, pB): (a - b)^2 summed x, y, z in f32
        const float ex = pA.x - pB.x, ey = pA.y - pB.y, ez = pA.z - pB.z;
        const float d2 = ex * ex + ey * ey + ez * ez;
        const float dist = ref_sqrt(d2);
        const float w = ref_min(inlier_probability_exact(dist, emA), inlier_probability_exact(dist, emB));
        const float w2 = w * w;
        if (w < threshold) continue;
        ew += (double)w2;
        ed += (double)(w2 * d2);
        sx += w2 * (pB.x - pA.x); sy += w2 * (pB.y - pA.y); sz += w2 * (pB.z - pA.z);
        sw += w2;
    }
    point_sums[a] = make_float4(sx, sy, sz, sw);
    if (pt_energy) { pt_energy[2 * (size_t)r] = ed; pt_energy[2 * (size_t)r + 1] = ew; }
}

// the points' energy terms of one image added in point order (block = one wavefront: 64 loads side by side, lane 0 adds)
__global__ __launch_bounds__(64) void ref_image_energy_kernel(const double *pt_energy, const uint32_t *poff, uint32_t image_begin,
                                                              uint32_t own_pt_begin, double *img_energy)
{
    __shared__ double buf[2][64];
    const uint32_t r0 = poff[image_begin + blockIdx.x] - own_pt_begin, r1 = poff[image_begin + blockIdx.x + 1] - own_pt_begin;
    double a = 0, b = 0;
    for (uint32_t base = r0; base < r1; base += 64) {
        const uint32_t r = base + threadIdx.x;
        if (r < r1) { buf[0][threadIdx.x] = pt_energy[2 * (size_t)r]; buf[1][threadIdx.x] = pt_energy[2 * (size_t)r + 1]; }
        __syncthreads();
        if (threadIdx.x == 0)
            for (uint32_t j = 0; j < min(64u, r1 - base); j++) { a += buf[0][j]; b += buf[1][j]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { img_energy[2 * blockIdx.x] = a; img_energy[2 * blockIdx.x + 1] = b; }
}

// The B-spline scatter of one image (imageGroup.cxx:301-338), block = one wavefront, lane = tap (i + 4 j + 16 k): the
// image's points in index order, one at a time; every lane forms the point's cell and cubic weights (f64 divide rounded to
// f32, floor, weights of the f32 fraction: :303-310), its tap's weight w = wx[i] * wy[j] * wz[k] in f64 and adds
// (float)((double) g + w * (double) s) to the four components of its node.  The 64 taps of a point are 64 different nodes;
// a barrier between points orders the read-modify-writes of consecutive points on a shared node.  `gradf` was zeroed (:249).
__global__ __launch_bounds__(64) void ref_scatter_kernel(const float4 *pos, const float4 *point_sums, const uint32_t *new_of_old,
                                                         const uint32_t *poff, uint32_t image_begin, uint32_t own_pt_begin,
                                                         const GeomDev g, float4 *gradf)
{
    const int lane = threadIdx.x;
    const int ti = lane & 3, tj = (lane >> 2) & 3, tk = lane >> 4;
    const uint32_t r0 = poff[image_begin + blockIdx.x] - own_pt_begin, r1 = poff[image_begin + blockIdx.x + 1] - own_pt_begin;
    float4 *grad = gradf + (size_t)blockIdx.x * g.n_cp;
    for (uint32_t r = r0; r < r1; r++) {
        const uint32_t p = new_of_old[r];
        const float4 s = point_sums[p];
        if (s.w == 0.0f) continue;                  // :299 (uniform: every lane reads the same point)
        const float4 v = pos[p];
        const float in[3] = { v.x, v.y, v.z };
        double W[3][4];
        int i0[3];
        #pragma unroll
        for (int k = 0; k < 3; k++) {
            const float coord = (float)(((double)in[k] - g.origin[k]) / g.spacing[k]);
            const float fl = floorf(coord);
            i0[k] = (int)fl - 1;
            bspline_weights(W[k], (double)(coord - fl));
        }
        const double w = W[0][ti] * W[1][tj] * W[2][tk];
        const int x = i0[0] + ti, y = i0[1] + tj, z = i0[2] + tk;
        if (x >= 0 && y >= 0 && z >= 0 && x < g.dims[0] && y < g.dims[1] && z < g.dims[2]) {      // outside: undefined upstream
            float4 *node = grad + ((size_t)x + (size_t)g.dims[0] * ((size_t)y + (size_t)g.dims[1] * (size_t)z));
            float4 t = *node;
            t.x = (float)((double)t.x + w * (double)s.x);
            t.y = (float)((double)t.y + w * (double)s.y);
            t.z = (float)((double)t.z + w * (double)s.z);
            t.w = (float)((double)t.w + w * (double)s.w);
            *node = t;
        }
        __syncthreads();
    }
}

// Control-point step (imageGroup.cxx:346-375) and the sum of the proposals over the owned images in image order (:411-415,
// before the division).  Thread per control point.  grad receives (proposal xyz, gradient weight).
__global__ __launch_bounds__(256) void ref_cp_step_kernel(const float4 *gradf, const float4 *coeff, float4 *grad, uint32_t n_owned,
                                                          int n_cp, float alpha, double *gridsum)
{
    const int cp = blockIdx.x * blockDim.x + threadIdx.x;
    if (cp >= n_cp) return;
    double sx = 0, sy = 0, sz = 0;
    for (uint32_t i = 0; i < n_owned; i++) {
        const size_t o = (size_t)i * n_cp + cp;
        const float4 gr = gradf[o], c = coeff[o];
        float4 n4;
        const float gw = gr.w;
        if (gw > 0) {
            n4.x = c.x + alpha * gr.x / gw;
            n4.y = c.y + alpha * gr.y / gw;
            n4.z = c.z + alpha * gr.z / gw;
        } else {
            n4.x = c.x; n4.y = c.y; n4.z = c.z;
        }
        n4.w = gw;
        grad[o] = n4;
        sx += (double)n4.x; sy += (double)n4.y; sz += (double)n4.z;
    }
    gridsum[3 * (size_t)cp] = sx; gridsum[3 * (size_t)cp + 1] = sy; gridsum[3 * (size_t)cp + 2] = sz;
}

// vtkBSplineTransform's cubic forward transform, BorderModeZero: separable x -> y -> z accumulation in f64 with a separate multiplication and addition per tap (the
// product kernels fuse them: one rounding at 1e-16 less before the result is rounded to f32).  Thread per owned point.
// `proposal` / `energy` / `host_scalars`: as transform_bspline_kernel (k_grid.hip.h).
__global__ __launch_bounds__(256) void ref_transform_bspline_kernel(float4 *pos, P3 *pos2, const float4 *coeff, uint32_t pt_begin,
                                                                    uint32_t pt_end, uint32_t image_begin, const GeomDev g, int apply,
                                                                    const float4 *proposal, const double *energy, int guarantee,
                                                                    double *host_scalars, double seq)
{
    publish_step_scalars(energy, host_scalars, seq);
    if (proposal && !(guarantee && energy[2] > 0.0)) coeff = proposal;
    const uint32_t p = pt_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= pt_end) return;
    const float4 v = pos[p];
    const float4 *cf = coeff + (size_t)(__float_as_int(v.w) - (int)image_begin) * g.n_cp;
    const float in[3] = { v.x, v.y, v.z };
    double F[3][4];
    int i0[3];
    #pragma unroll
    for (int k = 0; k < 3; k++) {
        const double q = ((double)in[k] - g.origin[k]) / g.spacing[k];
        const double fl = floor(q);
        i0[k] = (int)fl - 1;
        bspline_weights(F[k], q - fl);
    }
    double disp[3] = { 0, 0, 0 };
    const int dx = g.dims[0], dy = g.dims[1], dz = g.dims[2];
    for (int k = 0; k < 4; k++) {
        const int z = i0[2] + k;
        if (z < 0 || z >= dz) continue;
        double vz[3] = { 0, 0, 0 };
        for (int j = 0; j < 4; j++) {
            const int y = i0[1] + j;
            if (y < 0 || y >= dy) continue;
            double vy[3] = { 0, 0, 0 };
            const float4 *row = cf + (size_t)dx * ((size_t)y + (size_t)dy * z);
            for (int i = 0; i < 4; i++) {
                const int x = i0[0] + i;
                if (x < 0 || x >= dx) continue;
                const float4 c = row[x];
                const double f = F[0][i];
                vy[0] += (double)c.x * f; vy[1] += (double)c.y * f; vy[2] += (double)c.z * f;
            }
            const double f = F[1][j];
            vz[0] += vy[0] * f; vz[1] += vy[1] * f; vz[2] += vy[2] * f;
        }
        const double f = F[2][k];
        disp[0] += vz[0] * f; disp[1] += vz[1] * f; disp[2] += vz[2] * f;
    }
    float4 o;
    o.x = (float)((double)in[0] + disp[0] * 1.0);
    o.y = (float)((double)in[1] + disp[1] * 1.0);
    o.z = (float)((double)in[2] + disp[2] * 1.0);
    o.w = v.w;
    pos2[p] = P3{ o.x, o.y, o.z };
    if (apply) pos[p] = o;
}

} // namespace frog
