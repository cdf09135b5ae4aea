// k_ransac.hip.h -- inlier census of many candidate similarity transforms at once
// (the inner loops of ImageGroup::RANSACBatch, imageGroup.cxx:762-785).
//
// The reference evaluates 5 000 candidates one after the other, each over every half-link of
// the image.  Here one launch evaluates all candidates: a thread keeps RANSAC_LINKS half-links
// (own point's xyz, partner's xyz2) in registers and walks the candidate list; the matrices are
// wave-uniform (scalar loads).  Per candidate the wave counts its inliers with a ballot and
// parks the count in the lane `candidate % 64`, so that global memory sees one 64-wide atomic
// per 64 candidates instead of one per candidate.
// Arithmetic as upstream: vtkLinearTransform::TransformPoint(float in, float out) = f64 row
// sums rounded to f32, vtkMath::Distance2BetweenPoints(float, float) in f32, `<` against
// (float)pow(distance, 2).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ctx.h"

namespace frog {

constexpr int RANSAC_LINKS = 8;          // half-links per thread

__global__ __launch_bounds__(256) void ransac_count_kernel(
    const float4 *__restrict__ pos, const P3 *__restrict__ pos2,
    const uint64_t *__restrict__ ref_rowptr, const uint32_t *__restrict__ ref_link, const uint32_t *__restrict__ new_of_old,
    uint32_t pt_begin, uint32_t pt_end,             // the image's points, local reference numbering
    const double *__restrict__ cand, uint32_t n_cand, float max_d2, unsigned int *__restrict__ counts)
{
    const uint64_t l_begin = ref_rowptr[pt_begin], l_end = ref_rowptr[pt_end];
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float ax[RANSAC_LINKS], ay[RANSAC_LINKS], az[RANSAC_LINKS], bx[RANSAC_LINKS], by[RANSAC_LINKS], bz[RANSAC_LINKS];
    bool live[RANSAC_LINKS];
    #pragma unroll
    for (int k = 0; k < RANSAC_LINKS; k++) {
        const uint64_t l = l_begin + (wave * RANSAC_LINKS + k) * 64 + lane;
        live[k] = l < l_end;
        ax[k] = ay[k] = az[k] = bx[k] = by[k] = bz[k] = 0.f;
        if (!live[k]) continue;
        uint32_t lo = pt_begin, hi = pt_end;        // largest p with rowptr[p] <= l
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (ref_rowptr[mid] <= l) lo = mid; else hi = mid;
        }
        const float4 a = pos[new_of_old[lo]];
        const P3 b = pos2[ref_link[l]];
        ax[k] = a.x; ay[k] = a.y; az[k] = a.z; bx[k] = b.x; by[k] = b.y; bz[k] = b.z;
    }
    unsigned int mine = 0;
    for (uint32_t c = 0; c < n_cand; c++) {
        const double *m = cand + (size_t)c * 12;
        const double m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5], m6 = m[6], m7 = m[7],
                     m8 = m[8], m9 = m[9], m10 = m[10], m11 = m[11];
        unsigned int n = 0;
        #pragma unroll
        for (int k = 0; k < RANSAC_LINKS; k++) {
            const float tx = (float)(m0 * ax[k] + m1 * ay[k] + m2 * az[k] + m3);
            const float ty = (float)(m4 * ax[k] + m5 * ay[k] + m6 * az[k] + m7);
            const float tz = (float)(m8 * ax[k] + m9 * ay[k] + m10 * az[k] + m11);
            const float dx = tx - bx[k], dy = ty - by[k], dz = tz - bz[k];
            const float d2 = dx * dx + dy * dy + dz * dz;
            n += (unsigned int)__popcll(__ballot(live[k] && d2 < max_d2));
        }
        if ((int)(c & 63u) == lane) mine = n;
        if ((c & 63u) == 63u || c + 1 == n_cand) {
            const uint32_t slot = (c & ~63u) + lane;
            if (slot < n_cand && mine) atomicAdd(&counts[slot], mine);
            mine = 0;
        }
    }
}

} // namespace frog
