// match_oracle.cpp -- CPU restatement of the reference's pairing stage (match/match.cpp).
// TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and the cpu_baseline leg of the
// benches may use it; the product (libfrog_hip.so) never does.
//
// PARITY PINNED (round 6) for ComputeMatches and `norm`: match.cpp as a whole cannot be built here (its main needs boost
// filesystem / iostreams and the VTK transform reader), but `struct Point` (:28-48), the scalar `norm` (:243-251) and
// ComputeMatches (:255-336) are std-only, so `make -C oracle ref` compiles exactly those, cut out of the file where it lies, into
// oracle/_ref/libfrog_refmatch.so (ref_match_api.cpp).  tests/test_match_oracle_ref.py: this restatement returns the reference
// build's pair lists on every option set (-d, -d2, -anat, -sym, -all, ties, exact duplicates, empty images, six descriptor
// lengths), its norm has the same bits, and tests/golden/match_golden.json holds the reference build's answers for where
// oracle/_ref is absent; the hand-worked cases of tests/test_match_oracle.py run against both.  The pair LOOP of main (:616-660,
// which image pairs, in which order, -sym appended) stays a restatement: main is what cannot be built.  What is restated:
//   norm (scalar build)          match.cpp:242-251
//   ComputeMatches               match.cpp:255-336
//   the pair loop of main        match.cpp:616-660
#include "../include/frog_match.h"

#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <omp.h>
#include <vector>

namespace {

// match.cpp:242-251
inline float norm_sq(const float *p1, const float *p2, int size)
{
    float result = .0;
    for (int i = 0; i < size; i++) result += (p1[i] - p2[i]) * (p1[i] - p2[i]);
    return result;
}

// match.cpp:255-336.  points2 = candidates, points1 = queries.
void compute_matches(const frog_keypoints &points2, const frog_keypoints &points1, float threshold, float dist2second,
                     bool matchAll, float anatVal, bool sym, std::vector<uint32_t> &out_a, std::vector<uint32_t> &out_b)
{
    float d1, d2;
    int match = 0;                                  // declared outside the query loop upstream: it carries over
    const int end1 = (int)points1.n;
    for (int i = 0; i < end1; i++) {
        d1 = d2 = FLT_MAX;
        const int end2 = (int)points2.n;
        for (int j = 0; j < end2; j++) {
            if (points1.laplacian[i] != points2.laplacian[j]) continue;                         // :270
            if ((points1.scale[i] / points2.scale[j] > 1.3) || (points2.scale[j] / points1.scale[i] > 1.3))   // :273-275
                continue;
            if (anatVal != 0) {                                                                 // :278-291
                float x1 = points1.xyz[3 * i], y1 = points1.xyz[3 * i + 1], z1 = points1.xyz[3 * i + 2];
                float x2 = points2.xyz[3 * j], y2 = points2.xyz[3 * j + 1], z2 = points2.xyz[3 * j + 2];
                float euclNorm = std::sqrt((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
                if (euclNorm > anatVal) continue;
            }
            float dist = norm_sq(points1.desc + (size_t)i * points1.dim, points2.desc + (size_t)j * points2.dim, (int)points1.dim);
            if (matchAll && std::sqrt(dist) < threshold) {                                      // :297-302: pushes `match`, not j
                if (sym) { out_a.push_back((uint32_t)i); out_b.push_back((uint32_t)match); }
                else { out_a.push_back((uint32_t)match); out_b.push_back((uint32_t)i); }
            } else {
                if (dist < d1) { d2 = d1; d1 = dist; match = j; }                               // :303-313
                else if (dist < d2) { d2 = dist; }
            }
        }
        if (matchAll) continue;                                                                 // :318
        if ((std::sqrt(d1 / d2) < dist2second || (d2 == FLT_MAX)) && (std::sqrt(d1) < threshold)) {   // :320-321
            if (sym) { out_a.push_back((uint32_t)i); out_b.push_back((uint32_t)match); }        // make_pair(i, match)
            else { out_a.push_back((uint32_t)match); out_b.push_back((uint32_t)i); }            // make_pair(match, i)
        }
    }
}

} // namespace

extern "C" {

void frogo_match_set_threads(int n) { omp_set_num_threads(n); }
int frogo_match_get_max_threads(void) { return omp_get_max_threads(); }

// main's pair loop, match.cpp:638-660: jobs in parallel (dynamic schedule), each job =
// ComputeMatches(first, second) [+ the sym direction appended].  Output as frog_matcher_run.
int frogo_match_run(const frog_keypoints *images, uint32_t n_images, const uint16_t *first, const uint16_t *second,
                    size_t n_jobs, const frog_match_options *o, uint64_t *offset, uint32_t **p_first, uint32_t **p_second)
{
    std::vector<std::vector<uint32_t>> a(n_jobs), b(n_jobs);
    #pragma omp parallel for schedule(dynamic)
    for (long k = 0; k < (long)n_jobs; k++) {
        const frog_keypoints &A = images[first[k]], &B = images[second[k]];
        compute_matches(A, B, o->threshold, o->dist2second, o->all != 0, o->anat, false, a[k], b[k]);
        if (o->sym) compute_matches(B, A, o->threshold, o->dist2second, o->all != 0, o->anat, true, a[k], b[k]);
    }
    offset[0] = 0;
    for (size_t k = 0; k < n_jobs; k++) offset[k + 1] = offset[k] + a[k].size();
    *p_first = (uint32_t *)std::malloc(std::max<size_t>(1, offset[n_jobs]) * sizeof(uint32_t));
    *p_second = (uint32_t *)std::malloc(std::max<size_t>(1, offset[n_jobs]) * sizeof(uint32_t));
    if (!*p_first || !*p_second) return 5;
    for (size_t k = 0; k < n_jobs; k++) {
        if (a[k].empty()) continue;
        std::memcpy(*p_first + offset[k], a[k].data(), a[k].size() * sizeof(uint32_t));
        std::memcpy(*p_second + offset[k], b[k].data(), b[k].size() * sizeof(uint32_t));
    }
    return 0;
}

void frogo_match_free(void *p) { std::free(p); }

}
