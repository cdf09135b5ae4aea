// ref_match_api.cpp -- C ABI around the REFERENCE's own ComputeMatches (match/match.cpp:255-336) and its scalar `norm`
// (match.cpp:243-251), over the reference's own `struct Point` (match.cpp:28-48).
//
// TEST INFRASTRUCTURE.  This file contains no reference code.  match.cpp as a whole cannot be built here (its main needs
// boost::filesystem / iostreams and the VTK transform reader), but these three pieces use the standard library only, so
// oracle/Makefile (target `ref`) cuts exactly them out of /root/reference/match/match.cpp where it lies -- the type block from
// `using namespace std;` to `typedef vector< Point > Points;`, the LAST definition of `norm` (the scalar one: the build's
// default, USE_SSE_FOR_MATCHING is OFF in match/CMakeLists.txt:4) and `ComputeMatches` -- into temporary files outside the
// repository, passes their paths as REF_MATCH_TYPES / REF_MATCH_FUNCS, compiles this wrapper around them with the reference's
// own tools/pointIdType.h on the include path and -DINT_PTIDS (the top-level CMakeLists.txt:11 default) into
// oracle/_ref/libfrog_refmatch.so, and deletes the temporary files.  Only the .so stays (git-ignored; it travels to the GPU
// box like any other built .so); no reference text enters the tree.
//
// Purpose: pin SURVEY.md 8(f) row 2 -- the pairing stage -- on the reference's own code: tests/test_match_oracle_ref.py (the
// oracle's pair lists against this build, and the committed fixture generated from it) and tests/test_gpu_match.py (the
// device's pair lists against this build directly).

#if !defined(REF_MATCH_TYPES) || !defined(REF_MATCH_FUNCS)
#error "REF_MATCH_TYPES / REF_MATCH_FUNCS must name the files holding the sliced definitions (see oracle/Makefile)"
#endif

#include <array>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "pointIdType.h"            // the reference's own header, found through -I$(REF)/tools at build time
#include REF_MATCH_TYPES
#include REF_MATCH_FUNCS

namespace {

Points *build(uint32_t n, uint32_t dim, const float *xyz, const float *scale, const float *laplacian, const float *desc)
{
    Points *pts = new Points(n);
    for (uint32_t i = 0; i < n; i++) {
        Point &p = (*pts)[i];
        p.desc.assign(desc + (size_t)i * dim, desc + (size_t)(i + 1) * dim);
        for (int k = 0; k < 3; k++) p.coordinates[k] = p.transformedCoordinates[k] = xyz[3 * (size_t)i + k];
        p.scale = scale[i];
        p.laplacianSign = laplacian[i];
        p.response = 0.0f;
    }
    return pts;
}

}  // namespace

// One call of ComputeMatches(points2 = image "first" (candidates), points1 = image "second" (queries), ...), as main makes it
// (match.cpp:642, :644).  Pairs are written as (pair.first, pair.second) into out_a / out_b up to `capacity`; the return value
// is the number of pairs the reference produced (compare with capacity).
extern "C" long refmatch_compute(uint32_t n2, const float *xyz2, const float *scale2, const float *lap2, const float *desc2,
                                 uint32_t n1, const float *xyz1, const float *scale1, const float *lap1, const float *desc1,
                                 uint32_t dim, float threshold, float dist2second, int matchAll, float anatVal, int sym,
                                 uint32_t *out_a, uint32_t *out_b, long capacity)
{
    Points *p2 = build(n2, dim, xyz2, scale2, lap2, desc2);
    Points *p1 = build(n1, dim, xyz1, scale1, lap1, desc1);
    MatchVect *m = ComputeMatches(*p2, *p1, threshold, dist2second, matchAll != 0, anatVal, sym != 0);
    const long n = (long)m->size();
    for (long k = 0; k < n && k < capacity; k++) { out_a[k] = (uint32_t)(*m)[k].first; out_b[k] = (uint32_t)(*m)[k].second; }
    delete m; delete p1; delete p2;
    return n;
}

// the reference's scalar norm on two descriptors (squared L2, f32, summed in dimension order)
extern "C" float refmatch_norm(const float *a, const float *b, int dim)
{
    Descriptor da(a, a + dim), db(b, b + dim);
    return norm(da, db, dim);
}

extern "C" int refmatch_sizeof_point_id(void) { return (int)sizeof(pointIdType); }
