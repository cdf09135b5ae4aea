// match.hip -- keypoint pairing on MI355X (include/frog_match.h).
//
// Two ways to the same pair lists.  (1) The exact vector kernel, match_kernel: one thread = one query
// keypoint, its descriptor in registers; every lane of a wavefront meets the SAME candidate at the same
// time; a (query, candidate) distance costs 3 vector instructions per dimension, formed exactly as the
// reference's scalar sum, term by term in dimension order.  (2) The matrix cores as a FILTER with a proven
// error bound (match_mfma_kernel / match_scan_kernel, further down): a |q|^2 - 2qc + |c|^2 product rounds
// differently, so it never decides a pair -- it only tells which 32 of a query's ~3 400 candidates can be
// among its two nearest, and those get the arithmetic of (1).  (2) is the default for descriptors of up
// to 64 finite values; (1) runs otherwise and under FROG_MATCH_VALU=1.
//
// Exactness (index work, bit-exact): sub / mul / add in dimension order without contraction;
// (d1, d2, match) updated per candidate; candidates are split into contiguous ranges over
// blockIdx.y and the partial (d1, d2, match) triples are merged, which gives the sequential result (d1 = minimum, first index attaining it;
// d2 = second smallest of the multiset).  The scale-ratio test `s1/s2 > 1.3 || s2/s1 > 1.3`
// (f32 divisions compared with the double 1.3, match.cpp:273-275) is monotone in the query
// scale, so it is turned on the host into an exact open interval (lo, hi) per candidate.
//
// Work avoided, not approximated: every image is kept sorted by (Laplacian sign, scale), so
// the 256 queries of a block share their sign and a narrow scale window, and the candidates
// that can pass the two filters form ONE contiguous index range (lo and hi grow with the
// scale), found by binary search per block.  Only that range is streamed (about 1/5 of the
// candidates for surf3d-like scales); the per-candidate tests inside stay the exact ones, so
// the range only has to be a superset.  Sorting changes the scan order, hence the rule
// "first index attaining the minimum" is kept explicitly: ties on d1 go to the smaller
// ORIGINAL index.
#include <hip/hip_runtime.h>

#include "frog_match.h"
#include "../common/usable_cpus.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

extern "C" const char *frog_last_error(void);
namespace frog { void set_last_error(const std::string &s); }

namespace {

constexpr int MATCH_BLOCK = 256;        // queries per block
// statistics: n_dist[0] pairs passing the filters (MFMA path: one add per query block); spread counters from STAT_BASE:
// [0, S) exact distances of the scan kernel, [S, 2S) pairs computed (tiles), [2S, 3S) distances evaluated by the vector kernel
constexpr int STAT_BASE = 4, STAT_SLOTS = 256;

struct DevImage {
    uint32_t n = 0;
    float *desc = nullptr;              // [n][dp], zero padded
    float *sign = nullptr, *scale = nullptr, *lo = nullptr, *hi = nullptr;
    float *xyz = nullptr;               // [n][3]
    float *norm = nullptr;              // |descriptor|^2 (f64 sum rounded to f32), for the MFMA filter
    float *mf = nullptr;                // extended vectors in MFMA operand order: [ceil(n/32)][(dp+2)/2][64]
    uint16_t *mfa16 = nullptr, *mfb16 = nullptr;   // bf16 (hi, lo) splits of the extended vectors in the operand order of
                                        // v_mfma_f32_32x32x16_bf16, candidate form and query form: [ceil(n/32)][steps16][64][8]
    bool bf16_ok = false;               // every entry splits without overflow (|v| < 2^126): the bf16 filter's bound holds
    float norm_max = 0.f;
    bool finite = true;                 // every descriptor value is finite
    uint32_t *orig = nullptr;           // sorted position -> index in the caller's order
    uint32_t *pos = nullptr;            // index in the caller's order -> sorted position (matchAll walks the caller's order)
    std::vector<uint32_t> h_orig;
};

struct Partial { float d1, d2; int j; };

struct MatchArgs {
    const float *q_desc, *q_sign, *q_scale, *q_xyz;
    const float *c_desc, *c_sign, *c_lo, *c_hi, *c_xyz;
    const uint32_t *c_orig;
    uint32_t nq, nc, splits;
    uint32_t hmax_stride;               // entries per query of the half-tile maxima: 2 * tiles of the candidate image
    float anat;
    float mf_eps;                       // the matrix-core filter's error bound in use (MF_EPS or MF_EPS_BF16), relative to |q|^2 + |c|^2
    Partial *partial;                   // [splits][nq]
    unsigned long long *n_dist;         // distances evaluated (statistics)
};

// The candidates the queries of one block can pass (a superset): images are sorted by (sign, scale).
__global__ void match_range_kernel(const MatchArgs a, uint32_t q_blocks, uint2 *ranges)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= q_blocks) return;
    const uint32_t q0 = b * MATCH_BLOCK, q1 = min(a.nq, q0 + MATCH_BLOCK) - 1;
    uint32_t cb = 0, ce = a.nc;
    const float sg = a.q_sign[q0];
    if (sg == a.q_sign[q1]) {                          // the block has one sign
        const float s_min = a.q_scale[q0], s_max = a.q_scale[q1];
        uint32_t lo = 0, hi = a.nc;                    // first candidate with sign >= sg
        while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (a.c_sign[mid] < sg) lo = mid + 1; else hi = mid; }
        const uint32_t sb = lo;
        hi = a.nc;                                     // first candidate with sign > sg
        while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (a.c_sign[mid] <= sg) lo = mid + 1; else hi = mid; }
        const uint32_t se = lo;
        lo = sb; hi = se;                              // first candidate whose upper bound exceeds the smallest query scale
        while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (!(a.c_hi[mid] > s_min)) lo = mid + 1; else hi = mid; }
        cb = lo;
        hi = se;                                       // first candidate whose lower bound reaches the largest query scale
        while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (a.c_lo[mid] < s_max) lo = mid + 1; else hi = mid; }
        ce = lo;
    }
    ranges[b] = make_uint2(cb, ce);
}

constexpr int CAND_TILE = 64;           // candidates staged per LDS tile (D must be a multiple of 8)

template <int D>
__global__ __launch_bounds__(MATCH_BLOCK) void match_kernel(const MatchArgs a, const uint2 *ranges)
{
    // Every lane of a wavefront meets the SAME candidate at the same time: the candidate tile is
    // staged in LDS with coalesced loads and read back with broadcast ds_read_b128 (all lanes one
    // address).  The reads of one half descriptor are issued while the other half is being
    // subtracted / squared / summed (register arrays A / B, branch-free body), so their latency
    // overlaps arithmetic instead of sitting in front of every group of four dimensions -- which is
    // what a plain loop compiles to, several times slower.
    __shared__ float4 tile[CAND_TILE][D / 4];
    __shared__ float cs[CAND_TILE], clo[CAND_TILE], chi[CAND_TILE], cx[CAND_TILE], cy[CAND_TILE], cz[CAND_TILE];
    __shared__ int co[CAND_TILE];
    const uint32_t qi = blockIdx.x * MATCH_BLOCK + threadIdx.x;
    const bool valid = qi < a.nq;
    float q[D];
    float qsign = 0.f, qscale = 0.f, qx = 0.f, qy = 0.f, qz = 0.f;
    if (valid) {
        const float4 *row = reinterpret_cast<const float4 *>(a.q_desc + (size_t)qi * D);
        #pragma unroll
        for (int k = 0; k < D / 4; k++) { const float4 v = row[k]; q[4 * k] = v.x; q[4 * k + 1] = v.y; q[4 * k + 2] = v.z; q[4 * k + 3] = v.w; }
        qsign = a.q_sign[qi]; qscale = a.q_scale[qi];
        qx = a.q_xyz[3 * (size_t)qi]; qy = a.q_xyz[3 * (size_t)qi + 1]; qz = a.q_xyz[3 * (size_t)qi + 2];
    } else {
        #pragma unroll
        for (int k = 0; k < D; k++) q[k] = 0.f;
    }
    float d1 = FLT_MAX, d2 = FLT_MAX;
    int match = -1;
    unsigned int evaluated = 0, computed = 0;
    const bool use_anat = a.anat != 0.f;

    // the filters and the (d1, d2, match) update of one candidate (match.cpp:270-313), as selects
    auto finish = [&](float dist, uint32_t c) {
        bool pass = valid && qsign == cs[c]                             // match.cpp:270
                          && qscale > clo[c] && qscale < chi[c];        // :273-275 as an exact interval
        if (use_anat) {                                                 // :278-291
            const float ex = qx - cx[c], ey = qy - cy[c], ez = qz - cz[c];
            const float eucl = sqrtf(ex * ex + ey * ey + ez * ez);
            pass = pass && !(eucl > a.anat);
        }
        const int orig = co[c];
        const bool better = pass && dist < d1;                          // :303-313
        const bool second = pass && !better && dist < d2;
        const bool tie = pass && !better && dist == d1 && orig < match; // upstream scans in original order: first index wins
        d2 = better ? d1 : (second ? dist : d2);
        d1 = better ? dist : d1;
        match = (better || tie) ? orig : match;
        evaluated += pass ? 1u : 0u;
    };

    // this block's share of the candidates its queries can pass
    const uint2 rg = ranges[blockIdx.x];
    const uint32_t per = (rg.y - rg.x + a.splits - 1) / a.splits;
    const uint32_t c_begin = min(rg.y, rg.x + blockIdx.y * per), c_end = min(rg.y, c_begin + per);
    constexpr int HQ = D / 8;                           // float4 registers per half descriptor
    for (uint32_t base = c_begin; base < c_end; base += CAND_TILE) {
        const uint32_t cnt = min((uint32_t)CAND_TILE, c_end - base);
        __syncthreads();
        {   // stage the tile: cnt rows of D floats, contiguous in memory
            const float4 *src = reinterpret_cast<const float4 *>(a.c_desc + (size_t)base * D);
            float4 *dst = &tile[0][0];
            for (uint32_t k = threadIdx.x; k < cnt * (D / 4); k += MATCH_BLOCK) dst[k] = src[k];
            if (threadIdx.x < cnt) {
                const uint32_t c = base + threadIdx.x;
                cs[threadIdx.x] = a.c_sign[c]; clo[threadIdx.x] = a.c_lo[c]; chi[threadIdx.x] = a.c_hi[c];
                co[threadIdx.x] = (int)a.c_orig[c];
                cx[threadIdx.x] = a.c_xyz[3 * (size_t)c]; cy[threadIdx.x] = a.c_xyz[3 * (size_t)c + 1]; cz[threadIdx.x] = a.c_xyz[3 * (size_t)c + 2];
            }
        }
        __syncthreads();
        computed += cnt;
        // A = first half of a descriptor, B = second half; each is read while the other is used
        float4 A[HQ], B[HQ];
        #pragma unroll
        for (int k = 0; k < HQ; k++) A[k] = tile[0][k];
        for (uint32_t c = 0; c < cnt; c++) {
            #pragma unroll
            for (int k = 0; k < HQ; k++) B[k] = tile[c][HQ + k];        // in flight during the first half
            float dist = 0.f;                                           // norm, match.cpp:242-251
            #pragma unroll
            for (int k = 0; k < HQ; k++) {
                float t;
                t = q[4 * k] - A[k].x;     dist += t * t;
                t = q[4 * k + 1] - A[k].y; dist += t * t;
                t = q[4 * k + 2] - A[k].z; dist += t * t;
                t = q[4 * k + 3] - A[k].w; dist += t * t;
            }
            const uint32_t cn = min(c + 1, cnt - 1);
            #pragma unroll
            for (int k = 0; k < HQ; k++) A[k] = tile[cn][k];            // in flight during the second half
            #pragma unroll
            for (int k = 0; k < HQ; k++) {
                float t;
                t = q[4 * (HQ + k)] - B[k].x;     dist += t * t;
                t = q[4 * (HQ + k) + 1] - B[k].y; dist += t * t;
                t = q[4 * (HQ + k) + 2] - B[k].z; dist += t * t;
                t = q[4 * (HQ + k) + 3] - B[k].w; dist += t * t;
            }
            finish(dist, c);
        }
    }
    if (valid) a.partial[(size_t)blockIdx.y * a.nq + qi] = Partial{ d1, d2, match };
    // statistics: one atomic per wavefront
    unsigned int total = evaluated;
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
    // spread over STAT_SLOTS counters: atomics on one address serialise in L2
    if ((threadIdx.x & 63) == 0 && total) atomicAdd(a.n_dist + STAT_BASE + 2 * STAT_SLOTS + (blockIdx.x + blockIdx.y) % STAT_SLOTS, (unsigned long long)total);
    if ((threadIdx.x & 63) == 0 && computed) atomicAdd(a.n_dist + STAT_BASE + STAT_SLOTS + (blockIdx.x + blockIdx.y) % STAT_SLOTS, 64ull * computed);
}

// merge the candidate ranges in order, then the acceptance test (match.cpp:320-321).
// out[q]: accepted -> candidate index (>= 0), or -2 when no candidate had a distance below
// FLT_MAX (upstream then emits its `match` variable, which is declared outside the query loop
// and still holds the previous queries' value: resolved on the host, which walks the queries
// in order); rejected -> -1 without candidate, -(index + 3) with one (it updates `match` too).
__global__ void match_decide_kernel(const Partial *partial, uint32_t nq, uint32_t splits, float threshold,
                                    float dist2second, int *out)
{
    const uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    float d1 = FLT_MAX, d2 = FLT_MAX;
    int match = -1;
    for (uint32_t s = 0; s < splits; s++) {
        const Partial p = partial[(size_t)s * nq + qi];
        // a range's values enter as a sequential scan would see them: its minimum first (ties on
        // the minimum go to the smaller original index), then its second
        if (p.d1 < d1) { d2 = d1; d1 = p.d1; match = p.j; }
        else {
            if (p.d1 < d2) d2 = p.d1;
            if (p.d1 == d1 && p.j >= 0 && p.j < match) match = p.j;
        }
        if (p.d2 < d2) d2 = p.d2;
    }
    const bool ok = (sqrtf(d1 / d2) < dist2second || d2 == FLT_MAX) && (sqrtf(d1) < threshold);
    out[qi] = ok ? (match >= 0 ? match : -2) : (match >= 0 ? -(match + 3) : -1);
}


// ---- the same result through the matrix cores ---------------------------------------------------
// The distances that decide a query's outcome are its two smallest; everything else only has to be
// known to be larger.  Every keypoint is also stored as an EXTENDED vector (p, -|p|^2/2, 1): with the last two
// entries of one side swapped the product of two of them is q.c - |q|^2/2 - |c|^2/2 = -d^2/2, so ONE f32 MFMA
// chain (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain, error bounded, MF_EPS below) yields an approximate
// -d^2/2 for 32 x 32 (candidate, query) pairs, and the same array serves an image as query and as
// candidate: it is kept in the operand's own lane order ([group of 32 points][step][lane]), so both
// operands are plain coalesced loads -- no LDS, no barrier, a wavefront is on its own.
//   match_mfma_kernel: the products; per query and HALF TILE (the 16 candidates a lane holds of a 32 x 32 product)
//                      the largest product of a candidate that passes the query's filters -- one max per candidate;
//   match_scan_kernel: per query, the second largest half-tile maximum minus the error bound is a threshold no
//                      candidate among the two nearest can be below; the few half tiles that reach it (two per
//                      query, typically) get the reference's own arithmetic (sequential sum of (q_k - c_k)^2,
//                      strict-< bookkeeping) on their 16 candidates, then the acceptance test.
// Candidates outside the bound provably cannot be one of the two nearest, so the pair lists are those of
// the exact kernel above, bit for bit; its 3 vector instructions per (query, candidate, dimension) become
// 1/32 of a matrix instruction plus 4 vector instructions per (query, candidate), and 32 exact distances
// per query remain of the ~3 400 the vector kernel evaluates.
// The sign / scale filters of a query select ONE contiguous range of the sorted candidates (see the header),
// found per query by binary search (match_qrange_kernel): the test is two integer compares.
constexpr int MF_TILE = 32;             // points per MFMA operand group
// |(-2 x product) - reference distance| <= MF_EPS * (|q|^2 + |c|^2), with u = 2^-24 and S = |q|^2 + |c|^2:
//   reference: sub, mul and D - 1 adds in f32 per term    -> |d_ref - d| <= (D + 3) u d <= 2 (D + 3) u S   (d <= 2 S)
//   product:   an fmaf chain of D + 2 steps (one rounding each) over terms whose magnitudes add up to at most S
//                                                         -> |P - P_exact| <= (D + 2) u S
//   the stored half norms are the f64 sums rounded to f32 -> |-2 P_exact - d| <= u S
// together (4 D + 11) u S = 267 u S at D = 64, 1.6e-5 S; MF_EPS = 2^-15 = 3.1e-5 leaves a factor 2.
constexpr float MF_EPS = 1.0f / 32768.0f;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct QRange { uint32_t first, last; };        // candidates [first, last) pass the query's sign and scale tests

// per query: the exact set {c : sign == qsign, lo[c] < qscale < hi[c]} as a range of the sorted candidates;
// per block of MATCH_BLOCK queries: the union of their ranges (what match_range_kernel estimates for the vector kernel)
__global__ __launch_bounds__(MATCH_BLOCK) void match_qrange_kernel(const MatchArgs a, QRange *qr, uint2 *ranges)
{
    __shared__ uint32_t lo_s, hi_s, n_s;
    if (threadIdx.x == 0) { lo_s = 0xFFFFFFFFu; hi_s = 0u; n_s = 0u; }
    __syncthreads();
    const uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi < a.nq) {
    const float sg = a.q_sign[qi], sc = a.q_scale[qi];
    // Two searches over the WHOLE sorted array with the sign folded into the predicate, side by side -- 15 dependent steps of
    // four loads each, where the sign segment was searched first and the two bounds inside it one after the other: 60
    // dependent loads, the whole of this kernel's 23 us.  (The candidates are sorted by (sign, scale); lo and hi grow with the
    // scale.)   first = first c with (sign, hi) beyond (sg, sc];  last = first c with (sign, lo) at or beyond (sg, sc) -- never
    // before `first`, since lo[c] < hi[c].
    uint32_t lo1 = 0, hi1 = a.nc, lo2 = 0, hi2 = a.nc;
    while (lo1 < hi1 || lo2 < hi2) {
        const uint32_t m1 = min((lo1 + hi1) / 2, a.nc - 1), m2 = min((lo2 + hi2) / 2, a.nc - 1);
        const float s1 = a.c_sign[m1], h1 = a.c_hi[m1], s2 = a.c_sign[m2], l2 = a.c_lo[m2];
        if (lo1 < hi1) { if (s1 > sg || (s1 == sg && h1 > sc)) hi1 = m1; else lo1 = m1 + 1; }
        if (lo2 < hi2) { if (s2 > sg || (s2 == sg && !(l2 < sc))) hi2 = m2; else lo2 = m2 + 1; }
    }
    const uint32_t first = lo1;
    const uint32_t lo = max(lo1, lo2);
    qr[qi] = QRange{ first, lo };
    if (lo > first) { atomicMin(&lo_s, first); atomicMax(&hi_s, lo); atomicAdd(&n_s, lo - first); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        ranges[blockIdx.x] = lo_s <= hi_s ? make_uint2(lo_s, hi_s) : make_uint2(0u, 0u);
        if (n_s) atomicAdd(a.n_dist, (unsigned long long)n_s);      // (query, candidate) pairs that pass the sign and scale tests
    }
}


// hmax[query][2 * tile + half] = the largest product among the 16 candidates of that half tile that pass the query's
// filters (-inf if none); blockIdx.y splits the candidate tiles of the query block's range.
template <int D, bool ANAT, int G>
__global__ __launch_bounds__(MATCH_BLOCK * 2 / G) void match_mfma_kernel(const MatchArgs a, const uint2 *ranges, const QRange *qr,
                                                                 const float *q_mf, const float *c_mf, float *hmax)
{
    constexpr int STEPS = (D + 2) / 2;              // MFMA instructions per 32 x 32 tile (K = 2 each)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    // a wavefront multiplies every candidate tile it loads with G groups of 32 queries (g0 .. g0 + G - 1); the block's
    // MATCH_BLOCK / (32 G) wavefronts cover the MATCH_BLOCK queries that share a candidate range
    const uint32_t g0 = (blockIdx.x * MATCH_BLOCK + wave * 32 * G) / MF_TILE;
    const uint32_t n_qgroups = (a.nq + MF_TILE - 1) / MF_TILE;
    if (g0 >= n_qgroups) return;

    float b[G][STEPS], qx[G], qy[G], qz[G];
    uint32_t qfirst[G], qcount[G], qidx[G];
    bool qvalid[G];
    #pragma unroll
    for (int g = 0; g < G; g++) {
        const uint32_t grp = min(g0 + g, n_qgroups - 1);
        const float *src = q_mf + (size_t)grp * STEPS * 64 + lane;
        #pragma unroll
        for (int s2 = 0; s2 < STEPS - 1; s2++) b[g][s2] = src[s2 * 64];
        // the last step holds (-|p|^2/2, 1): as the B operand its two dimensions trade places, so that the product is
        // (c, -|c|^2/2, 1) . (q, 1, -|q|^2/2) = c.q - |c|^2/2 - |q|^2/2
        b[g][STEPS - 1] = q_mf[((size_t)grp * STEPS + STEPS - 1) * 64 + (lane ^ 32)];
        const uint32_t qi = (g0 + g) * MF_TILE + j;
        qidx[g] = qi;
        qvalid[g] = (g0 + g) < n_qgroups && qi < a.nq;
        const QRange r = qvalid[g] ? qr[qi] : QRange{ 0, 0 };
        qfirst[g] = r.first; qcount[g] = r.last - r.first;                      // 0 for an invalid query: nothing passes
        qx[g] = qy[g] = qz[g] = 0.f;
        if (ANAT && qvalid[g]) { qx[g] = a.q_xyz[3 * (size_t)qi]; qy[g] = a.q_xyz[3 * (size_t)qi + 1]; qz[g] = a.q_xyz[3 * (size_t)qi + 2]; }
    }

    // this block's share of the candidate tiles its queries can pass
    const uint2 rg = ranges[blockIdx.x];
    const uint32_t t_first = rg.x / MF_TILE, t_last = (rg.y + MF_TILE - 1) / MF_TILE;      // tiles [t_first, t_last)
    const uint32_t per = (t_last - t_first + a.splits - 1) / a.splits;
    const uint32_t t_begin = min(t_last, t_first + blockIdx.y * per), t_end = min(t_last, t_begin + per);
    if (t_begin >= t_end) return;

    // the candidate operand of the NEXT tile is loaded while the current one is multiplied (a tile is 25 coalesced
    // loads that the 50 matrix instructions would otherwise wait for)
    float nx[STEPS];
    {
        const float *src = c_mf + (size_t)t_begin * STEPS * 64 + lane;
        #pragma unroll
        for (int s2 = 0; s2 < STEPS; s2++) nx[s2] = src[s2 * 64];
    }
    for (uint32_t t = t_begin; t < t_end; t++) {
        float av[STEPS];
        #pragma unroll
        for (int s2 = 0; s2 < STEPS; s2++) av[s2] = nx[s2];
        {
            const float *src = c_mf + (size_t)min(t + 1, t_last - 1) * STEPS * 64 + lane;
            #pragma unroll
            for (int s2 = 0; s2 < STEPS; s2++) nx[s2] = src[s2 * 64];
        }
        f32x16 acc[G];
        #pragma unroll
        for (int g = 0; g < G; g++)
            #pragma unroll
            for (int r = 0; r < 16; r++) acc[g][r] = 0.f;
        #pragma unroll
        for (int s2 = 0; s2 < STEPS; s2++)
            #pragma unroll
            for (int g = 0; g < G; g++) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s2], b[g][s2], acc[g], 0, 0, 0);
        const uint32_t base = t * MF_TILE + 4 * h;
        #pragma unroll
        for (int g = 0; g < G; g++) {
            float hm = -INFINITY;
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t c = base + (uint32_t)((r & 3) + 8 * (r >> 2));   // row of the product = candidate
                bool pass = (c - qfirst[g]) < qcount[g];                        // sign and scale tests, match.cpp:270-275
                if (ANAT) {                                                     // :278-291
                    const uint32_t cc = min(c, a.nc - 1);
                    const float ex = qx[g] - a.c_xyz[3 * (size_t)cc], ey = qy[g] - a.c_xyz[3 * (size_t)cc + 1], ez = qz[g] - a.c_xyz[3 * (size_t)cc + 2];
                    pass = pass && !(sqrtf(ex * ex + ey * ey + ez * ez) > a.anat);
                }
                hm = fmaxf(hm, pass ? acc[g][r] : -INFINITY);
            }
            if (qvalid[g]) hmax[(size_t)qidx[g] * a.hmax_stride + t * 2 + h] = hm;  // read back 16 entries at a time by the query's team
        }
    }
    // statistics go to one of STAT_SLOTS counters: thousands of atomics on ONE address serialise in L2 (20 000 of
    // them, one per query, cost the scan kernel 60 of its 70 us)
    if (lane == 0)
        atomicAdd(a.n_dist + STAT_BASE + STAT_SLOTS + (blockIdx.x + blockIdx.y) % STAT_SLOTS, 64ull * MF_TILE * (t_end - t_begin));
}

// ---- the same filter on the bf16 matrix cores (16 x the f32 rate) -------------------------------------------------------
// Every entry v of an extended vector is split  v = vh + vl + vr,  vh = bf16(v), vl = bf16(v - vh) (round to nearest even;
// v - vh is exact in f32), |vl| <= 2^-8 |v|, |vr| <= 2^-16 |v|.  With DE = D + 2 entries per vector the candidate operand is
// the K-vector [ch | cl | ch] and the query operand [qh' | qh' | ql'] (q' = the query's vector with its last two entries
// exchanged, as in the f32 form), 3 DE products per pair:  sum_k (ch qh' + cl qh' + ch ql')_k  -- what is left out per entry is
// cl ql' + cr q' + c qr' - cr qr', at most 3.0001 * 2^-16 |c_k| |q'_k| (0 for the entries that are 1: 1 = bf16(1)).
// Error of the product P against the real -d^2/2, S = |q|^2 + |c|^2, u = 2^-24:
//   splits        sum_k |c_k q_k| <= S / 2 over the D descriptor entries: 1.5 * 2^-16 S; the two half norms: 2^-16 S / 2
//                                                                                              -> 2.0 * 2^-16 S = 3.05e-5 S
//   accumulation  the products of two bf16 values are exact in f32; each of the K3 = 3 DE of them (160 or 208 with padding) is
//                 taken to enter the f32 accumulation with an error of at most 2 u (truncating adders allowed for) of the sum
//                 of the magnitudes so far, at most 1.01 S: 2 u K3 1.01 S                       -> 1.9e-5 S (2.5e-5 at D = 64)
//                 (an assumption about v_mfma_f32_32x32x16_bf16's internal adder, which the ISA document does not describe;
//                 tests/test_gpu_match.py::test_bf16_filter_products_within_the_bound measures the products against f64
//                 for random, cancelling and large-norm vectors: observed maximum in DESIGN.md section 10)
//   stored norms  f64 sums rounded to f32: u S / 2; the reference distance's own roundings: 2 (D + 3) u S in d, half in P
// together |P - (-d_ref / 2)| <= 5.3e-5 S (5.9e-5 at D = 64), i.e. |-2 P - d_ref| <= 1.2e-4 S; MF_EPS_BF16 = 2^-12 = 2.44e-4
// leaves a factor 2.  A wider bound only lets more half tiles through to the exact verification (typically still two per query:
// 2.4e-4 S is far below the gaps between a query's nearest descriptors); it never changes a pair.
// A block stages the candidate tile (10 or 13 KB) in LDS once for its four wavefronts (double buffered, one barrier per
// tile): at this rate each wavefront reading its own copy through the L1 would need twice the L1's 64 B per clock.
constexpr float MF_EPS_BF16 = 1.0f / 4096.0f;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__host__ __device__ constexpr int mf16_steps(int D) { return (3 * (D + 2) + 15) / 16; }

template <int D, bool ANAT>
__global__ __launch_bounds__(MATCH_BLOCK) void match_mfma16_kernel(const MatchArgs a, const uint2 *ranges, const QRange *qr,
                                                                   const uint4 *q_mfb, const uint4 *c_mfa, float *hmax)
{
    constexpr int G = 2;                            // query groups of 32 per wavefront
    constexpr int STEPS = mf16_steps(D);            // matrix instructions per 32 x 32 tile (K = 16 each)
    constexpr int TILE16 = STEPS * 64;              // uint4 per operand tile
    __shared__ uint4 ctile[2][TILE16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t g0 = (blockIdx.x * MATCH_BLOCK + wave * 32 * G) / MF_TILE;
    const uint32_t n_qgroups = (a.nq + MF_TILE - 1) / MF_TILE;
    // (no early return: every wavefront of the block takes part in the staging and its barriers)
    uint4 b[G][STEPS];
    float qx[G], qy[G], qz[G];
    uint32_t qfirst[G], qcount[G], qidx[G];
    bool qvalid[G];
    #pragma unroll
    for (int g = 0; g < G; g++) {
        const uint32_t grp = min(g0 + g, n_qgroups - 1);
        #pragma unroll
        for (int s2 = 0; s2 < STEPS; s2++) b[g][s2] = q_mfb[((size_t)grp * STEPS + s2) * 64 + lane];
        const uint32_t qi = (g0 + g) * MF_TILE + j;
        qidx[g] = qi;
        qvalid[g] = (g0 + g) < n_qgroups && qi < a.nq;
        const QRange r = qvalid[g] ? qr[qi] : QRange{ 0, 0 };
        qfirst[g] = r.first; qcount[g] = r.last - r.first;
        qx[g] = qy[g] = qz[g] = 0.f;
        if (ANAT && qvalid[g]) { qx[g] = a.q_xyz[3 * (size_t)qi]; qy[g] = a.q_xyz[3 * (size_t)qi + 1]; qz[g] = a.q_xyz[3 * (size_t)qi + 2]; }
    }
    const uint2 rg = ranges[blockIdx.x];
    const uint32_t t_first = rg.x / MF_TILE, t_last = (rg.y + MF_TILE - 1) / MF_TILE;
    const uint32_t per = (t_last - t_first + a.splits - 1) / a.splits;
    const uint32_t t_begin = min(t_last, t_first + blockIdx.y * per), t_end = min(t_last, t_begin + per);
    if (t_begin >= t_end) return;                   // block-uniform

    constexpr int PER_THREAD = (TILE16 + MATCH_BLOCK - 1) / MATCH_BLOCK;
    uint4 stage[PER_THREAD];
    auto fetch = [&](uint32_t t) {
        const uint4 *src = c_mfa + (size_t)t * TILE16;
        #pragma unroll
        for (int k = 0; k < PER_THREAD; k++) {
            const int e = (int)threadIdx.x + k * MATCH_BLOCK;
            stage[k] = e < TILE16 ? src[e] : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto park = [&](int buf) {
        #pragma unroll
        for (int k = 0; k < PER_THREAD; k++) {
            const int e = (int)threadIdx.x + k * MATCH_BLOCK;
            if (e < TILE16) ctile[buf][e] = stage[k];
        }
    };
    fetch(t_begin);
    park(0);
    __syncthreads();
    for (uint32_t t = t_begin; t < t_end; t++) {
        const int buf = (int)((t - t_begin) & 1u);
        if (t + 1 < t_end) fetch(t + 1);            // in flight while this tile is multiplied
        f32x16 acc[G];
        #pragma unroll
        for (int g = 0; g < G; g++)
            #pragma unroll
            for (int r = 0; r < 16; r++) acc[g][r] = 0.f;
        #pragma unroll
        for (int s2 = 0; s2 < STEPS; s2++) {
            const uint4 av = ctile[buf][s2 * 64 + lane];
            #pragma unroll
            for (int g = 0; g < G; g++)
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, b[g][s2]), acc[g], 0, 0, 0);
        }
        const uint32_t base = t * MF_TILE + 4 * h;
        #pragma unroll
        for (int g = 0; g < G; g++) {
            float hm = -INFINITY;
            #pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t c = base + (uint32_t)((r & 3) + 8 * (r >> 2));   // row of the product = candidate
                bool pass = (c - qfirst[g]) < qcount[g];                        // sign and scale tests, match.cpp:270-275
                if (ANAT) {                                                     // :278-291
                    const uint32_t cc = min(c, a.nc - 1);
                    const float ex = qx[g] - a.c_xyz[3 * (size_t)cc], ey = qy[g] - a.c_xyz[3 * (size_t)cc + 1], ez = qz[g] - a.c_xyz[3 * (size_t)cc + 2];
                    pass = pass && !(sqrtf(ex * ex + ey * ey + ez * ez) > a.anat);
                }
                hm = fmaxf(hm, pass ? acc[g][r] : -INFINITY);
            }
            if (qvalid[g]) hmax[(size_t)qidx[g] * a.hmax_stride + t * 2 + h] = hm;
        }
        if (t + 1 < t_end) park(buf ^ 1);           // the other buffer: last read one barrier ago
        __syncthreads();
    }
    if (threadIdx.x == 0)
        atomicAdd(a.n_dist + STAT_BASE + STAT_SLOTS + (blockIdx.x + blockIdx.y) % STAT_SLOTS, 64ull * MF_TILE * (MATCH_BLOCK / 64) * (t_end - t_begin));
}

// test hook: the 32 x 32 products of one candidate group and one query group, f32 chain (form 0) or bf16 splits (form 1)
template <int D>
__global__ __launch_bounds__(64) void match_products_kernel(const float *q_mf, const float *c_mf, const uint4 *q_mfb, const uint4 *c_mfa,
                                                            int form, float *out /*[32 candidates][32 queries]*/)
{
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    f32x16 acc;
    #pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.f;
    if (form == 0) {
        constexpr int STEPS = (D + 2) / 2;
        #pragma unroll
        for (int s2 = 0; s2 < STEPS; s2++) {
            const float bq = s2 < STEPS - 1 ? q_mf[s2 * 64 + lane] : q_mf[s2 * 64 + (lane ^ 32)];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(c_mf[s2 * 64 + lane], bq, acc, 0, 0, 0);
        }
    } else {
        constexpr int STEPS = mf16_steps(D);
        #pragma unroll
        for (int s2 = 0; s2 < STEPS; s2++)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c_mfa[s2 * 64 + lane]), __builtin_bit_cast(bf16x8, q_mfb[s2 * 64 + lane]), acc, 0, 0, 0);
    }
    #pragma unroll
    for (int r = 0; r < 16; r++) out[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + j] = acc[r];
}

constexpr int SCAN_BLOCK = 64;          // one wavefront = 4 query teams
constexpr int SCAN_HITS = 64;           // half tiles listed per query before the slow path

// Pass 2 without the matrix cores: a team of 16 lanes per query walks the half-tile maxima pass 1 left behind; for
// every half tile whose maximum reaches the query's threshold (a few per query) each lane of the team gives ONE of
// its 16 candidates the reference's arithmetic; the team's (d1, d2, match) partials are merged as a sequential scan
// would see them and lane 0 applies the acceptance test.  Replaces threshold + second MFMA pass + decide.
template <int D, bool ANAT>
__global__ __launch_bounds__(SCAN_BLOCK) void match_scan_kernel(const MatchArgs a, const uint2 *ranges, const QRange *qr,
                                                         const float *q_norm, float c_norm_max, const float *hmax,
                                                         float threshold, float dist2second, int *out)
{
    const int lane = threadIdx.x & 63, tl = lane & 15;              // team = 16 consecutive lanes
    const uint32_t qi = blockIdx.x * (SCAN_BLOCK / 16) + (threadIdx.x >> 4);
    const bool valid = qi < a.nq;
    const uint32_t qc = min(qi, a.nq - 1);
    const QRange r = valid ? qr[qc] : QRange{ 0, 0 };
    const uint2 rg = ranges[qc / MATCH_BLOCK];
    const uint32_t t_first = rg.x / MF_TILE, t_last = (rg.y + MF_TILE - 1) / MF_TILE;
    const uint32_t n_entries = valid ? (t_last - t_first) * 2 : 0;
    const float *hrow = hmax + (size_t)qc * a.hmax_stride + t_first * 2;
    // the two largest half-tile maxima (the second can only be smaller than the true second largest product, when both
    // sit in one half tile: a lower threshold, more candidates verified, never fewer)
    float t1 = -INFINITY, t2 = -INFINITY;
    for (uint32_t e0 = 0; e0 < n_entries; e0 += 64) {
        float v[4];
        #pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t e = e0 + 16 * k + tl; v[k] = e < n_entries ? hrow[e] : -INFINITY; }
        #pragma unroll
        for (int k = 0; k < 4; k++) { t2 = __builtin_amdgcn_fmed3f(t1, t2, v[k]); t1 = fmaxf(t1, v[k]); }
    }
    #pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
        const float o1 = __shfl_xor(t1, off, 64), o2 = __shfl_xor(t2, off, 64);
        const float lo1 = fminf(t1, o1);
        t1 = fmaxf(t1, o1);
        t2 = fmaxf(lo1, fmaxf(t2, o2));
    }
    // -2 x product approximates the distance to MF_EPS * (|q|^2 + |c|^2): every candidate whose exact distance is at most
    // the second smallest exact distance has a product >= (second largest product) - MF_EPS * (|q|^2 + max |c|^2)
    // (1e-30: products of bf16 splits below the normal range may be flushed; nothing for descriptors of any real size)
    const float thr = t2 - a.mf_eps * (q_norm[qc] + c_norm_max) - 1e-30f;      // -inf when fewer than two half tiles hold a candidate
    // per team: its 16 candidate rows, D/4 + 1 float4 apart: the odd stride spreads the rows over all banks
    constexpr int ROW4 = D / 4 + 1;
    __shared__ float4 rows[SCAN_BLOCK / 16][16 * ROW4];
    const int team = threadIdx.x >> 4;
    float4 q4[D / 4];                                               // the query's descriptor, kept for all its hits
    {
        const float4 *qrow = reinterpret_cast<const float4 *>(a.q_desc + (size_t)qc * D);
        #pragma unroll
        for (int k = 0; k < D / 4; k++) q4[k] = qrow[k];
    }
    float d1 = FLT_MAX, d2 = FLT_MAX;
    int match = -1;
    unsigned int verified = 0;
    // one verification: the half tile `eh` of this team's query -- 16 candidates, one per lane
    auto verify = [&](uint32_t eh) {
        const uint32_t row0 = (t_first + eh / 2) * MF_TILE + 4 * (eh & 1);      // the half tile: rows row0 + {0..3} + 8 * {0..3}
        const uint32_t c = row0 + (uint32_t)((tl & 3) + 8 * (tl >> 2));
        // the team fetches its 16 candidate rows together: 4 runs of 4 consecutive rows, 16 lanes x 16 bytes per load
        #pragma unroll
        for (int k = 0; k < D / 4; k++) {
            const int idx = k * 16 + tl, run = idx / (D), off = idx % (D);          // float4 units: D per run of 4 rows
            const uint32_t src_row = min(row0 + 8u * (uint32_t)run, a.nc - 1);
            const float4 *src = reinterpret_cast<const float4 *>(a.c_desc + (size_t)src_row * D);
            // a run of 4 rows is contiguous only while it stays inside the array: the last tile is read row by row
            const uint32_t rr = min(row0 + 8u * (uint32_t)run + (uint32_t)(off / (D / 4)), a.nc - 1);
            rows[team][(run * 4 + off / (D / 4)) * ROW4 + off % (D / 4)] =
                (row0 + 8u * (uint32_t)run + 3u < a.nc) ? src[off] : reinterpret_cast<const float4 *>(a.c_desc + (size_t)rr * D)[off % (D / 4)];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool pass = (c - r.first) < (r.last - r.first);
        if (ANAT && pass) {
            const float ex = a.q_xyz[3 * (size_t)qc] - a.c_xyz[3 * (size_t)c], ey = a.q_xyz[3 * (size_t)qc + 1] - a.c_xyz[3 * (size_t)c + 1],
                        ez = a.q_xyz[3 * (size_t)qc + 2] - a.c_xyz[3 * (size_t)c + 2];
            pass = !(sqrtf(ex * ex + ey * ey + ez * ez) > a.anat);
        }
        float dist = 0.f;                                               // norm, match.cpp:242-251: sequential f32 sum
        {
            const float4 *mine = &rows[team][((tl >> 2) * 4 + (tl & 3)) * ROW4];
            #pragma unroll
            for (int k = 0; k < D / 4; k++) {
                const float4 y = mine[k];
                float t;
                t = q4[k].x - y.x; dist += t * t;
                t = q4[k].y - y.y; dist += t * t;
                t = q4[k].z - y.z; dist += t * t;
                t = q4[k].w - y.w; dist += t * t;
            }
        }
        __builtin_amdgcn_wave_barrier();                                // the rows are overwritten by the next verification
        if (pass) {
            const int orig = (int)a.c_orig[c];
            const bool better = dist < d1;                              // match.cpp:303-313, in any scan order
            const bool second = !better && dist < d2;
            const bool tie = !better && dist == d1 && orig < match;
            d2 = better ? d1 : (second ? dist : d2);
            d1 = better ? dist : d1;
            match = (better || tie) ? orig : match;
            verified++;
        }
    };
    // The half tiles to verify are first LISTED per team, then verification k of all four teams runs together: a
    // team's hits sit in different rounds of its walk, and verifying them where they are found would run the
    // wavefront once per hit of any team with three quarters of its lanes idle.
    __shared__ unsigned short hits[SCAN_BLOCK / 16][SCAN_HITS];
    __shared__ unsigned int n_hits[SCAN_BLOCK / 16];
    if (tl == 0) n_hits[team] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t e0 = 0; e0 < n_entries; e0 += 64) {
        float v[4];
        #pragma unroll
        for (int k = 0; k < 4; k++) { const uint32_t e = e0 + 16 * k + tl; v[k] = e < n_entries ? hrow[e] : -INFINITY; }
        #pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t e = e0 + 16 * k + tl;
            if (e < n_entries && v[k] >= thr && v[k] > -INFINITY) {     // -inf: no candidate of that half tile passes the filters
                const unsigned int p = atomicAdd(&n_hits[team], 1u);
                if (p < (unsigned int)SCAN_HITS) hits[team][p] = (unsigned short)e;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const unsigned int listed = n_hits[team];
    const bool overflow = listed > (unsigned int)SCAN_HITS || n_entries > 65535u;   // more hits than the list holds: walk again
    uint32_t n_mine = overflow ? 0u : listed, n_max = n_mine;
    #pragma unroll
    for (int off = 16; off < 64; off <<= 1) n_max = max(n_max, (uint32_t)__shfl_xor((int)n_max, off, 64));
    for (uint32_t k = 0; k < n_max; k++)
        if (k < n_mine) verify(hits[team][k]);
    if (overflow)
        for (uint32_t e = 0; e < n_entries; e++)
            if (hrow[e] >= thr && hrow[e] > -INFINITY) verify(e);
    #pragma unroll
    for (int off = 8; off > 0; off >>= 1) {                             // merge the team's partials (as match_decide_kernel)
        const float od1 = __shfl_xor(d1, off, 64), od2 = __shfl_xor(d2, off, 64);
        const int om = __shfl_xor(match, off, 64);
        if (od1 < d1) { d2 = d1; d1 = od1; match = om; }
        else {
            if (od1 < d2) d2 = od1;
            if (od1 == d1 && om >= 0 && (match < 0 || om < match)) match = om;
        }
        if (od2 < d2) d2 = od2;
        verified += (unsigned int)__shfl_xor((int)verified, off, 64);
    }
    {
        unsigned int wave_total = tl == 0 ? verified : 0u;
        #pragma unroll
        for (int off = 16; off < 64; off <<= 1) wave_total += (unsigned int)__shfl_xor((int)wave_total, off, 64);
        if (lane == 0 && wave_total) atomicAdd(a.n_dist + STAT_BASE + blockIdx.x % STAT_SLOTS, (unsigned long long)wave_total);
    }
    if (tl == 0 && valid) {
        const bool ok = (sqrtf(d1 / d2) < dist2second || d2 == FLT_MAX) && (sqrtf(d1) < threshold);   // :320-321
        out[qi] = ok ? (match >= 0 ? match : -2) : (match >= 0 ? -(match + 3) : -1);
    }
}

#define MCHECK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            frog::set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));         \
            return FROG_E_HIP;                                                               \
        }                                                                                    \
    } while (0)

// ---- matchAll (`-all`, match.cpp:297-302) -----------------------------------------------------------------------------
// With matchAll every candidate that passes the filters and lies within the threshold (sqrt(dist) < threshold) emits a pair
// at once -- and what upstream pushes is not that candidate but its `match` variable: the nearest so far among the candidates
// that were NOT within the threshold (only those reach the d1 / d2 / match update of :303-313), or, before the first such
// candidate, whatever an earlier query of the same call left there (`int match = 0;` lives outside the query loop, :259).
// The emitted values therefore depend on the order the candidates are met in: the caller's.  One lane = one query, every
// lane of the wavefront meets the same candidate at the same time, candidates staged in the CALLER's order (rows gathered
// through `c_pos`: original index -> position in the device's sorted copy); no candidate range is skipped and none is split
// over blocks -- the state is sequential.  Two passes: COUNT leaves the number of emissions and the final `match` of
// every query (-1: no update), the host turns the counts into offsets, EMIT runs the same loop again and writes the values
// (-1 where the value is the one carried in from earlier queries: the host, which walks the queries in order, knows it).
// Rarely used and quadratic by definition (no filter can bound "every candidate within the threshold" away): built to be
// upstream's list bit for bit, not to be fast -- 20 000 x 20 000 keypoints take two passes of ~6 ms.
constexpr int ALL_BLOCK = 64;

struct AllArgs {
    const float *q_desc, *q_sign, *q_scale, *q_xyz;
    const float *c_desc, *c_sign, *c_lo, *c_hi, *c_xyz;
    const uint32_t *c_pos;              // candidate, caller's index -> sorted position
    uint32_t nq, nc;
    float anat, threshold;
    int *count, *last;                  // [nq] (sorted query order)
    const unsigned long long *offset;   // EMIT: where a query's values start
    int *values;
};

template <int D, bool EMIT>
__global__ __launch_bounds__(ALL_BLOCK) void match_all_kernel(const AllArgs a)
{
    __shared__ float4 tile[CAND_TILE][D / 4];
    __shared__ float cs[CAND_TILE], clo[CAND_TILE], chi[CAND_TILE], cx[CAND_TILE], cy[CAND_TILE], cz[CAND_TILE];
    const uint32_t qi = blockIdx.x * ALL_BLOCK + threadIdx.x;
    const bool valid = qi < a.nq;
    float q[D];
    float qsign = 0.f, qscale = 0.f, qx = 0.f, qy = 0.f, qz = 0.f;
    if (valid) {
        const float4 *row = reinterpret_cast<const float4 *>(a.q_desc + (size_t)qi * D);
        #pragma unroll
        for (int k = 0; k < D / 4; k++) { const float4 v = row[k]; q[4 * k] = v.x; q[4 * k + 1] = v.y; q[4 * k + 2] = v.z; q[4 * k + 3] = v.w; }
        qsign = a.q_sign[qi]; qscale = a.q_scale[qi];
        qx = a.q_xyz[3 * (size_t)qi]; qy = a.q_xyz[3 * (size_t)qi + 1]; qz = a.q_xyz[3 * (size_t)qi + 2];
    } else {
        #pragma unroll
        for (int k = 0; k < D; k++) q[k] = 0.f;
    }
    float d1 = FLT_MAX, d2 = FLT_MAX;
    int match = -1, emitted = 0;
    const bool use_anat = a.anat != 0.f;
    int *dst = EMIT && valid ? a.values + a.offset[qi] : nullptr;

    for (uint32_t base = 0; base < a.nc; base += CAND_TILE) {
        const uint32_t cnt = min((uint32_t)CAND_TILE, a.nc - base);
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < cnt * (D / 4); k += ALL_BLOCK) {
            const uint32_t c = k / (D / 4), w = k - c * (D / 4);
            tile[c][w] = reinterpret_cast<const float4 *>(a.c_desc + (size_t)a.c_pos[base + c] * D)[w];
        }
        if (threadIdx.x < cnt) {
            const uint32_t c = a.c_pos[base + threadIdx.x];
            cs[threadIdx.x] = a.c_sign[c]; clo[threadIdx.x] = a.c_lo[c]; chi[threadIdx.x] = a.c_hi[c];
            cx[threadIdx.x] = a.c_xyz[3 * (size_t)c]; cy[threadIdx.x] = a.c_xyz[3 * (size_t)c + 1]; cz[threadIdx.x] = a.c_xyz[3 * (size_t)c + 2];
        }
        __syncthreads();
        for (uint32_t c = 0; c < cnt; c++) {
            float dist = 0.f;                                               // norm, match.cpp:242-251
            #pragma unroll
            for (int k = 0; k < D / 4; k++) {
                const float4 v = tile[c][k];
                float t;
                t = q[4 * k] - v.x;     dist += t * t;
                t = q[4 * k + 1] - v.y; dist += t * t;
                t = q[4 * k + 2] - v.z; dist += t * t;
                t = q[4 * k + 3] - v.w; dist += t * t;
            }
            bool pass = valid && qsign == cs[c] && qscale > clo[c] && qscale < chi[c];      // :270-275
            if (use_anat) {                                                                 // :278-291
                const float ex = qx - cx[c], ey = qy - cy[c], ez = qz - cz[c];
                const float eucl = sqrtf(ex * ex + ey * ey + ez * ez);
                pass = pass && !(eucl > a.anat);
            }
            const bool near = pass && sqrtf(dist) < a.threshold;                            // :297
            if (near) {
                if (EMIT) dst[emitted] = match;
                emitted++;
            }
            const bool far = pass && !near;                                                 // :303-313
            const bool better = far && dist < d1;
            const bool second = far && !better && dist < d2;
            d2 = better ? d1 : (second ? dist : d2);
            d1 = better ? dist : d1;
            match = better ? (int)(base + c) : match;
        }
    }
    if (!EMIT && valid) { a.count[qi] = emitted; a.last[qi] = match; }
}

int fail(int code, const std::string &msg) { frog::set_last_error(msg); return code; }

// smallest positive float s with (double)(s / sc) > 1.3  (reject when s >= hi), +inf if none
float scale_hi(float sc)
{
    auto rej = [sc](float s) { return (double)(s / sc) > 1.3; };
    uint32_t lo = 1u, hi = 0x7F7FFFFFu;                 // positive floats are ordered like their bits
    auto f = [](uint32_t b) { float v; std::memcpy(&v, &b, 4); return v; };
    if (!rej(f(hi))) return INFINITY;
    if (rej(f(lo))) return f(lo);
    while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (rej(f(mid))) hi = mid; else lo = mid; }
    return f(hi);
}

// largest positive float s with (double)(sc / s) > 1.3  (reject when s <= lo), 0 if none
float scale_lo(float sc)
{
    auto rej = [sc](float s) { return (double)(sc / s) > 1.3; };
    uint32_t lo = 1u, hi = 0x7F7FFFFFu;
    auto f = [](uint32_t b) { float v; std::memcpy(&v, &b, 4); return v; };
    if (!rej(f(lo))) return 0.f;
    if (rej(f(hi))) return f(hi);
    while (hi - lo > 1) { const uint32_t mid = lo + (hi - lo) / 2; if (rej(f(mid))) lo = mid; else hi = mid; }
    return f(lo);
}

} // namespace

struct frog_matcher {
    int device = 0;
    uint32_t dim = 0, dp = 0;
    std::vector<DevImage> img;
    hipStream_t stream = nullptr;
    hipStream_t extra[3] = { nullptr, nullptr, nullptr };   // image pairs rotate over stream + extra[]: one pair's tail overlaps the next pairs' start
    unsigned long long *n_dist = nullptr;
    double last_ms = 0, last_dist = 0, last_computed = 0, last_fallback = 0;
    uint64_t last_forms[3] = { 0, 0, 0 };      // passes of the last run by form: exact vector kernel / f32 matrix-core filter / bf16 matrix-core filter
};

// frog_matcher_run with o->all (see match_all_kernel).  Passes run one after the other: the second launch of a pass needs the
// first one's counts.
static int run_all(frog_matcher *m, const uint16_t *first, const uint16_t *second, size_t n_jobs, const frog_match_options *o,
                   std::vector<std::vector<uint32_t>> &ja, std::vector<std::vector<uint32_t>> &jb)
{
    uint32_t max_n = 1;
    for (size_t k = 0; k < n_jobs; k++) max_n = std::max(max_n, std::max(m->img[first[k]].n, m->img[second[k]].n));
    int *d_count = nullptr, *d_last = nullptr, *d_values = nullptr;
    unsigned long long *d_offset = nullptr;
    size_t values_cap = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    auto cleanup = [&]() {
        if (d_count) (void)hipFree(d_count);
        if (d_last) (void)hipFree(d_last);
        if (d_values) (void)hipFree(d_values);
        if (d_offset) (void)hipFree(d_offset);
        if (t0) (void)hipEventDestroy(t0);
        if (t1) (void)hipEventDestroy(t1);
    };
#define ACHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { frog::set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_)); cleanup(); return FROG_E_HIP; } } while (0)
    ACHECK(hipMalloc((void **)&d_count, (size_t)max_n * sizeof(int)));
    ACHECK(hipMalloc((void **)&d_last, (size_t)max_n * sizeof(int)));
    ACHECK(hipMalloc((void **)&d_offset, (size_t)max_n * sizeof(unsigned long long)));
    ACHECK(hipEventCreate(&t0));
    ACHECK(hipEventCreate(&t1));
    ACHECK(hipEventRecord(t0, m->stream));
    std::vector<int> count(max_n), last(max_n), values;
    std::vector<unsigned long long> offset(max_n);
    double distances = 0;
    for (size_t k = 0; k < n_jobs; k++)
        for (int dir = 0; dir < (o->sym ? 2 : 1); dir++) {
            const DevImage &C = m->img[dir ? second[k] : first[k]], &Q = m->img[dir ? first[k] : second[k]];
            const uint32_t nq = Q.n;
            if (!nq) continue;
            AllArgs a;
            a.q_desc = Q.desc; a.q_sign = Q.sign; a.q_scale = Q.scale; a.q_xyz = Q.xyz;
            a.c_desc = C.desc; a.c_sign = C.sign; a.c_lo = C.lo; a.c_hi = C.hi; a.c_xyz = C.xyz; a.c_pos = C.pos;
            a.nq = nq; a.nc = C.n; a.anat = o->anat; a.threshold = o->threshold;
            a.count = d_count; a.last = d_last; a.offset = d_offset; a.values = nullptr;
            const uint32_t blocks = (nq + ALL_BLOCK - 1) / ALL_BLOCK;
#define ALL_LAUNCH(EMIT)                                                                                       \
            switch (m->dp) {                                                                                   \
            case 48: match_all_kernel<48, EMIT><<<blocks, ALL_BLOCK, 0, m->stream>>>(a); break;                \
            case 64: match_all_kernel<64, EMIT><<<blocks, ALL_BLOCK, 0, m->stream>>>(a); break;                \
            case 96: match_all_kernel<96, EMIT><<<blocks, ALL_BLOCK, 0, m->stream>>>(a); break;                \
            default: match_all_kernel<128, EMIT><<<blocks, ALL_BLOCK, 0, m->stream>>>(a); break;               \
            }
            ALL_LAUNCH(false)
            ACHECK(hipGetLastError());
            ACHECK(hipMemcpyAsync(count.data(), d_count, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, m->stream));
            ACHECK(hipMemcpyAsync(last.data(), d_last, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, m->stream));
            ACHECK(hipStreamSynchronize(m->stream));
            distances += (double)nq * (double)C.n;
            unsigned long long total = 0;
            for (uint32_t s = 0; s < nq; s++) { offset[s] = total; total += (unsigned long long)count[s]; }
            if (total) {
                if (total > values_cap) {
                    if (d_values) { (void)hipFree(d_values); d_values = nullptr; }
                    values_cap = (size_t)total + (size_t)total / 2;
                    ACHECK(hipMalloc((void **)&d_values, values_cap * sizeof(int)));
                }
                ACHECK(hipMemcpyAsync(d_offset, offset.data(), (size_t)nq * sizeof(unsigned long long), hipMemcpyHostToDevice, m->stream));
                a.values = d_values;
                ALL_LAUNCH(true)
                ACHECK(hipGetLastError());
                values.resize(total);
                ACHECK(hipMemcpyAsync(values.data(), d_values, (size_t)total * sizeof(int), hipMemcpyDeviceToHost, m->stream));
                ACHECK(hipStreamSynchronize(m->stream));
                distances += (double)nq * (double)C.n;
            }
#undef ALL_LAUNCH
            // the queries in the caller's order; `stale` is upstream's `match` between two queries (match.cpp:259)
            std::vector<uint32_t> pos(nq);
            for (uint32_t s = 0; s < nq; s++) pos[Q.h_orig[s]] = s;
            int stale = 0;
            for (uint32_t q = 0; q < nq; q++) {
                const uint32_t s = pos[q];
                for (int e = 0; e < count[s]; e++) {
                    int v = values[offset[s] + (unsigned long long)e];
                    if (v < 0) v = stale;
                    if (dir) { ja[k].push_back(q); jb[k].push_back((uint32_t)v); }        // make_pair(i, match)
                    else { ja[k].push_back((uint32_t)v); jb[k].push_back(q); }            // make_pair(match, i)
                }
                if (last[s] >= 0) stale = last[s];
            }
        }
    ACHECK(hipEventRecord(t1, m->stream));
    ACHECK(hipStreamSynchronize(m->stream));
    float ms = 0;
    ACHECK(hipEventElapsedTime(&ms, t0, t1));
#undef ACHECK
    m->last_ms = ms; m->last_dist = distances; m->last_computed = distances; m->last_fallback = 0;
    m->last_forms[0] = n_jobs; m->last_forms[1] = m->last_forms[2] = 0;      // matchAll: the exact vector kernels only
    cleanup();
    return FROG_OK;
}

// bf16 splits of one f32 value (round to nearest even; finite inputs below 2^126 in magnitude: no overflow on the way up)
static inline uint16_t bf16_bits(float f)
{
    uint32_t u; std::memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static inline float bf16_value(uint16_t b) { const uint32_t u = (uint32_t)b << 16; float f; std::memcpy(&f, &u, 4); return f; }

// the two bf16 operand arrays of `n` points (k_mfma16: candidate form [ch | cl | ch], query form [qh' | qh' | ql']) from the
// padded descriptors and their squared norms; false when an entry is too large to split
static bool build_bf16_operands(const float *pad, const float *nrm, uint32_t n, uint32_t dp, std::vector<uint16_t> &fa, std::vector<uint16_t> &fb)
{
    const uint32_t de = dp + 2, steps = (uint32_t)mf16_steps((int)dp), groups = (n + 31) / 32;
    fa.assign((size_t)groups * steps * 64 * 8, 0); fb.assign(fa.size(), 0);
    std::vector<float> ec(de), eq(de);
    const float big = 8.5070592e37f;        // 2^126
    for (uint32_t p = 0; p < n; p++) {
        for (uint32_t k = 0; k < dp; k++) ec[k] = eq[k] = pad[(size_t)p * dp + k];
        ec[dp] = -0.5f * nrm[p]; ec[dp + 1] = 1.0f;
        eq[dp] = 1.0f; eq[dp + 1] = -0.5f * nrm[p];
        for (uint32_t k = 0; k < de; k++) if (!(std::fabs(ec[k]) < big)) return false;
        for (uint32_t kk = 0; kk < 3 * de; kk++) {
            const uint32_t k = kk % de, part = kk / de;
            const uint16_t ch = bf16_bits(ec[k]), cl = bf16_bits(ec[k] - bf16_value(ch));
            const uint16_t qh = bf16_bits(eq[k]), ql = bf16_bits(eq[k] - bf16_value(qh));
            const size_t at = ((((size_t)(p / 32) * steps + kk / 16) * 64) + ((kk % 16) / 8) * 32 + (p & 31)) * 8 + kk % 8;
            fa[at] = part == 1 ? cl : ch;
            fb[at] = part == 2 ? ql : qh;
        }
    }
    return true;
}

extern "C" {

void frog_match_options_default(frog_match_options *o)
{
    if (!o) return;
    std::memset(o, 0, sizeof *o);
    o->threshold = 0.22f;           // match.cpp:357
    o->dist2second = 1.0f;          // :358
}

void frog_matcher_destroy(frog_matcher *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    for (DevImage &d : m->img) {
        if (d.desc) (void)hipFree(d.desc);
        if (d.sign) (void)hipFree(d.sign);
        if (d.scale) (void)hipFree(d.scale);
        if (d.lo) (void)hipFree(d.lo);
        if (d.hi) (void)hipFree(d.hi);
        if (d.xyz) (void)hipFree(d.xyz);
        if (d.orig) (void)hipFree(d.orig);
        if (d.pos) (void)hipFree(d.pos);
        if (d.norm) (void)hipFree(d.norm);
        if (d.mf) (void)hipFree(d.mf);
        if (d.mfa16) (void)hipFree(d.mfa16);
        if (d.mfb16) (void)hipFree(d.mfb16);
    }
    if (m->n_dist) (void)hipFree(m->n_dist);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    for (hipStream_t e : m->extra) if (e) (void)hipStreamDestroy(e);
    delete m;
}

int frog_matcher_create(const frog_keypoints *images, uint32_t n_images, int device, frog_matcher **out)
{
    if (!images || !out || n_images == 0 || n_images > 65535) return fail(FROG_E_INVALID, "bad arguments to frog_matcher_create");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(FROG_E_NODEVICE, "no HIP device: the matcher has no CPU fallback");
    if (device < 0 || device >= count) return fail(FROG_E_INVALID, "bad device index");
    const uint32_t dim = images[0].dim;
    if (dim == 0 || dim > 128) return fail(FROG_E_INVALID, "descriptor length must be 1..128");
    for (uint32_t i = 0; i < n_images; i++) {
        if (images[i].dim != dim) return fail(FROG_E_INVALID, "all images must share one descriptor length");
        if (images[i].n && (!images[i].xyz || !images[i].scale || !images[i].laplacian || !images[i].desc))
            return fail(FROG_E_INVALID, "null keypoint array");
        for (uint32_t p = 0; p < images[i].n; p++) {
            if (!(images[i].scale[p] > 0.f) || !std::isfinite(images[i].scale[p]))
                return fail(FROG_E_INVALID, "keypoint scales must be finite and positive");
            if (std::isnan(images[i].laplacian[p])) return fail(FROG_E_INVALID, "NaN Laplacian sign");
        }
    }
    MCHECK(hipSetDevice(device));
    frog_matcher *m = new (std::nothrow) frog_matcher;
    if (!m) return fail(FROG_E_NOMEM, "out of host memory");
    m->device = device;
    m->dim = dim;
    m->dp = dim <= 48 ? 48 : dim <= 64 ? 64 : dim <= 96 ? 96 : 128;
    m->img.resize(n_images);
#define CCHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { frog::set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_)); frog_matcher_destroy(m); return FROG_E_HIP; } } while (0)
    CCHECK(hipStreamCreate(&m->stream));
    for (hipStream_t &e : m->extra) CCHECK(hipStreamCreate(&e));
    CCHECK(hipMalloc((void **)&m->n_dist, (STAT_BASE + 3 * STAT_SLOTS) * sizeof(unsigned long long)));
#define ICHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { _Pragma("omp critical(frog_matcher_create_error)") { if (first_error == hipSuccess) { first_error = e_; first_expr = #expr; } } thread_ok = false; } } while (0); if (!thread_ok) continue
    // Every image is prepared (sorted by sign and scale, padded, its norms, the operands of the two matrix-core forms) and uploaded
    // on its own: images over the host threads -- one after the other they took 2.0 of bin/match's 3.6 s for 100 images x 20 000.
    // Device buffers first, one thread, image by image in the order the serial version allocated them: with the allocations made
    // from the host threads as they came, an image's operands lay scattered among the other images' and the filter ran 4-8 %
    // slower (kernel seconds 0.307-0.326 against 0.285-0.296 over 4 950 image pairs).
    for (uint32_t i = 0; i < n_images; i++) {
        DevImage &d = m->img[i];
        const size_t n = std::max<uint32_t>(1, images[i].n);
        const size_t steps = (m->dp + 2) / 2, groups = (n + 31) / 32;
        CCHECK(hipMalloc((void **)&d.norm, n * sizeof(float)));
        CCHECK(hipMalloc((void **)&d.mf, groups * steps * 64 * sizeof(float)));
        if (m->dp <= 64) {
            const size_t n16 = std::max<size_t>(1, (images[i].n + 31) / 32) * (size_t)mf16_steps((int)m->dp) * 64 * 8;
            CCHECK(hipMalloc((void **)&d.mfa16, n16 * sizeof(uint16_t)));
            CCHECK(hipMalloc((void **)&d.mfb16, n16 * sizeof(uint16_t)));
        }
        CCHECK(hipMalloc((void **)&d.desc, n * m->dp * sizeof(float)));
        CCHECK(hipMalloc((void **)&d.sign, n * sizeof(float)));
        CCHECK(hipMalloc((void **)&d.scale, n * sizeof(float)));
        CCHECK(hipMalloc((void **)&d.lo, n * sizeof(float)));
        CCHECK(hipMalloc((void **)&d.hi, n * sizeof(float)));
        CCHECK(hipMalloc((void **)&d.xyz, n * 3 * sizeof(float)));
        CCHECK(hipMalloc((void **)&d.orig, n * sizeof(uint32_t)));
        if (images[i].n) CCHECK(hipMalloc((void **)&d.pos, n * sizeof(uint32_t)));
    }
    hipError_t first_error = hipSuccess;
    std::string first_expr;
    #pragma omp parallel num_threads(frog::host_threads())
    {
    std::vector<float> pad, lo, hi, sg, sc, xyz, nrm, mfv;
    std::vector<uint16_t> fa16, fb16;
    bool thread_ok = true;
    {
        const hipError_t e_ = hipSetDevice(device);                     // the current device is a property of the thread
        if (e_ != hipSuccess) {
            #pragma omp critical(frog_matcher_create_error)
            { if (first_error == hipSuccess) { first_error = e_; first_expr = "hipSetDevice(device)"; } }
            thread_ok = false;
        }
    }
    #pragma omp for schedule(dynamic, 1)
    for (long long i = 0; i < (long long)n_images; i++) {
        if (!thread_ok) continue;
        const frog_keypoints &k = images[i];
        DevImage &d = m->img[i];
        d.n = k.n;
        const size_t n = std::max<uint32_t>(1, k.n);
        // device order: by (Laplacian sign, scale, original index)
        d.h_orig.resize(k.n);
        for (uint32_t p = 0; p < k.n; p++) d.h_orig[p] = p;
        std::sort(d.h_orig.begin(), d.h_orig.end(), [&k](uint32_t x, uint32_t y) {
            if (k.laplacian[x] != k.laplacian[y]) return k.laplacian[x] < k.laplacian[y];
            if (k.scale[x] != k.scale[y]) return k.scale[x] < k.scale[y];
            return x < y;
        });
        pad.assign(n * m->dp, 0.f);
        lo.assign(n, 0.f); hi.assign(n, 0.f); sg.assign(n, 0.f); sc.assign(n, 0.f); xyz.assign(3 * n, 0.f);
        for (uint32_t p = 0; p < k.n; p++) {
            const uint32_t o = d.h_orig[p];
            std::memcpy(&pad[(size_t)p * m->dp], k.desc + (size_t)o * dim, dim * sizeof(float));
            lo[p] = scale_lo(k.scale[o]); hi[p] = scale_hi(k.scale[o]);
            sg[p] = k.laplacian[o]; sc[p] = k.scale[o];
            for (int c = 0; c < 3; c++) xyz[3 * (size_t)p + c] = k.xyz[3 * (size_t)o + c];
        }
        nrm.assign(n, 0.f);
        for (uint32_t p = 0; p < k.n; p++) {
            double sum = 0;
            for (uint32_t c = 0; c < m->dp; c++) { const double v = pad[(size_t)p * m->dp + c]; sum += v * v; }
            nrm[p] = (float)sum;
            if (!std::isfinite(nrm[p])) d.finite = false;
            d.norm_max = std::max(d.norm_max, nrm[p]);
        }
        ICHECK(hipMemcpy(d.norm, nrm.data(), n * sizeof(float), hipMemcpyHostToDevice));
        {
            // (p, -|p|^2/2, 1) by groups of 32 points in the lane order of v_mfma_f32_32x32x2_f32's operands:
            // lane l of step s holds dimension 2s + (l >> 5) of point l & 31 -- the same for A and for B
            const uint32_t steps = (m->dp + 2) / 2, groups = (uint32_t)((n + 31) / 32);
            mfv.assign((size_t)groups * steps * 64, 0.f);
            for (uint32_t p = 0; p < k.n; p++)
                for (uint32_t dim2 = 0; dim2 < m->dp + 2; dim2++) {
                    const float v = dim2 < m->dp ? pad[(size_t)p * m->dp + dim2] : (dim2 == m->dp ? -0.5f * nrm[p] : 1.0f);
                    mfv[((size_t)(p / 32) * steps + dim2 / 2) * 64 + (dim2 & 1) * 32 + (p & 31)] = v;
                }
            ICHECK(hipMemcpy(d.mf, mfv.data(), mfv.size() * sizeof(float), hipMemcpyHostToDevice));
        }
        if (m->dp <= 64 && d.finite) {
            d.bf16_ok = build_bf16_operands(pad.data(), nrm.data(), k.n, m->dp, fa16, fb16);
            if (d.bf16_ok) {
                if (fa16.empty()) { fa16.assign((size_t)mf16_steps((int)m->dp) * 64 * 8, 0); fb16 = fa16; }    // an image without keypoints
                ICHECK(hipMemcpy(d.mfa16, fa16.data(), fa16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
                ICHECK(hipMemcpy(d.mfb16, fb16.data(), fb16.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            }
        }
        ICHECK(hipMemcpy(d.desc, pad.data(), n * m->dp * sizeof(float), hipMemcpyHostToDevice));
        if (k.n) {
            ICHECK(hipMemcpy(d.sign, sg.data(), k.n * sizeof(float), hipMemcpyHostToDevice));
            ICHECK(hipMemcpy(d.scale, sc.data(), k.n * sizeof(float), hipMemcpyHostToDevice));
            ICHECK(hipMemcpy(d.lo, lo.data(), k.n * sizeof(float), hipMemcpyHostToDevice));
            ICHECK(hipMemcpy(d.hi, hi.data(), k.n * sizeof(float), hipMemcpyHostToDevice));
            ICHECK(hipMemcpy(d.xyz, xyz.data(), (size_t)k.n * 3 * sizeof(float), hipMemcpyHostToDevice));
            ICHECK(hipMemcpy(d.orig, d.h_orig.data(), k.n * sizeof(uint32_t), hipMemcpyHostToDevice));
            std::vector<uint32_t> inv(k.n);
            for (uint32_t p = 0; p < k.n; p++) inv[d.h_orig[p]] = p;
            ICHECK(hipMemcpy(d.pos, inv.data(), k.n * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }
    }
#undef ICHECK
    if (first_error != hipSuccess) {
        frog::set_last_error(first_expr + ": " + hipGetErrorString(first_error));
        frog_matcher_destroy(m);
        return FROG_E_HIP;
    }
#undef CCHECK
    *out = m;
    return FROG_OK;
}

// Test hook: the matrix-core filter's products for 32 candidates x 32 queries (descriptors of `dim` = 48 or 64 values, row
// major), through the same operand builders and instruction sequences as the filter kernels: form 0 = the f32 chain,
// 1 = three products of bf16 splits.  out[c * 32 + q] approximates -|q - c|^2 / 2; bound[0] = the bound the scan kernel uses
// for that form (relative to |q|^2 + |c|^2, on -2 x product).  tests/test_gpu_match.py compares with f64.
int frog_match_test_products(int device, const float *cand, const float *query, uint32_t dim, int form, float *out, float *bound)
{
    if (!cand || !query || !out || (dim != 48 && dim != 64) || (form != 0 && form != 1)) return fail(FROG_E_INVALID, "bad arguments to frog_match_test_products");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(FROG_E_NODEVICE, "no HIP device");
    if (device < 0 || device >= count) return fail(FROG_E_INVALID, "bad device index");
    MCHECK(hipSetDevice(device));
    const uint32_t steps = (dim + 2) / 2, steps16 = (uint32_t)mf16_steps((int)dim);
    std::vector<float> nc(32), nq(32), mfc((size_t)steps * 64, 0.f), mfq((size_t)steps * 64, 0.f);
    auto norms = [&](const float *d, std::vector<float> &n) {
        for (int p = 0; p < 32; p++) { double sum = 0; for (uint32_t c = 0; c < dim; c++) sum += (double)d[p * dim + c] * d[p * dim + c]; n[p] = (float)sum; }
    };
    norms(cand, nc); norms(query, nq);
    auto f32_form = [&](const float *d, const std::vector<float> &n, std::vector<float> &mf) {
        for (uint32_t p = 0; p < 32; p++)
            for (uint32_t k = 0; k < dim + 2; k++)
                mf[((size_t)(k / 2)) * 64 + (k & 1) * 32 + p] = k < dim ? d[p * dim + k] : (k == dim ? -0.5f * n[p] : 1.0f);
    };
    f32_form(cand, nc, mfc); f32_form(query, nq, mfq);
    std::vector<uint16_t> ca, cb, qa, qb;
    if (!build_bf16_operands(cand, nc.data(), 32, dim, ca, cb) || !build_bf16_operands(query, nq.data(), 32, dim, qa, qb))
        return fail(FROG_E_INVALID, "descriptor entries too large for the bf16 split");
    float *d_mfc = nullptr, *d_mfq = nullptr, *d_out = nullptr; uint16_t *d_ca = nullptr, *d_qb = nullptr;
    auto release = [&]() { (void)hipFree(d_mfc); (void)hipFree(d_mfq); (void)hipFree(d_out); (void)hipFree(d_ca); (void)hipFree(d_qb); };
#define TCHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { frog::set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_)); release(); return FROG_E_HIP; } } while (0)
    TCHECK(hipMalloc((void **)&d_mfc, mfc.size() * 4)); TCHECK(hipMalloc((void **)&d_mfq, mfq.size() * 4)); TCHECK(hipMalloc((void **)&d_out, 1024 * 4));
    TCHECK(hipMalloc((void **)&d_ca, (size_t)steps16 * 64 * 16)); TCHECK(hipMalloc((void **)&d_qb, (size_t)steps16 * 64 * 16));
    TCHECK(hipMemcpy(d_mfc, mfc.data(), mfc.size() * 4, hipMemcpyHostToDevice)); TCHECK(hipMemcpy(d_mfq, mfq.data(), mfq.size() * 4, hipMemcpyHostToDevice));
    TCHECK(hipMemcpy(d_ca, ca.data(), (size_t)steps16 * 64 * 16, hipMemcpyHostToDevice)); TCHECK(hipMemcpy(d_qb, qb.data(), (size_t)steps16 * 64 * 16, hipMemcpyHostToDevice));
    if (dim == 48) match_products_kernel<48><<<1, 64>>>(d_mfq, d_mfc, (const uint4 *)d_qb, (const uint4 *)d_ca, form, d_out);
    else match_products_kernel<64><<<1, 64>>>(d_mfq, d_mfc, (const uint4 *)d_qb, (const uint4 *)d_ca, form, d_out);
    TCHECK(hipGetLastError());
    TCHECK(hipMemcpy(out, d_out, 1024 * 4, hipMemcpyDeviceToHost));
#undef TCHECK
    release();
    if (bound) *bound = form ? MF_EPS_BF16 : MF_EPS;
    return FROG_OK;
}

int frog_matcher_last_forms(const frog_matcher *m, uint64_t passes_by_form[3])
{
    if (!m || !passes_by_form) return FROG_E_INVALID;
    for (int k = 0; k < 3; k++) passes_by_form[k] = m->last_forms[k];
    return FROG_OK;
}

int frog_matcher_last_stats(const frog_matcher *m, double *kernel_ms, double *distances)
{
    if (!m) return FROG_E_INVALID;
    if (kernel_ms) *kernel_ms = m->last_ms;
    if (distances) *distances = m->last_dist;
    return FROG_OK;
}

void frog_match_free(void *p) { std::free(p); }

int frog_matcher_run(frog_matcher *m, const uint16_t *first, const uint16_t *second, size_t n_jobs,
                     const frog_match_options *o, uint64_t *offset, uint32_t **p_first, uint32_t **p_second)
{
    if (!m || !o || !offset || !p_first || !p_second || (n_jobs && (!first || !second)))
        return fail(FROG_E_INVALID, "bad arguments to frog_matcher_run");
    for (size_t k = 0; k < n_jobs; k++)
        if (first[k] >= m->img.size() || second[k] >= m->img.size()) return fail(FROG_E_INVALID, "image index out of range");
    MCHECK(hipSetDevice(m->device));
    *p_first = *p_second = nullptr;

    // a job = up to two passes (forward, and the reverse one with -sym); passes run back to back on
    // one stream, results come back through a ring of pinned buffers while the next passes compute
    struct Pass { uint32_t job, cand, query; bool sym; };
    std::vector<Pass> passes;
    uint32_t max_n = 1;
    for (size_t k = 0; k < n_jobs; k++) {
        passes.push_back(Pass{ (uint32_t)k, first[k], second[k], false });
        if (o->sym) passes.push_back(Pass{ (uint32_t)k, second[k], first[k], true });
        max_n = std::max(max_n, std::max(m->img[first[k]].n, m->img[second[k]].n));
    }
    constexpr int RING = 8;
    const uint32_t splits_max = 64;
    const uint32_t q_blocks_max = (max_n + MATCH_BLOCK - 1) / MATCH_BLOCK;
    Partial *partial = nullptr;
    QRange *qrange = nullptr;
    float *hmax = nullptr;                      // [tiles of the largest image][2][max_n] per ring slot
    // FROG_MATCH_VALU=1 keeps every pass on the exact vector-ALU kernel (test hook; also taken for descriptors
    // longer than 64 values or with non-finite entries, where the MFMA filter's error bound means nothing)
    const bool force_valu = getenv("FROG_MATCH_VALU") != nullptr;
    m->last_forms[0] = m->last_forms[1] = m->last_forms[2] = 0;
    uint2 *ranges = nullptr;
    int *d_out = nullptr, *h_out = nullptr;
    hipEvent_t done[RING] = {}, t0 = nullptr, t1 = nullptr;
    std::vector<std::vector<uint32_t>> ja(n_jobs), jb(n_jobs);
    int rc = FROG_OK;
    auto hand_over = [&]() -> int {
        offset[0] = 0;
        for (size_t k = 0; k < n_jobs; k++) offset[k + 1] = offset[k] + ja[k].size();
        *p_first = (uint32_t *)std::malloc(std::max<size_t>(1, offset[n_jobs]) * sizeof(uint32_t));
        *p_second = (uint32_t *)std::malloc(std::max<size_t>(1, offset[n_jobs]) * sizeof(uint32_t));
        if (!*p_first || !*p_second) { std::free(*p_first); std::free(*p_second); *p_first = *p_second = nullptr; return fail(FROG_E_NOMEM, "out of host memory"); }
        // 66 M pairs of a 100-image group are 530 MB to touch for the first time and fill: by a few threads, a slice of the jobs each
        auto fill = [&](size_t k0, size_t k1) {
            for (size_t k = k0; k < k1; k++) {
                if (ja[k].empty()) continue;
                std::memcpy(*p_first + offset[k], ja[k].data(), ja[k].size() * sizeof(uint32_t));
                std::memcpy(*p_second + offset[k], jb[k].data(), jb[k].size() * sizeof(uint32_t));
                std::vector<uint32_t>().swap(ja[k]); std::vector<uint32_t>().swap(jb[k]);
            }
        };
        const size_t n_fill = offset[n_jobs] > (1u << 22) ? std::min<size_t>(4, n_jobs) : 1;
        std::vector<std::thread> fillers;
        size_t done_to = n_jobs / n_fill;             // jobs [0, done_to) are this thread's; a slice whose thread cannot be started joins them
        std::vector<std::pair<size_t, size_t>> mine{ { 0, done_to } };
        for (size_t t = 1; t < n_fill; t++) {
            const size_t k0 = n_jobs * t / n_fill, k1 = n_jobs * (t + 1) / n_fill;
            try { fillers.emplace_back(fill, k0, k1); } catch (...) { mine.push_back({ k0, k1 }); }
        }
        for (const auto &r : mine) fill(r.first, r.second);
        for (std::thread &t : fillers) t.join();
        return FROG_OK;
    };
    if (o->all) {                                   // matchAll: its own sequential kernels
        rc = run_all(m, first, second, n_jobs, o, ja, jb);
        return rc ? rc : hand_over();
    }
    std::vector<std::thread> collectors;
    std::atomic<size_t> launched{0};
    std::atomic<bool> stop{false}, failed{false};
    std::atomic<bool> slot_free[RING];
    std::mutex mtx;
    std::condition_variable cv_launched, cv_slot;
    auto cleanup = [&]() {
        { std::lock_guard<std::mutex> lk(mtx); stop.store(true); }     // an error on the way: the collectors give up where they are
        cv_launched.notify_all(); cv_slot.notify_all();
        for (std::thread &t : collectors) if (t.joinable()) t.join();
        collectors.clear();
        if (partial) (void)hipFree(partial);
        if (qrange) (void)hipFree(qrange);
        if (hmax) (void)hipFree(hmax);
        if (ranges) (void)hipFree(ranges);
        if (d_out) (void)hipFree(d_out);
        if (h_out) (void)hipHostFree(h_out);
        for (int r = 0; r < RING; r++) if (done[r]) (void)hipEventDestroy(done[r]);
        if (t0) (void)hipEventDestroy(t0);
        if (t1) (void)hipEventDestroy(t1);
    };
#define RCHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { frog::set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_)); cleanup(); return FROG_E_HIP; } } while (0)
    RCHECK(hipMalloc((void **)&partial, (size_t)RING * splits_max * max_n * sizeof(Partial)));
    RCHECK(hipMalloc((void **)&qrange, (size_t)RING * max_n * sizeof(QRange)));
    const size_t hmax_slot = (size_t)((max_n + MF_TILE - 1) / MF_TILE) * 2 * max_n;
    // the half-tile maxima need n^2 / 16 floats per image pair in flight (100 MB at 20 000 keypoints): images beyond
    // ~90 000 keypoints (16 GB for the ring) take the vector kernel
    const bool hmax_fits = (double)RING * (double)hmax_slot * sizeof(float) <= 16.0 * 1024 * 1024 * 1024;
    if (!force_valu && m->dp <= 64 && hmax_fits) RCHECK(hipMalloc((void **)&hmax, (size_t)RING * hmax_slot * sizeof(float)));
    RCHECK(hipMalloc((void **)&ranges, (size_t)RING * q_blocks_max * sizeof(uint2)));
    RCHECK(hipMalloc((void **)&d_out, (size_t)RING * max_n * sizeof(int)));
    RCHECK(hipHostMalloc((void **)&h_out, (size_t)RING * max_n * sizeof(int)));
    for (int r = 0; r < RING; r++) RCHECK(hipEventCreateWithFlags(&done[r], hipEventDisableTiming | hipEventBlockingSync));
    RCHECK(hipEventCreate(&t0));
    RCHECK(hipEventCreate(&t1));
    RCHECK(hipMemsetAsync(m->n_dist, 0, (STAT_BASE + 3 * STAT_SLOTS) * sizeof(unsigned long long), m->stream));
    RCHECK(hipEventRecord(t0, m->stream));
    static const int n_streams = getenv("FROG_MATCH_STREAMS") ? std::min(4, std::max(1, atoi(getenv("FROG_MATCH_STREAMS")))) : 4;   // measured, image pairs/s: 1 stream 4537, 2: 5832, 3: 6345, 4: 6652 (a slot is reused only after the host has waited for it)
    for (int k = 1; k < n_streams; k++) RCHECK(hipStreamWaitEvent(m->extra[k - 1], t0, 0));

    // upstream's `match` variable lives across the queries of one ComputeMatches call
    auto collect = [&](size_t pi, std::vector<int> &by_query, std::vector<uint32_t> &ta, std::vector<uint32_t> &tb) {
        const Pass &ps = passes[pi];
        const int *res = h_out + (size_t)(pi % RING) * max_n;
        const DevImage &Q = m->img[ps.query];
        const uint32_t nq = Q.n;
        by_query.resize(nq);
        for (uint32_t s = 0; s < nq; s++) by_query[Q.h_orig[s]] = res[s];     // back to the caller's query order
        // into the thread's own buffers first: the jobs' vectors sit side by side in memory, and two threads appending to
        // neighbouring jobs would hand the cache line of their headers back and forth at every push_back
        ta.clear(); tb.clear();
        int stale = 0;                                  // `int match = 0;`, match.cpp:259
        for (uint32_t q = 0; q < nq; q++) {
            int v = by_query[q];
            if (v <= -3) { stale = -(v + 3); continue; }
            if (v == -1) continue;
            if (v >= 0) stale = v; else v = stale;      // -2: accepted, no candidate
            if (ps.sym) { ta.push_back(q); tb.push_back((uint32_t)v); }     // make_pair(i, match)
            else { ta.push_back((uint32_t)v); tb.push_back(q); }           // make_pair(match, i)
        }
        ja[ps.job].insert(ja[ps.job].end(), ta.begin(), ta.end());
        jb[ps.job].insert(jb[ps.job].end(), tb.begin(), tb.end());
    };

    // Turning a pass's result codes into its job's pair list is a walk over the queries in the caller's order (the `match`
    // that survives from query to query) -- ~90 us of host time per pass of 20 000 queries, more than the device needs for the
    // pass since the filter moved to the bf16 cores.  Collector threads do it beside the thread that queues the work: a job's
    // passes (forward, then -sym's reverse) always go to the same thread, in order; a ring slot is reused once its pass has
    // been collected.
    static const int n_collectors = getenv("FROG_MATCH_COLLECTORS") ? std::min(8, std::max(1, atoi(getenv("FROG_MATCH_COLLECTORS")))) : 3;
    for (int r = 0; r < RING; r++) slot_free[r].store(true);
    for (int k = 0; k < n_collectors; k++)
        try { collectors.emplace_back([&, k]() {
            (void)hipSetDevice(m->device);
            std::vector<int> by_query;
            std::vector<uint32_t> ta, tb;
            for (size_t pi = 0; pi < passes.size(); pi++) {
                if ((int)(passes[pi].job % (uint32_t)n_collectors) != k) continue;
                {
                    std::unique_lock<std::mutex> lk(mtx);           // asleep until the pass is in the queue: a spinning collector
                    cv_launched.wait(lk, [&] { return launched.load() > pi || stop.load(); });     // takes a core from the queueing thread
                }
                if (stop.load()) return;
                const int slot = (int)(pi % RING);
                if (hipEventSynchronize(done[slot]) != hipSuccess) { failed.store(true); stop.store(true); cv_slot.notify_all(); return; }
                collect(pi, by_query, ta, tb);
                { std::lock_guard<std::mutex> lk(mtx); slot_free[slot].store(true); }
                cv_slot.notify_all();
            }
        }); } catch (...) {                             // no thread to be had: nothing has been queued yet
            frog::set_last_error("cannot start a collector thread");
            cleanup();
            return FROG_E_NOMEM;
        }

    for (size_t pi = 0; pi < passes.size(); pi++) {
        const int slot = (int)(pi % RING);
        {
            std::unique_lock<std::mutex> lk(mtx);
            cv_slot.wait(lk, [&] { return slot_free[slot].load() || stop.load(); });
        }
        if (failed.load()) { frog::set_last_error("hipEventSynchronize failed in a collector thread"); cleanup(); return FROG_E_HIP; }
        slot_free[slot].store(false);
        struct Launched {                               // at the end of the iteration, also on `continue`
            std::atomic<size_t> &n; size_t v; std::mutex &m; std::condition_variable &cv;
            ~Launched() { { std::lock_guard<std::mutex> lk(m); n.store(v); } cv.notify_all(); }
        } mark{ launched, pi + 1, mtx, cv_launched };
        const Pass &ps = passes[pi];
        const DevImage &Q = m->img[ps.query], &C = m->img[ps.cand];
        const uint32_t nq = Q.n;
        hipStream_t st = (pi % n_streams) ? m->extra[pi % n_streams - 1] : m->stream;
        if (nq) {
            const uint32_t q_blocks = (nq + MATCH_BLOCK - 1) / MATCH_BLOCK;
            // every query block splits ITS candidate range over `splits` blocks (whole tiles)
            // (measured on 20 000 x 20 000 x 48: 8 splits 655, 13: 905, 26: 1318, 52: 1387 image pairs/s)
            const uint32_t splits = std::max(1u, std::min(splits_max, (4096u + q_blocks - 1) / q_blocks));
            MatchArgs a;
            a.q_desc = Q.desc; a.q_sign = Q.sign; a.q_scale = Q.scale; a.q_xyz = Q.xyz;
            a.c_desc = C.desc; a.c_sign = C.sign; a.c_lo = C.lo; a.c_hi = C.hi; a.c_xyz = C.xyz;
            a.c_orig = C.orig;
            a.nq = nq; a.nc = C.n; a.splits = splits; a.anat = o->anat;
            a.partial = partial + (size_t)slot * splits_max * max_n;
            a.n_dist = m->n_dist;
            a.mf_eps = MF_EPS;
            uint2 *rg = ranges + (size_t)slot * q_blocks_max;
            const bool mfma = hmax != nullptr && Q.finite && C.finite && C.n > 0;
            if (!mfma) m->last_forms[0]++;
            if (mfma) {
                // candidate tiles per block: few = even load over the 256 CUs (the query blocks' ranges differ a lot in
                // length), many = the block's 256 queries are loaded as operands less often.  Measured, image pairs/s
                // at 20 000 x 20 000: 2 tiles 1584, 4: 2059, 8: 2249, 16: 2110
                // The bf16 form (round 5) multiplies a tile five times faster and pays the same for its block's query operands:
                // 64 tiles per block (image pairs/s with 4 / 8 / 16 / 32 / 64 / 128: 7 957 / 8 606 / 8 475 / 8 474 / 9 209 / 8 562;
                // the blocks of a pass no longer fill the chip -- 158 of them -- but four streams of passes do)
                static const int tiles_env = getenv("FROG_MATCH_TILES") ? std::max(1, atoi(getenv("FROG_MATCH_TILES"))) : 0;
                static const bool force_f32 = getenv("FROG_MATCH_F32") != nullptr;
                const bool bf16 = !force_f32 && Q.bf16_ok && C.bf16_ok;
                m->last_forms[bf16 ? 2 : 1]++;
                const uint32_t tiles_per_block = tiles_env ? (uint32_t)tiles_env : (bf16 ? 64u : 8u);
                a.splits = std::max(1u, std::min(splits_max, (C.n / 5 / MF_TILE + tiles_per_block - 1) / tiles_per_block));
                QRange *qr = qrange + (size_t)slot * max_n;
                float *hm = hmax + (size_t)slot * hmax_slot;
                a.hmax_stride = 2 * ((C.n + MF_TILE - 1) / MF_TILE);
                const dim3 mgrid(q_blocks, a.splits);
                int *dst = d_out + (size_t)slot * max_n;
                const bool anat = o->anat != 0.f;
                // query groups per wavefront: 4 halves the candidate loads per product but needs 234 registers (2 wavefronts per
                // SIMD instead of 3); measured the same within noise (5 971 vs 5 982 image pairs/s), so 2 stays the default
                static const int mf_groups = getenv("FROG_MATCH_GROUPS") && atoi(getenv("FROG_MATCH_GROUPS")) == 4 ? 4 : 2;
                match_qrange_kernel<<<q_blocks, MATCH_BLOCK, 0, st>>>(a, qr, rg);
                // the bf16 matrix cores by default (16 x the f32 rate, a wider but proven bound: match_mfma16_kernel);
                // FROG_MATCH_F32=1 keeps the f32 chain (A/B, tests)
                a.mf_eps = bf16 ? MF_EPS_BF16 : MF_EPS;
#define MF_LAUNCH(DD, AA)                                                                                               \
                do {                                                                                                    \
                    if (bf16) match_mfma16_kernel<DD, AA><<<mgrid, MATCH_BLOCK, 0, st>>>(a, rg, qr, (const uint4 *)Q.mfb16, (const uint4 *)C.mfa16, hm); \
                    else if (mf_groups == 4) match_mfma_kernel<DD, AA, 4><<<mgrid, MATCH_BLOCK / 2, 0, st>>>(a, rg, qr, Q.mf, C.mf, hm); \
                    else match_mfma_kernel<DD, AA, 2><<<mgrid, MATCH_BLOCK, 0, st>>>(a, rg, qr, Q.mf, C.mf, hm);       \
                    match_scan_kernel<DD, AA><<<(nq + SCAN_BLOCK / 16 - 1) / (SCAN_BLOCK / 16), SCAN_BLOCK, 0, st>>>(a, rg, qr, Q.norm, C.norm_max, hm, \
                                                                                       o->threshold, o->dist2second, dst); \
                } while (0)
                if (m->dp == 48) { if (anat) MF_LAUNCH(48, true); else MF_LAUNCH(48, false); }
                else { if (anat) MF_LAUNCH(64, true); else MF_LAUNCH(64, false); }
#undef MF_LAUNCH
                RCHECK(hipGetLastError());
                RCHECK(hipMemcpyAsync(h_out + (size_t)slot * max_n, dst, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost, st));
                RCHECK(hipEventRecord(done[slot], st));
                continue;
            }
            const dim3 grid(q_blocks, splits);
            match_range_kernel<<<(q_blocks + 63) / 64, 64, 0, st>>>(a, q_blocks, rg);
            switch (m->dp) {
            case 48: match_kernel<48><<<grid, MATCH_BLOCK, 0, st>>>(a, rg); break;
            case 64: match_kernel<64><<<grid, MATCH_BLOCK, 0, st>>>(a, rg); break;
            case 96: match_kernel<96><<<grid, MATCH_BLOCK, 0, st>>>(a, rg); break;
            default: match_kernel<128><<<grid, MATCH_BLOCK, 0, st>>>(a, rg); break;
            }
            match_decide_kernel<<<(nq + 255) / 256, 256, 0, st>>>(a.partial, nq, splits, o->threshold, o->dist2second,
                                                                         d_out + (size_t)slot * max_n);
            RCHECK(hipGetLastError());
            RCHECK(hipMemcpyAsync(h_out + (size_t)slot * max_n, d_out + (size_t)slot * max_n, (size_t)nq * sizeof(int),
                                  hipMemcpyDeviceToHost, st));
        }
        RCHECK(hipEventRecord(done[slot], st));
    }
    {
        for (int k = 1; k < n_streams; k++) {
            hipEvent_t joined = nullptr;
            RCHECK(hipEventCreateWithFlags(&joined, hipEventDisableTiming));
            hipError_t e1 = hipEventRecord(joined, m->extra[k - 1]), e2 = hipStreamWaitEvent(m->stream, joined, 0);
            (void)hipEventDestroy(joined);
            RCHECK(e1); RCHECK(e2);
        }
    }
    RCHECK(hipEventRecord(t1, m->stream));
    RCHECK(hipStreamSynchronize(m->stream));
    for (std::thread &t : collectors) t.join();         // every pass has been queued: they run to the end of the list
    collectors.clear();
    if (failed.load()) { frog::set_last_error("hipEventSynchronize failed in a collector thread"); cleanup(); return FROG_E_HIP; }
    float ms = 0;
    RCHECK(hipEventElapsedTime(&ms, t0, t1));
    std::vector<unsigned long long> stat(STAT_BASE + 3 * STAT_SLOTS, 0ull);
    RCHECK(hipMemcpy(stat.data(), m->n_dist, stat.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long nd[3] = { stat[0], stat[1], stat[2] };
    for (int k = 0; k < STAT_SLOTS; k++) { nd[2] += stat[STAT_BASE + k]; nd[1] += stat[STAT_BASE + STAT_SLOTS + k]; nd[0] += stat[STAT_BASE + 2 * STAT_SLOTS + k]; }
    m->last_ms = ms; m->last_dist = (double)nd[0]; m->last_computed = (double)nd[1]; m->last_fallback = (double)nd[2];
    if (getenv("FROG_MATCH_DEBUG")) std::fprintf(stderr, "matcher: %.3f ms, %.4g pairs pass the filters, %.4g computed, %.4g exact distances after the filter\n", ms, (double)nd[0], (double)nd[1], (double)nd[2]);
#undef RCHECK
    cleanup();

    rc = hand_over();
    return rc;
}

} // extern "C"
