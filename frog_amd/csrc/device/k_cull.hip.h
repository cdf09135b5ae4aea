// k_cull.hip.h -- certified outlier culling for the deformable half-link sweep.
//
// In updateDeformableTransforms a half-link whose weight is below inlierThreshold contributes NOTHING:
// not to the point's sums, not to the energy (imageGroup.cxx:268-278: `if ( weight < inlierThreshold ) continue`).
// A third of the links of a typical group are false matches whose end points lie hundreds of millimetres
// apart, and they stay outliers for the whole run -- but the sweep pays a record, a scattered 12-byte gather
// and four exponentials for each of them, every iteration.  This file lets the sweep skip the links that are
// PROVABLY outliers, the way a molecular-dynamics code keeps a neighbour list with a skin:
//
//   cut_now[k]   per image, refreshed with the mixture: a distance D_k such that every d >= D_k has
//                getInlierProbability_k(d) <= inlierThreshold / 2 (cull_cutoff_kernel; the probability of a
//                two-component Maxwell mixture with c1 < c2 decreases in d beyond sqrt(2) c1, see there).
//                weight = min(pA, pB), so d >= min(D_A, D_B) makes the link an outlier, whatever the other end says.
//   list         built once in a while (cull_build_kernel): the records of every (tile, partner group) whose
//                distance at build time is below min(cut_list[A], cut_list[B]), compacted IN THEIR ORDER into a
//                second record array; cut_list[k] = scale * cut_now[k] + pad is generous (the skin).
//   check        before every sweep (cull_validate_kernel): D = largest distance of any point from where it was at
//                build time (measured by the B-spline transform that produced the coordinates, or by
//                cull_disp_kernel).  A link left out at build time had d_build >= cut_list[k] for k = A or B, so
//                now d >= cut_list[k] - 2 D; if
//                    cut_list[k] - 2 D >= cut_now[k]  (+ rounding margin)   for EVERY image k
//                every left-out link is still a certain outlier and the sweep may walk the list instead of all
//                records.  Otherwise the flag tells the sweep (on the device, same launch) to walk all records,
//                and the host (through the fourth scalar it reads back per iteration anyway) to rebuild the list.
//
// The listed records keep their relative order and the skipped ones would have added nothing: the f32 per-point sums --
// and with them lattices, coordinates, the census -- are bit-identical to the full sweep's.  The energy's two f64 sums are
// accumulated per lane and then over the wavefront; compaction moves records to other lanes, so their association
// changes: equal up to f64 re-association of f32 terms (identical bits in every run compared so far:
// tests/test_gpu_round2.py compares whole runs with FROG_CULL=0 and 1, and with a zero skin that invalidates the list at
// every step).  Nothing is approximated and no decision is cached: a listed link is evaluated from scratch each iteration.
#pragma once

#include "ctx.h"
#include "k_links.hip.h"
#include "k_stats.hip.h"

namespace frog {

// Stats::getInlierProbability (stats.h:84-92) in f64, real-valued form (no f32 roundings): only used to
// place the cutoff, with a factor-two margin on the probability, so 1e-6-level differences are irrelevant.
__device__ inline double mixture_probability(double d, double c1, double c2, double ratio)
{
    const double eps = 1e-10, c = 0.797884560802865;
    const double a = d / (c1 + eps), b = d / (c2 + eps);
    const double x1 = ratio * c * a * a * exp(-0.5 * a * a) / (c1 + eps);
    const double x2 = (1.0 - ratio) * c * b * b * exp(-0.5 * b * b) / (c2 + eps);
    return x1 / (x1 + x2 + eps);
}

// cut_now[k]: smallest distance (to bisection accuracy, rounded up) from which image k's inlier probability is
// certainly below the threshold.  p = 1 / (1 + x2/x1 + eps/x1); x2/x1 = A exp(d^2 (1/c1^2 - 1/c2^2) / 2) grows with d
// when c1 < c2, and eps/x1 grows once chi(d/c1) falls, i.e. for d > sqrt(2) c1: p decreases on [sqrt(2) c1, inf).
// So p(D) <= threshold/2 at some D >= sqrt(2) c1 implies p(d) <= threshold/2 for all d >= D.  The sweep's f32
// evaluation is within 1e-5 of the real value (k_links.hip.h, bound at inlier_probability), far inside the margin
// between threshold/2 and threshold - THRESHOLD_BAND for any threshold >= 1e-3.  No cutoff (+inf: the image's links
// are always listed) when the mixture is degenerate (c1 >= c2, ratio outside (0,1), non-finite) or the threshold tiny.
__device__ inline float cull_cutoff_of(const float4 e, float threshold);

__global__ void cull_cutoff_kernel(const float4 *em, uint32_t n_images, float threshold, float *cut_now)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_images) return;
    cut_now[i] = cull_cutoff_of(em[i], threshold);
}

// what frog_stats_publish does per image in ONE launch: the weight constants of the new mixture (k_stats.hip.h em_derive_kernel)
// and its certified cutoff (two launches of 5 + 12 us each with a gap between them, at every statistics refresh)
// `linear`: the cutoff of the LINEAR stage's list instead (cull_cutoff_linear_of below)
__device__ inline float cull_cutoff_linear_of(const EmDerived d);
__global__ void stats_publish_kernel(const float4 *em, EmDerived *emd, EmFast *emf, uint32_t n_images, float threshold, float theta, float *cut_now, int linear)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_images) return;
    const float4 e = em[i];
    // grid.y = 2: the certified cutoff and the one-exponential form's ranges are two searches of some thousand dependent
    // instructions each, per image, in front of the sweep that waits for both: side by side in different wavefronts
    if (blockIdx.y == 1) { emf[i] = em_fast_of(e, theta); return; }      // theta = threshold - THRESHOLD_BAND, or NaN: no ranges (test hook)
    const EmDerived d = em_derived_of(e);
    emd[i] = d;
    cut_now[i] = linear ? cull_cutoff_linear_of(d) : cull_cutoff_of(e, threshold);
}

// The linear stage has no threshold (imageGroup.cxx:1100-1117: every half-link enters the 18 sums with its weight), but the
// sweep's weight (k_links.hip.h inlier_probability) is EXACTLY zero from some distance on: x1 = kq1 (d2 2^(s1 d2)) with
// v_exp_f32, whose result is +0 once the true value is below half the smallest denormal, i.e. for s1 d2 <= -150; then
// p = div_fast(0, x2 + 1e-10) = +0 and w = min(pA, pB) = +0 (the other probability is >= 0 or NaN, and fminf returns the
// number), and the link adds +-0.0 to sums that are never -0.0: nothing.  The cutoff takes s1 d2 <= -160 (2^-160 is 2^-11
// smallest denormals: no rounding of the instruction's last bit reaches it; the f32 product d2 s1 is within 2^-24 of -160) and
// is at least 0.2 (d < 0.1 gives weight 1, stats.h:87).  weight = min over the two images, so the smaller cutoff decides,
// as in the deformable stage.  Not used with FROG_WEIGHT_EXACT (another arithmetic, another zero set).
// What is and is not identical to the full sweep: every listed link is evaluated as before and every left-out link would have
// added +-0.0 -- the 18 sums of an image are the same REAL numbers.  They are accumulated per lane in f64 and then over the
// wavefront, and compaction moves records to other lanes: the f64 association differs, so the sums -- and the matrices,
// which keep their f64 translation -- agree to f64 rounding (1e-16 relative), not by construction to the last bit
// (tests/test_gpu_round3.py asserts 1e-13 on the energies, 1e-12 on the matrices, one f32 ulp on the coordinates, and
// reports when they are in fact equal -- as they have been in every run so far).
__device__ inline float cull_cutoff_linear_of(const EmDerived d)
{
    const double s1 = -(double)d.s1;
    if (!(s1 > 0.0) || !(s1 < 1e30)) return __builtin_inff();          // no usable exponent: every link stays listed
    // "x1 = +0 makes the weight +0" needs p = x1 / (x1 + x2 + 1e-10) with a positive, finite denominator: a mixture whose
    // ratio is NaN or outside [0, 1] (kq1 or kq2 negative or not finite) can give 0 * inf = NaN there, and min(NaN, pB) = pB
    if (!(d.kq1 >= 0.0f) || !(d.kq2 >= 0.0f) || !(d.kq1 < __builtin_inff()) || !(d.kq2 < __builtin_inff())) return __builtin_inff();
    const float c = (float)sqrt(160.0 / s1);
    return fmaxf(c * 1.00001f + 1e-6f, 0.2f);
}

__device__ inline float cull_cutoff_of(const float4 e, float threshold)
{
    const double c1 = e.x, c2 = e.y, r = e.z;
    float out = __builtin_inff();
    if (threshold >= 1e-3f && c1 > 0.0 && c2 > c1 && r > 0.0 && r < 1.0 && c2 < 1e30) {
        const double target = 0.5 * (double)threshold;
        double lo = fmax(1.4142135623730951 * (c1 + 1e-10) * 1.000001, 0.2);      // also past the `d < 0.1 -> 1` branch
        if (mixture_probability(lo, c1, c2, r) <= target) {
            out = (float)lo;
        } else {
            double hi = lo;
            bool found = false;
            for (int k = 0; k < 200 && !found; k++) {
                hi *= 2.0;
                found = mixture_probability(hi, c1, c2, r) <= target;
            }
            if (found) {
                for (int k = 0; k < 32; k++) {           // from a factor-two bracket: 2^-32 relative, the result is stored as f32 (`hi`: certified side)
                    const double mid = 0.5 * (lo + hi);
                    if (mixture_probability(mid, c1, c2, r) <= target) hi = mid; else lo = mid;
                }
                out = (float)hi;
            }
        }
        // round up: (float) may have rounded down
        if (out < __builtin_inff()) out = out * 1.000001f + 1e-6f;
    }
    return out;
}

__global__ void cull_list_cutoff_kernel(const float *cut_now, uint32_t n_images, float scale, float pad, float *cut_list)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_images) cut_list[i] = scale * cut_now[i] + pad;          // inf stays inf
}

// Compacts the records of every (tile, partner group) range whose end points are closer than the list cutoff into
// act_recs, at the same offsets and in the same chunked / transposed storage as the full array (ctx.h REC_CHUNK), in
// their order; act_cnt[tile][group] = how many.  One wavefront per range, as the sweep; grid.y = sub-pass.  The
// distance is the sweep's own expression (f32, no contraction).  Runs when the list is (re)built only: no pipeline.
template <bool WIDE>
__global__ __launch_bounds__(256) void cull_build_kernel(const SweepArgs a, const float *cut_list, void *act_recs, uint32_t *act_cnt)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t xcd = blockIdx.x % N_XCD;
    const uint32_t grp = blockIdx.y * N_XCD + xcd;
    const uint32_t t = (blockIdx.x / N_XCD) * 4 + wave;
    if (t >= a.n_tiles) return;
    const Tile &tl = a.tiles[t];
    const uint32_t pt_begin = tl.pt_begin, image = tl.image;
    const uint32_t rec_lo = tl.rec_begin + tl.group_off[grp], rec_n = tl.group_cnt[grp];
    const uint32_t g_first = a.group_begin[grp];
    const float cutA = cut_list[image];
    using Rec = std::conditional_t<WIDE, unsigned long long, unsigned int>;
    const Rec *src = reinterpret_cast<const Rec *>(a.recs);
    Rec *dst = reinterpret_cast<Rec *>(act_recs);
    auto phys = [&](uint32_t k) { return (size_t)rec_lo + (k / REC_CHUNK) * REC_CHUNK + (k % 64u) * 2u + (k % REC_CHUNK) / 64u; };
    // Does any step of the LISTED range (64 consecutive listed records = the lanes of one sweep step) hold two records of
    // the same own point?  last_step[point] = the step its latest listed record went to; ds_wrxchg is serialised per
    // address, so two lanes of one trip that carry the same point see each other.  Ranges without such a step (records are
    // partner-major then point order: all of them, unless a point has several links into one partner image or a tile
    // has fewer than 64 links into a partner image) are swept without the lane election (CULL_DUP_BIT clear).
    __shared__ uint32_t last_step_s[4][TILE_POINTS];
    uint32_t *last_step = last_step_s[wave];
    for (int k = lane; k < TILE_POINTS; k += 64) last_step[k] = 0xFFFFFFFFu;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool dup = false;
    uint32_t base = 0;
    // UNROLL steps of 64 records per trip, all their loads issued before the first is used (a dependent chain of
    // record -> two coordinates per step left the wavefront idle most of the time: 0.77 ms for 1e8 records)
    constexpr int UNROLL = 4;
    for (uint32_t k0 = 0; k0 < rec_n; k0 += 64 * UNROLL) {
        Rec rq[UNROLL];
        bool have[UNROLL];
        #pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t k = k0 + 64 * u + lane;
            have[u] = k < rec_n;
            rq[u] = have[u] ? src[phys(k)] : (Rec)0;
        }
        P3 pa[UNROLL], pb[UNROLL];
        uint32_t imgB[UNROLL];
        #pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t ia = (uint32_t)rq[u] & 0xFFu;
            uint32_t pb_index;
            if constexpr (WIDE) {
                imgB[u] = (uint32_t)rq[u] >> 8;
                pb_index = (uint32_t)(rq[u] >> 32);
            } else {
                imgB[u] = g_first + __builtin_amdgcn_ubfe((uint32_t)rq[u], 8u, a.img_bits);
                pb_index = a.poff[imgB[u]] + ((uint32_t)rq[u] >> (8u + a.img_bits));
            }
            // a null record (lanes past the end) decodes to a valid point: loaded, not used
            pa[u] = a.pos2[pt_begin + ia];
            pb[u] = a.pos2[min(pb_index, a.point_last)];
        }
        #pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const float dx = pb[u].x - pa[u].x, dy = pb[u].y - pa[u].y, dz = pb[u].z - pa[u].z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            const float cut = have[u] ? fminf(cutA, cut_list[imgB[u]]) : 0.f;
            const bool keep = have[u] && !(d2 >= cut * cut);  // as the sweep's own list-writing form (k_links.hip.h BUILD)
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const uint32_t to = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                dst[phys(to)] = rq[u];
                dup |= atomicExch(&last_step[(uint32_t)rq[u] & 0xFFu], to >> 6) == (to >> 6);
            }
            base += (uint32_t)__popcll(m);
        }
    }
    const bool any_dup = __ballot(dup) != 0ull;
    if (lane == 0) act_cnt[(size_t)t * a.n_groups + grp] = base | (any_dup ? CULL_DUP_BIT : 0u);
    // null records as far as a listed sweep's run-ahead reaches behind the list (k_links.hip.h, the sweep's own list-writing form)
    const uint32_t cap = (rec_n + REC_CHUNK - 1) / REC_CHUNK * REC_CHUNK;
    const uint32_t reach = (((base + 63u) / 64u + BODY_STEPS - 1u) / BODY_STEPS * BODY_STEPS + PT_AHEAD) * 64u;
    for (uint32_t k = base + lane; k < min(cap, reach); k += 64) dst[phys(k)] = (Rec)0;
}

// Largest distance of a point from where it was at build time, as per-block maxima (f32 bits: non-negative floats order
// like unsigned integers; a NaN compares above every number and invalidates the list, as it should).  No atomics: 30 000
// device-scope atomic maxima on a handful of addresses cost the B-spline transform 100 us when it produced these (it
// does, for contexts that own every moving point: k_grid.hip.h); every block owns one slot.  grid = (chunks of
// CULL_BLOCK_POINTS points, image); a chunk past the end of its image stores 0.
constexpr int CULL_BLOCK_POINTS = 2048;

__global__ __launch_bounds__(256) void cull_disp_kernel(const P3 *pos2, const P3 *snap, const uint32_t *poff, uint32_t *disp_part)
{
    __shared__ uint32_t sh[4];
    const uint32_t img = blockIdx.y;
    const uint32_t p0 = poff[img] + blockIdx.x * CULL_BLOCK_POINTS, pe = poff[img + 1];
    const uint32_t p1 = min(p0 + (uint32_t)CULL_BLOCK_POINTS, pe);
    uint32_t m = 0;
    for (uint32_t p = p0 + threadIdx.x; p < p1 && p0 < pe; p += 256) {
        const P3 u = pos2[p], v = snap[p];
        const float dx = u.x - v.x, dy = u.y - v.y, dz = u.z - v.z;
        const float d = __builtin_sqrtf(dx * dx + dy * dy + dz * dz);
        m = max(m, __float_as_uint(d) & 0x7FFFFFFFu);
    }
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, off, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) disp_part[blockIdx.y * gridDim.x + blockIdx.x] = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
}

// allow[0] = the largest displacement D (of any point, since the list was built) for which the list is still good:
// the condition of cull_validate_kernel below solved for D,  min over images of (cut_list 0.99999 - cut_now 1.0001 - 0.01) / 2
// (images without a cutoff, cut_now = cut_list = inf, allow anything).  Recomputed whenever the mixtures or the list change;
// the B-spline transform, which measures the displacement of its own block of points anyway, compares with it and raises
// state[0] itself -- the stand-alone check then only runs when the cutoffs have changed.  One block.
__global__ __launch_bounds__(256) void cull_allow_kernel(const float *cut_now, const float *cut_list, uint32_t n_images, float *allow)
{
    __shared__ float sh[256];
    float a = __builtin_inff();
    for (uint32_t i = threadIdx.x; i < n_images; i += 256) {
        const float now = cut_now[i], list = cut_list[i];
        if (now == __builtin_inff() && list == __builtin_inff()) continue;
        const float v = 0.5f * (list * 0.99999f - (now * 1.0001f + 0.01f));
        a = (v < a) ? v : (v == v ? a : -1.0f);                 // NaN: nothing is allowed
    }
    sh[threadIdx.x] = a;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) sh[threadIdx.x] = fminf(sh[threadIdx.x], sh[threadIdx.x + h]);
        __syncthreads();
    }
    // a shade below the exact bound: the validate kernel adds 2 D in f32, this one divides by 2 -- one ulp of slack
    if (threadIdx.x == 0) allow[0] = sh[0] - fabsf(sh[0]) * 1e-6f;
}

// cull_allow_kernel followed by cull_validate_kernel in one launch (one block): what cull_prepare runs whenever the cutoffs
// or the list have changed -- at every statistics refresh.
__global__ __launch_bounds__(256) void cull_allow_validate_kernel(const float *cut_now, const float *cut_list, const uint32_t *disp_part,
                                                                  uint32_t n_part, uint32_t n_images, float *allow, uint32_t *state);

// state[0] = 1 when some left-out link could have come within its images' cutoff (the sweep then walks all records
// and the host rebuilds the list), else 0.  With D = the largest displacement of any point since the build, a left-out
// link (d_build >= cut_list[k] for k = A or B) is now at d >= cut_list[k] - 2 D, so the list is good while
//     cut_list[k] - 2 D >= cut_now[k]   for every image k
// (+ margin: 1e-4 relative and 0.01 absolute on the cutoff, 1e-5 relative on the list cutoff, for the f32 roundings of the
// distances involved -- a few ulps of coordinates that are < 1e5 in any unit a medical image uses).  One block.
__global__ __launch_bounds__(256) void cull_validate_kernel(const float *cut_now, const float *cut_list, const uint32_t *disp_part,
                                                            uint32_t n_part, uint32_t n_images, uint32_t *state)
{
    __shared__ uint32_t shm[256];
    __shared__ int bad_s;
    uint32_t mx = 0;
    for (uint32_t i = threadIdx.x; i < n_part; i += 256) mx = max(mx, disp_part[i]);
    shm[threadIdx.x] = mx;
    if (threadIdx.x == 0) bad_s = 0;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) shm[threadIdx.x] = max(shm[threadIdx.x], shm[threadIdx.x + h]);
        __syncthreads();
    }
    const float dispmax = __uint_as_float(shm[0]);          // NaN bits (> 0x7F800000) are the maximum when present
    int bad = 0;
    for (uint32_t i = threadIdx.x; i < n_images; i += 256) {
        const float need = cut_now[i] * 1.0001f + 0.01f + 2.0f * dispmax;        // inf when the image has no cutoff
        const float have = cut_list[i] * 0.99999f;
        if (!(need <= have)) bad = 1;                                           // also catches NaN
    }
    if (bad) atomicOr(&bad_s, 1);
    __syncthreads();
    if (threadIdx.x == 0) state[0] = bad_s ? 1u : 0u;
}

__global__ __launch_bounds__(256) void cull_allow_validate_kernel(const float *cut_now, const float *cut_list, const uint32_t *disp_part,
                                                                  uint32_t n_part, uint32_t n_images, float *allow, uint32_t *state)
{
    // the two kernels' bodies, one after the other, on the same 256 threads (their expressions unchanged: same bits)
    __shared__ float sh[256];
    __shared__ uint32_t shm[256];
    __shared__ int bad_s;
    float a = __builtin_inff();
    for (uint32_t i = threadIdx.x; i < n_images; i += 256) {
        const float now = cut_now[i], list = cut_list[i];
        if (now == __builtin_inff() && list == __builtin_inff()) continue;
        const float v = 0.5f * (list * 0.99999f - (now * 1.0001f + 0.01f));
        a = (v < a) ? v : (v == v ? a : -1.0f);
    }
    uint32_t mx = 0;
    for (uint32_t i = threadIdx.x; i < n_part; i += 256) mx = max(mx, disp_part[i]);
    sh[threadIdx.x] = a;
    shm[threadIdx.x] = mx;
    if (threadIdx.x == 0) bad_s = 0;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) {
            sh[threadIdx.x] = fminf(sh[threadIdx.x], sh[threadIdx.x + h]);
            shm[threadIdx.x] = max(shm[threadIdx.x], shm[threadIdx.x + h]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) allow[0] = sh[0] - fabsf(sh[0]) * 1e-6f;
    const float dispmax = __uint_as_float(shm[0]);
    int bad = 0;
    for (uint32_t i = threadIdx.x; i < n_images; i += 256) {
        const float need = cut_now[i] * 1.0001f + 0.01f + 2.0f * dispmax;
        const float have = cut_list[i] * 0.99999f;
        if (!(need <= have)) bad = 1;
    }
    if (bad) atomicOr(&bad_s, 1);
    __syncthreads();
    if (threadIdx.x == 0) state[0] = bad_s ? 1u : 0u;
}

// total of the listed records of a list (act_cnt without the flag bit) added to out[0] (u64, zeroed by the caller); every block
// takes a slice and adds its sum with one atomic (integers: the order does not matter)
constexpr int CULL_COUNT_BLOCKS = 64;
__global__ __launch_bounds__(256) void cull_count_kernel(const uint32_t *act_cnt, uint32_t n, unsigned long long *out)
{
    __shared__ unsigned long long sh[256];
    unsigned long long v = 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) v += act_cnt[i] & ~CULL_DUP_BIT;
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) sh[threadIdx.x] += sh[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0 && sh[0]) atomicAdd(out, sh[0]);
}

} // namespace frog
