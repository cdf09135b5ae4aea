// k_links.hip.h -- the half-link sweep (K3 / K6 / K14 of SURVEY.md section 2.2) and
// the per-image linear update (K4).
//
// One wavefront per (tile, partner group) (ctx.h, prep.h): 64 half-links per step, records
// (4 bytes each when the field widths allow, else 8) streamed from HBM as coalesced non-temporal
// loads of two steps at a time, the partner point gathered as a 12-byte load from the packed xyz2
// table, the own point and the partner image's constants read from LDS; a block only ever
// touches the coordinates of ONE partner group, which stay in its XCD's L2.  Nothing here is
// GEMM shaped: it is a gather + weighted reduction, no MFMA.  What bounds it (rocprofv3 and the A/B builds of
// DESIGN.md sections 4, 4j): the texture path's rate for a step's 64 scattered 12-byte gathers.  Not the vector ALU --
// the deformable sweeps' weight went from two inlier probabilities (a third of a step's vector instructions) to one
// exponential in round 5 for 2 % -- and not the number of loads in flight (one more step of run-ahead: 10 % slower).
//
// Arithmetic contract (reference lines in the comments):
//   dist2, dist  -- f32, no FMA contraction, correctly rounded sqrt: bit-exact
//                   with the reference, so `d < 0.1` and the sample values agree.
//   inlier weight -- f32 from d2 with precomputed constants: linear and counting sweeps v_exp_f32 and div_fast for
//                   each image's probability (inlier_probability), deformable sweeps ONE exponential for the smaller
//                   of the two inside certified ranges (inlier_weight_pair); either way within 2^-16 of the
//                   reference's mixed f32/f64 getInlierProbability (stats.h:84-92), bounds derived at the two
//                   functions; weights within 1e-4 of the inlier threshold are recomputed with the
//                   reference's own arithmetic, so no decision depends on a fast form.
//   linear sums  -- f32 products, f64 accumulation (imageGroup.cxx:1102-1117);
//                   per-tile partials reduced in a fixed order (deterministic).
//   deformable   -- f32 products and f32 per-point running sums in partner order,
//                   exactly as imageGroup.cxx:270-278, in LDS (race-free
//                   read-add-write, see the loop).
#pragma once

#include "ctx.h"

#include <type_traits>

namespace frog {

enum { SWEEP_LINEAR = 0, SWEEP_DEFORMABLE = 1, SWEEP_COUNT = 2 };
#ifndef FROG_W1
#define FROG_W1 1       // 0: the deformable sweeps evaluate inlier_probability twice per half-link, as the other sweeps do (A/B builds)
#endif
// Prefetch pipeline of the sweep: the record stream runs CHUNK_AHEAD chunks (of two steps)
// ahead of the arithmetic, the partner-point gather PT_AHEAD steps; the loop is unrolled over
// one period of both rings.
// (Deeper rings -- gathers three steps ahead, record chunks three, an eight-step body, 80 registers with two spilled -- were
// measured in round 5: every sweep 10 % slower, 0.259 against 0.236 ms for the fused one.  More requests in flight do not help
// a texture path that is the busy unit.)
constexpr int CHUNK_RING = 3, CHUNK_AHEAD = 2;
constexpr int PT_RING = 3, PT_AHEAD = 2;
constexpr int BODY_STEPS = 6;
static_assert(BODY_STEPS == 2 * CHUNK_RING && BODY_STEPS % PT_RING == 0 && PT_AHEAD < PT_RING && CHUNK_AHEAD < CHUNK_RING
              && 2 * CHUNK_AHEAD >= PT_AHEAD + 2, "ring periods must divide the unrolled body; a gather needs its record");
constexpr int EMD_LDS_IMAGES = 256;     // partner groups up to this many images keep their constants in LDS
// act_cnt: set when some step of the listed range holds two records of one own point (k_cull.hip.h); a range without
// one is swept without the lane election
constexpr uint32_t CULL_DUP_BIT = 0x80000000u;
constexpr int OWNER_WORDS = TILE_POINTS / 2;      // election words per wavefront (deformable sweep)
static_assert((OWNER_WORDS & (OWNER_WORDS - 1)) == 0, "the election word of a point is its index masked");
static_assert(EMD_LDS_IMAGES == 1 << 8, "prep.h admits narrow records for img_bits <= 8 only");
constexpr int LINEAR_SUMS = 18;     // sDisp3 sPosA3 sPosB3 sPosA2_3 sPosB2_3 sWeight sDistances sWeights

struct SweepArgs {
    const Tile *tiles;
    const void *recs;           // LinkRec (wide) or uint32_t (narrow) records, ctx.h
    const uint32_t *poff;       // [n_images + 1] first point of every image
    uint32_t img_bits;          // narrow records: bits of the partner image field
    uint32_t lds_images;        // entries of the block's LDS tables of partner-image constants (sweep_lds_images)
    uint32_t point_last;        // index of the last point of the model
    const P3 *pos2;
    const EmDerived *emd;
    const EmFast *emf;          // deformable sweeps: the one-exponential form (inlier_weight_pair)
    const float4 *em;           // (c1, c2, ratio) per image: the exact re-evaluation near the inlier threshold
    uint32_t n_tiles;
    uint32_t rec2_last;         // index of the last record PAIR (16 bytes) of the context (prefetch clamp)
    float threshold;
    float band;                 // weights within this of the threshold are recomputed with the reference's arithmetic (THRESHOLD_BAND;
                                // +inf with FROG_WEIGHT_EXACT=1, a test hook: every weight, also the linear sweep's)
    double *tile_partial;       // [n_tiles][n_groups][18] (linear) or [..][2] (deformable)
    long long *tile_counts;     // [n_tiles][n_groups][2]  (count)
    float4 *group_sums;         // [N_XCD][own points]  (deformable, one block per (4 tiles, group))
    float4 *point_sums;         // [P]  (deformable, FUSED: one block per tile adds its 8 group sums itself)
    const uint32_t *tile_order; // FUSED: block -> tile (0xFFFFFFFF: no tile), dealt so that block % 8 = the tile's octant of its image
    const Tile *tiles_bo;       // FUSED: the same tiles in BLOCK order (a zero tile where there is none): fetched side by side with
                                // tile_order instead of behind it
    uint32_t own_pt_begin, own_points;
    uint32_t sub;               // sub-pass of this launch
    uint32_t n_groups;
    uint32_t group_begin[MAX_GROUPS + 1];   // first image of every partner group
    // certified outlier culling (k_cull.hip.h), deformable sweep only; all three null = walk every record
    const void *act_recs;       // the listed records, same offsets and storage as `recs`
    const uint32_t *act_cnt;    // [n_tiles][n_groups] listed records per range | CULL_DUP_BIT
    const uint32_t *cull_state; // [0] != 0: the list is not valid for the current coordinates -> walk every record
    // BUILD launches (the sweep that walks every record also writes the next list, k_cull.hip.h): per-image list cutoffs
    // and the list's storage
    const float *cut_list;
    void *build_recs;
    uint32_t *build_cnt;
};

// The sweep is bound by vector-instruction issue (rocprofv3), and the compiler's generic
// expansions of sqrtf, expf and the f32 division carry range handling this kernel cannot need: the helpers below do
// without it.

// Correctly rounded sqrt: v_sqrt_f32 (1 ulp) and the two one-ulp corrections.  Dropped: the
// rescaling of denormal arguments (a denormal d2 means d < 0.1, where the weight is 1).
__device__ __forceinline__ float sqrt_rn(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const float lo = __uint_as_float(__float_as_uint(s) - 1u), hi = __uint_as_float(__float_as_uint(s) + 1u);
    const float e_lo = __builtin_fmaf(-lo, s, x), e_hi = __builtin_fmaf(-hi, s, x);
    float r = (0.0f >= e_lo) ? lo : s;
    r = (0.0f < e_hi) ? hi : r;
    return r;
}

// n / d for d >= 1e-10 and 0 <= n <= d: v_rcp_f32 (1 ulp), one Newton step on the reciprocal, one product: within 1.5 ulp
// of the quotient.  (The correctly rounded sequence is twice as long; what the weight needs is the bound below, and the
// decisions near the threshold are taken by inlier_probability_exact.)
__device__ __forceinline__ float div_fast(float n, float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);
    return n * r;
}

// getInlierProbability with the reference's own promotions (stats.h:10-16, 84-92: f32 quotients, the exp
// and one product chain in f64).  Ten times the work of the f32 form below, so it only decides the cases
// that form cannot be trusted with: weights within THRESHOLD_BAND of the inlier threshold, where a few
// f32 ulps would put a half-link on the other side of `weight < inlierThreshold` than the reference --
// a discontinuity that the solver amplifies (one link entering or leaving a point's sums moves a weakly
// supported control point by 1e-4 of the lattice's range).
constexpr float THRESHOLD_BAND = 1e-4f;
__device__ __forceinline__ float chi_pdf_exact(float x)
{
    const float c = 0.797884560802865f;
    const float x2 = x * x;
    return (float)((double)(c * x2) * exp(-0.5 * (double)x2));
}
__device__ __noinline__ float inlier_probability_exact(float d, const float4 em)
{
    const float eps = 1e-10f;
    if (d < 0.1f) return 1.0f;
    const float c1 = em.x + eps, c2 = em.y + eps;
    const float x1 = em.z * chi_pdf_exact(d / c1) / c1;
    const float x2 = (float)((1.0 - (double)em.z) * (double)chi_pdf_exact(d / c2) / (double)c2);
    return x1 / (x1 + x2 + eps);
}

// getInlierProbability (stats.h:84-92) from precomputed per-image constants (EmDerived, ctx.h), as a function of the
// SQUARED distance (the sweeps of the deformable stage never need the distance itself: the energy there is sum w2 d2,
// imageGroup.cxx:275):   x_k = kq_k d2 2^(s_k d2),   kq_k = ratio_k c0 / c_k^3,   s_k = -log2(e) / (2 c_k^2)
// -- eleven vector instructions for one probability (the product x log2(e) used to be formed in two words and the
// argument passed through (d / c)^2 first: twenty-two; the error analysis below shows what the second word bought: nothing
// that the argument's own roundings had not already spent).
//
// Error bound against the reference's form (f32 quotients, f64 exp: chi_pdf_exact / inlier_probability_exact above).
// e = 2^-24 bounds the relative error of one f32 rounding, u_k = (d / c_k)^2 in real numbers, d2 the f32 squared
// distance both sides compute with the same bits, c' = fl(c + eps) on both sides.
//   reference  d = fl(sqrt d2), x = fl(d / c'), u = fl(x x): u (1 + 5e).  chi = fl(c0 u), the f64 product with exp(-u/2)
//              rounded to f32, fl(ratio chi), fl(.. / c'): four more roundings, and the exponent's 5e enters as 2.5e u
//              => X_ref = X (1 + (9 + 2.5 u) e).
//   here       inv = fl(1 / c'), q = fl(inv inv): 3e.  s = fl(q K) with K = fl(-log2(e)/2): 5e.  ph = fl(d2 s): 6e relative
//              in the exponent, i.e. ln 2 |ph| 6e = 3e u relative in 2^ph; v_exp_f32: 1 ulp = 2e (it reduces its own argument;
//              results below the normal range flush towards 0, where the weight they enter is 0 to thirty digits).
//              kq = fl(fl(fl(ratio c0) inv) q): 7e (8e for 1 - ratio).  x = fl(kq fl(d2 2^ph)): 2e.
//              => x = X (1 + (12 + 3 u) e).
//   x_k        |x_k / x_k,ref - 1| <= (21 + 5.5 u_k) e.
//   p          p = x1 / (x1 + x2 + eps): dp = p (1 - p) (dx1/x1 - dx2/x2) + two roundings of the sums + div_fast (3e)
//              => |p - p_ref| <= p (1 - p) (42 + 5.5 u1 + 5.5 u2) e + 5 e p.
// p (1 - p) is only non-negligible where x1 ~ x2, i.e. u1/2 ~ 3 ln(c2 / c1) + ln((1 - r) / r): u1 <= 64 covers c2/c1 up to
// 4 10^4 at any ratio in [1e-6, 1 - 1e-6]; beyond it p <= exp(-u1 / 2) (c2/c1)^3 r / (1 - r) is itself below 1e-9.  With
// u2 <= u1: |p - p_ref| <= 0.25 (42 + 704) e + 5 e = 192 e = 1.14e-5 in the worst case; INLIER_PROBABILITY_BOUND = 2^-16
// rounds that up.  tests/test_gpu_round2.py::test_inlier_probability_against_the_reference_build evaluates this function
// on the device against the reference build of stats.cxx over d/c1 in [0.02, 60] for a set of mixtures and asserts the
// bound (the observed maximum is in DESIGN.md section 2; the exact form reproduces the reference build on every value).
// The bound is what THRESHOLD_BAND (1e-4, six times larger) relies on: a weight farther than the band from the threshold
// is on the same side of it as the reference's, a weight inside the band is recomputed with the reference's own arithmetic.
// Finite for every finite d2: c' >= 1e-10 keeps kq <= 8e29 and s finite; 2^(-inf) = 0 and d2 * 0 = 0.
constexpr float INLIER_PROBABILITY_BOUND = 1.52587890625e-05f;      // 2^-16
// `d < 0.1` (stats.h:87, d = the correctly rounded f32 sqrt of d2) in terms of d2: sqrt is monotone, and 0x3c23d70a
// (0.01f) is the smallest f32 whose square root rounds to >= 0.1f (checked over the 40 neighbouring floats).
constexpr float D2_FIX = __builtin_bit_cast(float, 0x3c23d70au);
__device__ __forceinline__ float inlier_probability(float d2, const EmDerived e)
{
    const float e1 = __builtin_amdgcn_exp2f(d2 * e.s1), e2 = __builtin_amdgcn_exp2f(d2 * e.s2);
    const float x1 = e.kq1 * (d2 * e1), x2 = e.kq2 * (d2 * e2);
    const float p = div_fast(x1, x1 + x2 + 1e-10f);
    return d2 < D2_FIX ? 1.0f : p;
}

// The deformable sweeps' weight, min(pA, pB), from ONE exponential and one reciprocal (ctx.h EmFast): without the `+ eps`
// of its denominator p_k = 1 / (1 + 2^(l_k + ds_k d2)), a decreasing function of its exponent, so the smaller probability
// is the one with the larger exponent -- whichever way each exponent depends on d2.  (The sweep spent a third of its vector
// instructions on four exponentials and two divisions per half-link; measured)
// What the missing eps costs, and where the form is used:
//   * p_noeps >= p_eps always.  A weight below threshold - band in this form is therefore below it in the reference's form
//     too (bound below): the link is an outlier in both, and nothing else about it is needed (imageGroup.cxx:268).  (The
//     reference's `d < 0.1 -> 1` is no exception: em_fast_of gives an image constants only if its one-exponential value is
//     above threshold - band for every d2 <= 0.01; such a link is then not dropped, is outside every range, and takes the
//     general form, which has the rule.)
//   * otherwise the weight's VALUE matters, and the form is used when d2 lies in both images' ranges [lo, hi]
//     (k_stats.hip.h em_fast_of: there p_noeps - p_eps <= EM_FAST_EPS_DROP = 7e-6 = 117 e is certain); a lane outside them
//     evaluates inlier_probability for both images, as the sweep always did (`in_range` false: also for a NaN distance, for
//     d < 0.1, and for mixtures the form is not derived for, whose range is empty).
// Error against the reference's form, in range (e = 2^-24; p_ref as analysed at inlier_probability above deviates from the
// real-valued p_eps by p (1 - p) (9 + 2.5 u1 + 9 + 2.5 u2) e + 3e <= 0.25 (18 + 320) e + 3e = 88 e for u <= 64):
//   l, ds     one rounding each of f64 values: the exponent E = l + ds d2 carries |l| e + |ds d2| e, and the fma one more
//             rounding of E itself: dE <= (|l| + |ds d2| + |E|) e <= (2 |l| + 2 |ds| d2) e.  |l| <= 66 (em_fast_of gives no range
//             otherwise; log2((1 - r) / r) + 3 log2(c2'/c1') for r in [1e-6, 1 - 1e-6] and c2/c1 <= 4e4); |ds| d2 <= 0.72 u1 <= 46:
//             dE <= 224 e.
//   2^E       v_exp_f32: 1 ulp; with dE: t = 2^E (1 + ln 2 dE + 2e) = 2^E (1 + 158 e)
//   1 + t, v_rcp_f32 (1 ulp): 3e more on p.
//   =>        |p_fast - p_noeps| <= p (1 - p) 158 e + 3e <= 43 e, 0 <= p_noeps - p_eps <= 117 e, |p_eps - p_ref| <= 88 e:
//             |p_fast - p_ref| <= 248 e < INLIER_PROBABILITY_BOUND = 2^-16 = 256 e, the bound THRESHOLD_BAND relies on.
//   outside the range only the one-sided statement is used: p_ref <= p_eps + 88 e <= p_noeps + 88 e <= p_fast + 131 e, so
//             p_fast < threshold - 1e-4 puts p_ref below the threshold by 9e-5.
// min(pA, pB): |min(a, b) - min(a', b')| <= max(|a - a'|, |b - b'|), and min(a, b) <= min(a', b') + c when a <= a' + c, b <= b' + c:
// the same statements for the weight.  tests/test_gpu_round2.py::test_inlier_weight_pair_against_the_reference_build.
// `b` carries the pair's range: [max(lo_a, lo_b), min(hi_a, hi_b)]
__device__ __forceinline__ float inlier_weight_pair(float d2, const EmFast a, const EmFast b, bool &in_range)
{
    const float e = fmaxf(__builtin_fmaf(a.ds, d2, a.l), __builtin_fmaf(b.ds, d2, b.l));
    in_range = d2 >= b.lo && d2 <= b.hi;                // false for a NaN distance and for empty ranges
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(e));
}

__device__ __forceinline__ double wave_sum(double v)
{
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ long long wave_sum_ll(long long v)
{
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// EMD_LDS: the partner group's EM constants fit emd_s (always, unless a group holds more than
// EMD_LDS_IMAGES images).  A template parameter and not a run-time select: a pointer that may
// be LDS or global compiles to a flat load, whose completion can only be awaited with
// vmcnt(0) -- which would also wait for the gathers just issued for later steps.
// WIDE: 8-byte records (ctx.h LinkRec); otherwise the 4-byte form.
// BUILD (deformable sweep, narrow records): while walking EVERY record of its range the wavefront also writes the culling
// list for the coordinates it reads -- what cull_build_kernel does in a pass of its own (0.6 ms for 1e8 records: a record,
// two coordinates and a distance per half-link, all of which this kernel has in hand anyway).
#ifdef FROG_W1_COUNT
__device__ unsigned long long g_w1_count[8];    // steps, steps with a lane in the general form, lanes, lanes in the general form, lanes skipped as outliers, general-form lanes below lo
#endif
#ifdef FROG_SWEEP_TRACE
// per (block, wavefront) of the last fused steady-state launch: wall_clock64 at kernel entry, tile known, staging barrier passed,
// first step done, walk done, final barrier passed, end; records walked (scripts/microbench/sweep_trace_an.py)
__device__ unsigned long long g_sweep_trace[8 * 8 * 16384];
#define FROG_TR(slot) do { if constexpr (FUSED && !BUILD) { if (lane == 0 && blockIdx.x < 16384) g_sweep_trace[((size_t)blockIdx.x * 8 + wave) * 8 + (slot)] = wall_clock64(); } } while (0)
#else
#define FROG_TR(slot) do { } while (0)
#endif

// FUSED (deformable sweep, one launch per pass): a block is ONE tile and all 8 partner groups, wavefront w = group w.  The
// eight per-group sums of a point are then in one block's LDS when the walk ends: they are added there, in group order --
// the same additions in the same order as combine_groups_kernel / the scatter's point phase, i.e. the same bits -- and ONE
// float4 per point goes to memory (32 MB instead of the 252 MB of per-group partial sums that the scatter then reads
// back, half of every line unused: 0.85 of an iteration's 2.0 GB of fabric traffic).  What it gives
// up is "one partner group per XCD": a block gathers from all partner images.  The coordinates an XCD touches are kept
// small the other way instead: blocks are dealt so that block % 8 = the tile's eighth of its image along the Morton curve
// (tile_order), and a true match's partner lies in the same eighth of the partner image -- with the certified outliers
// gone from the list (k_cull.hip.h) nearly every gather is a true match.
template <int MODE, bool EMD_LDS, bool WIDE, bool BUILD = false, bool FUSED = false>
#ifndef FROG_FUSED_MIN_WAVES
#define FROG_FUSED_MIN_WAVES 1      // 7 caps the kernel at 72 VGPRs (six wavefronts per SIMD and room for the side stream's selection kernel beside them): measured slower, 0.260 against 0.254 ms
#endif
__global__ __launch_bounds__(FUSED ? 512 : 256, FUSED ? FROG_FUSED_MIN_WAVES : 1) void sweep_kernel(const SweepArgs a)
{
    static_assert(!BUILD || (MODE != SWEEP_COUNT && EMD_LDS && !WIDE), "the list is built by the narrow linear / deformable sweeps");
    static_assert(!FUSED || (MODE == SWEEP_DEFORMABLE && EMD_LDS), "the fused form is the deformable sweep's");
    constexpr int WAVES = FUSED ? 8 : 4;                // wavefronts per block
    constexpr int OWN_TILES = FUSED ? 1 : 4;            // tiles whose own points the block stages
    // per-wave accumulators: (sDisp xyz, sWeight) of every point of the tile, f32 like the
    // reference's (imageGroup.cxx:256-257), plus one ownership word per point
    __shared__ float4 acc[(MODE == SWEEP_DEFORMABLE) ? WAVES * TILE_POINTS : 1];
    // one election word per pair of points (k, k + TILE_POINTS / 2): the lanes of a step hold a stretch of about
    // half a tile's points (records are partner-major, then point order), so the two rarely meet -- and when they do,
    // the higher lane waits one more round, nothing else
    __shared__ unsigned int owner[(MODE == SWEEP_DEFORMABLE) ? WAVES * OWNER_WORDS : 1];
    // Scattered vector loads cost ~48 CU-cycles per wave-instruction through the texture
    // path even when they hit L1 (measured: the sweep takes 0.26 ms without its three
    // gathers per step, 0.68 ms with them), so everything that can be staged is: the xyz2
    // of the tile's own points (per wave; per block when FUSED) and the EM constants of the partner group's
    // images (per block; per wave when FUSED) live in LDS; only the partner point is gathered from memory.
    // one array, x / y / z planes a constant distance apart: one address computation per step, the planes are DS offsets
    __shared__ float own_xyz[3 * OWN_TILES * TILE_POINTS];
    // a.lds_images entries each, sized at launch (sweep_lds_images: a power of two >= the largest group): with the
    // 30 KB above, a block stays under 32 KB and FIVE blocks share a CU's 160 KB -- the sweep's time follows its
    // resident wavefronts almost in proportion (measured: 12 instead of 16 per CU, +21 %).  FUSED: one set of tables per
    // wavefront (its group's), 41 KB per block of eight wavefronts, three blocks per CU.
    extern __shared__ __align__(16) unsigned char sweep_dyn_lds[];
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    unsigned char *tables = sweep_dyn_lds + (FUSED ? (size_t)wave * a.lds_images * (sizeof(EmDerived) + 2 * sizeof(uint32_t)) : (size_t)0);
    EmDerived *emd_s = reinterpret_cast<EmDerived *>(tables);
    EmFast *emf_s = reinterpret_cast<EmFast *>(tables);            // W1: the deformable sweeps keep the one-exponential form here instead
    static_assert(sizeof(EmFast) == sizeof(EmDerived), "the two tables share their place");
    constexpr bool W1 = FROG_W1 && MODE == SWEEP_DEFORMABLE;
    uint32_t *img_base_s = reinterpret_cast<uint32_t *>(tables + (size_t)a.lds_images * sizeof(EmDerived));   // narrow records: first point of the group's images
    float *cut_s = reinterpret_cast<float *>(img_base_s + a.lds_images);      // BUILD: list cutoff of the group's images
    __shared__ uint32_t last_step_s[BUILD ? WAVES * TILE_POINTS : 1];         // BUILD: see cull_build_kernel

    // One launch per sub-pass.  block -> (4 consecutive tiles, XCD): block % 8 is the XCD the
    // block lands on under round-robin dispatch (a performance assumption only) and selects
    // the partner group sub*8 + xcd it reads, so during a launch an XCD's L2 only has to hold
    // 1/n_groups of the coordinate table.  Later sub-passes continue the per-XCD partial sums.
    // FUSED: block -> one tile (tile_order), wavefront -> partner group.
    FROG_TR(0);
    const uint32_t xcd = FUSED ? (uint32_t)wave : blockIdx.x % N_XCD;
    const uint32_t grp = FUSED ? (uint32_t)wave : a.sub * N_XCD + xcd;
    const uint32_t t = FUSED ? a.tile_order[blockIdx.x] : (blockIdx.x / N_XCD) * 4 + wave;
    const bool live = t < a.n_tiles;

    uint32_t pt_begin = 0, pt_count = 0, rec_lo = 0, rec_n = 0, image = 0;
    if (FUSED || live) {
        const Tile &tl = FUSED ? a.tiles_bo[blockIdx.x] : a.tiles[t];
        pt_begin = tl.pt_begin; pt_count = tl.pt_count; image = tl.image;
        rec_lo = tl.rec_begin + tl.group_off[grp];
        rec_n = tl.group_cnt[grp];
    }
    // the listed records only, when there is a list and the check before this launch found it valid (k_cull.hip.h)
    bool listed = false;
    if constexpr (MODE != SWEEP_COUNT) listed = a.act_cnt != nullptr && a.cull_state[0] == 0u;
    bool elect = true;          // wave-uniform: lanes of one step may meet on a point (always assumed for unlisted ranges)
    if (listed && live) {
        const uint32_t c = a.act_cnt[(size_t)t * a.n_groups + grp];
        rec_n = c & ~CULL_DUP_BIT;
        elect = (c & CULL_DUP_BIT) != 0u;
    }
    elect = __builtin_amdgcn_readfirstlane((uint32_t)elect) != 0u;
    rec_lo = __builtin_amdgcn_readfirstlane(rec_lo);      // the same in every lane: keep them in SGPRs
    rec_n = __builtin_amdgcn_readfirstlane(rec_n);
    pt_begin = __builtin_amdgcn_readfirstlane(pt_begin);
    pt_count = __builtin_amdgcn_readfirstlane(pt_count);
    image = __builtin_amdgcn_readfirstlane(image);
    float4 *my = acc + (MODE == SWEEP_DEFORMABLE ? wave * TILE_POINTS : 0);
    unsigned int *own = owner + (MODE == SWEEP_DEFORMABLE ? wave * OWNER_WORDS : 0);

    if (MODE == SWEEP_DEFORMABLE) {
        for (int k = lane; k < TILE_POINTS; k += 64) {
            my[k] = (a.sub > 0 && (uint32_t)k < pt_count) ? a.group_sums[group_sum_index(xcd, pt_begin - a.own_pt_begin + k, a.own_points)]
                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < OWNER_WORDS) own[k] = 0xFFFFFFFFu;
        }
    }
    // The first chunks of the record stream are asked for HERE, before the own points and the group's tables are staged: they
    // come from HBM (the longest latency of the prologue) and need nothing but the range's offset.  A block lives ~25 us and
    // used to spend its first ~5 in a chain tile -> staging -> barrier -> records -> gathers (measured: 45 % of the sweep's
    // time does not scale with the links walked; it is per block).
    typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
    typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
    using Rec = std::conditional_t<WIDE, unsigned long long, unsigned int>;
    using Chunk = std::conditional_t<WIDE, v2u64, v2u32>;          // the lane's records of two steps
    const Chunk *rec2 = reinterpret_cast<const Chunk *>(listed ? a.act_recs : a.recs);
    const uint32_t rec2_lo = rec_lo / 2u + lane;
    auto chunk_at = [&](uint32_t c) { return __builtin_nontemporal_load(rec2 + min(rec2_lo + c * (REC_CHUNK / 2), a.rec2_last)); };
    FROG_TR(1);
    Chunk cq_first[CHUNK_AHEAD];
    #pragma unroll
    for (int k = 0; k < CHUNK_AHEAD; k++) cq_first[k] = chunk_at(k);

    float *px = own_xyz + (FUSED ? 0 : wave * TILE_POINTS);
    constexpr int PLANE = OWN_TILES * TILE_POINTS;
    if constexpr (FUSED) {
        for (uint32_t k = threadIdx.x; k < pt_count; k += 64 * WAVES) {     // the block's one tile, by all its wavefronts
            const P3 p = a.pos2[pt_begin + k];
            px[k] = p.x; px[k + PLANE] = p.y; px[k + 2 * PLANE] = p.z;
        }
    } else {
        for (uint32_t k = lane; k < pt_count; k += 64) {
            const P3 p = a.pos2[pt_begin + k];
            px[k] = p.x; px[k + PLANE] = p.y; px[k + 2 * PLANE] = p.z;
        }
    }
    const uint32_t g_first = a.group_begin[grp], g_count = a.group_begin[grp + 1] - g_first;
    // the group's tables: filled by the block (one group per block) or by the wavefront (FUSED: one group per wavefront)
    const uint32_t fill_id = FUSED ? (uint32_t)lane : threadIdx.x, fill_stride = FUSED ? 64u : 256u;
    if (EMD_LDS) {
        if constexpr (W1) {
            // FUSED: the partner's constants with the PAIR's range (a fused block has ONE own image, the same for every entry
            // of its tables): two compares per step instead of four.  The per-group form's table serves four tiles, of any
            // images: the ranges are intersected in the step.
            const EmFast fO = a.emf[image];
            for (uint32_t k = fill_id; k < g_count; k += fill_stride) {
                EmFast f = a.emf[g_first + k];
                if constexpr (FUSED) { f.lo = fmaxf(f.lo, fO.lo); f.hi = fminf(f.hi, fO.hi); }
                emf_s[k] = f;
            }
        }
        else { for (uint32_t k = fill_id; k < g_count; k += fill_stride) emd_s[k] = a.emd[g_first + k]; }
    }
    if (!WIDE)
        for (uint32_t k = fill_id; k < a.lds_images; k += fill_stride) img_base_s[k] = k < g_count ? a.poff[g_first + k] : 0u;
    uint32_t *last_step = last_step_s + (BUILD ? wave * TILE_POINTS : 0);
    if constexpr (BUILD) {
        for (uint32_t k = fill_id; k < a.lds_images; k += fill_stride) cut_s[k] = k < g_count ? a.cut_list[g_first + k] : 0.f;
        for (int k = lane; k < TILE_POINTS; k += 64) last_step[k] = 0xFFFFFFFFu;
    }
    __syncthreads();
    FROG_TR(2);
    const float cutA = BUILD ? a.cut_list[image] : 0.f;
    uint32_t built = 0;                 // BUILD: records listed so far
    bool dup = false;                   //        some step of the list holds one point twice

    const EmDerived eA = a.emd[image];
    const EmFast fA = a.emf[image];

    double s[(MODE == SWEEP_LINEAR) ? LINEAR_SUMS : 2];
    #pragma unroll
    for (int k = 0; k < ((MODE == SWEEP_LINEAR) ? LINEAR_SUMS : 2); k++) s[k] = 0.0;
    long long n_in = 0, n_out = 0;

    // Software pipeline, by hand.  A step is one long dependent chain (gather -> ~150 vector
    // instructions -> LDS election), so the partner point's 12-byte gather runs PT_AHEAD steps
    // ahead and the record stream (non-temporal: read once per pass; one 16-byte load per lane
    // and chunk of two steps) CHUNK_AHEAD chunks; own point and partner constants come from
    // LDS.  Two things keep the run-ahead real:
    //  - the queues are rings indexed by compile-time constants inside an unrolled loop:
    //    rotating them through registers makes the compiler wait for every outstanding load
    //    (s_waitcnt vmcnt(0)) at the end of each step;
    //  - every load is issued unconditionally, from an index clamped into the record array: a
    //    load under a lane-dependent branch makes the number of loads in flight unknown to the
    //    compiler, which then also waits with vmcnt(0).  Records read past the end of this
    //    (tile, group) range are padding (null records: point 0) or a neighbour's: real
    //    records, whose partner index is a real point, fetched and then ignored.
    const uint32_t img_bits = a.img_bits;
    // record fields: own point in the tile, partner image (index into emd_s / img_base_s), partner point
    auto own_of = [&](Rec rq) { return (uint32_t)rq & 0xFFu; };
    auto img_of = [&](Rec rq) {
        if constexpr (WIDE) return ((uint32_t)rq >> 8) - (EMD_LDS ? g_first : 0u);
        else return __builtin_amdgcn_ubfe((uint32_t)rq, 8u, img_bits);
    };
    auto gather = [&](Rec rq) {
        if constexpr (WIDE) return a.pos2[(uint32_t)(rq >> 32)];
        // the clamp matters for records prefetched past this range only: a neighbour's record is
        // relative to ITS group's images and may decode to any index here
        else return a.pos2[min(img_base_s[img_of(rq)] + ((uint32_t)rq >> (8u + img_bits)), a.point_last)];
    };
    Chunk cq[CHUNK_RING];
    P3 pbq[PT_RING];
    #pragma unroll
    for (int k = 0; k < CHUNK_RING; k++) cq[k] = (k < CHUNK_AHEAD) ? cq_first[k < CHUNK_AHEAD ? k : 0] : Chunk{ 0, 0 };
    #pragma unroll
    for (int k = 0; k < PT_RING; k++) pbq[k] = (k < PT_AHEAD) ? gather((k & 1) ? cq[k / 2].y : cq[k / 2].x) : P3{ 0.f, 0.f, 0.f };

    auto step = [&](const Rec rq, const P3 pb, auto elect_c, const bool valid) __attribute__((always_inline)) {
        constexpr bool ELECT = decltype(elect_c)::value;
        const uint32_t ia = own_of(rq);                 // own point inside the tile
        const P3 pa = { px[ia], px[ia + PLANE], px[ia + 2 * PLANE] };
        EmDerived eB;
        EmFast fB;
        if constexpr (W1) {
            fB = EMD_LDS ? emf_s[img_of(rq)] : a.emf[img_of(rq)];
            if constexpr (!(EMD_LDS && FUSED)) { fB.lo = fmaxf(fB.lo, fA.lo); fB.hi = fminf(fB.hi, fA.hi); }  // else: done when the table was filled
        }
        else eB = EMD_LDS ? emd_s[img_of(rq)] : a.emd[img_of(rq)];

        const float dx = pb.x - pa.x, dy = pb.y - pa.y, dz = pb.z - pa.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        if constexpr (BUILD) {
            // cull_build_kernel's criterion and compaction, on the distance this step has just formed (all 64 lanes are here:
            // `valid` = the lane holds a record of the range)
            const float cut = fminf(cutA, cut_s[img_of(rq)]);
            const bool keep = valid && !(d2 >= cut * cut);  // a NaN distance is listed (the linear sweep's sums must see it, as the full sweep's do)
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const uint32_t to = built + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                reinterpret_cast<Rec *>(a.build_recs)[(size_t)rec_lo + (to / REC_CHUNK) * REC_CHUNK + (to % 64u) * 2u + (to % REC_CHUNK) / 64u] = rq;
                dup |= atomicExch(&last_step[ia], to >> 6) == (to >> 6);
            }
            built += (uint32_t)__popcll(m);
            if (!valid) return;
        }
        float w;
        if constexpr (W1) {
            bool in_range;
            w = inlier_weight_pair(d2, fA, fB, in_range);
            asm volatile("" : "+v"(w));                     // formed by every lane, in front of the branch: the compiler otherwise moves it --
                                                            // and the table read it starts with -- behind the range test, one more LDS round trip in the chain
#ifdef FROG_W1_NOMID
            in_range = true;
#endif
#ifdef FROG_W1_COUNT
            if constexpr (FUSED && !BUILD) {
                const bool skip = w < a.threshold - a.band, mid = !in_range && !skip;
                const unsigned long long all = __ballot(true), mm = __ballot(mid), sk = __ballot(skip), lo_m = __ballot(mid && d2 < fB.lo);
                if (lane == __builtin_ctzll(all)) {
                    atomicAdd(&g_w1_count[0], 1ull); atomicAdd(&g_w1_count[1], mm ? 1ull : 0ull);
                    atomicAdd(&g_w1_count[2], (unsigned long long)__popcll(all)); atomicAdd(&g_w1_count[3], (unsigned long long)__popcll(mm));
                    atomicAdd(&g_w1_count[4], (unsigned long long)__popcll(sk)); atomicAdd(&g_w1_count[5], (unsigned long long)__popcll(lo_m));
                }
            }
#endif
            if (!in_range && !(w < a.threshold - a.band)) { // rare: far tails of the inlier component, d < 0.1, degenerate mixtures
                const uint32_t imgB = img_of(rq) + (WIDE && !EMD_LDS ? 0u : g_first);
                w = fminf(inlier_probability(d2, eA), inlier_probability(d2, a.emd[imgB]));
            }
        } else w = fminf(inlier_probability(d2, eA), inlier_probability(d2, eB));
        if constexpr (MODE != SWEEP_LINEAR) {
            // the threshold decision is taken on the reference's own arithmetic when it is close
            if (fabsf(w - a.threshold) < a.band) {
                const uint32_t imgB = img_of(rq) + (WIDE && !EMD_LDS ? 0u : g_first);
                const float d = sqrt_rn(d2);
                w = fminf(inlier_probability_exact(d, a.em[image]), inlier_probability_exact(d, a.em[imgB]));
            }
        } else if (a.band > 1.0f) {                         // test hook only (wave-uniform)
            const uint32_t imgB = img_of(rq) + (WIDE && !EMD_LDS ? 0u : g_first);
            const float d = sqrt_rn(d2);
            w = fminf(inlier_probability_exact(d, a.em[image]), inlier_probability_exact(d, a.em[imgB]));
        }

        if constexpr (MODE == SWEEP_LINEAR) {
            // imageGroup.cxx:1102-1117
            const float d = sqrt_rn(d2);
            s[16] += (double)(w * w * d * d);
            s[17] += (double)(w * w);
            s[0] += (double)(w * dx); s[1] += (double)(w * dy); s[2] += (double)(w * dz);
            s[3] += (double)(w * pa.x); s[4] += (double)(w * pa.y); s[5] += (double)(w * pa.z);
            s[6] += (double)(w * pb.x); s[7] += (double)(w * pb.y); s[8] += (double)(w * pb.z);
            s[9] += (double)(w * pa.x * pa.x); s[10] += (double)(w * pa.y * pa.y); s[11] += (double)(w * pa.z * pa.z);
            s[12] += (double)(w * pb.x * pb.x); s[13] += (double)(w * pb.y * pb.y); s[14] += (double)(w * pb.z * pb.z);
            s[15] += (double)w;
        } else if constexpr (MODE == SWEEP_DEFORMABLE) {
            // imageGroup.cxx:270-278.  The tile's accumulators belong to this wavefront alone,
            // so a plain LDS read-add-write is enough once lanes that hit the SAME point in this
            // step are serialised: every pending lane bids with ds_min_u32 (integer LDS atomics
            // are full rate, ds_add_f32 is ~38x slower on gfx950), the lowest lane adds first.
            // Links are in partner order, so each point's f32 sums are formed in the reference's
            // order.  Duplicates are rare (two links of one point into one image).
            const bool inlier = w >= a.threshold;
            const float w2 = w * w;
            s[1] += (double)(inlier ? w2 : 0.0f);         // adding +0.0 leaves the f64 sums unchanged
            s[0] += (double)(inlier ? w2 * d2 : 0.0f);
            if constexpr (!ELECT) {
                // no two lanes of this step hold the same point (certified when the list was built)
                if (inlier) {
                    float4 t = my[ia];
                    t.x += w2 * dx; t.y += w2 * dy; t.z += w2 * dz; t.w += w2;
                    my[ia] = t;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next step may touch this point from another lane
                __builtin_amdgcn_wave_barrier();
                return;
            }
            bool pending = inlier;
            const uint32_t io = ia & (OWNER_WORDS - 1);
            while (__ballot(pending)) {
                if (pending) atomicMin(&own[io], (unsigned int)lane);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (pending && __hip_atomic_load(&own[io], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == (unsigned int)lane) {
                    float4 t = my[ia];
                    t.x += w2 * dx; t.y += w2 * dy; t.z += w2 * dz; t.w += w2;
                    my[ia] = t;
                    own[io] = 0xFFFFFFFFu;
                    pending = false;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            // imageGroup.cxx:1022-1027
            if (w < a.threshold) n_out++; else n_in++;
        }
    };

    const uint32_t n_steps = (rec_n + 63) / 64;          // wave-uniform
    auto walk = [&](auto elect_c) __attribute__((always_inline)) {
        for (uint32_t base = 0; base < n_steps; base += BODY_STEPS) {
            #pragma unroll
            for (int j = 0; j < BODY_STEPS; j++) {
                const uint32_t r = lane + 64 * (base + j);   // step base+j, lane -> record r of the range
                if (j % 2 == 0) cq[(j / 2 + CHUNK_AHEAD) % CHUNK_RING] = chunk_at(base / 2 + j / 2 + CHUNK_AHEAD);
                {
                    const Chunk ahead = cq[((j + PT_AHEAD) / 2) % CHUNK_RING];
                    pbq[(j + PT_AHEAD) % PT_RING] = gather(((j + PT_AHEAD) & 1) ? ahead.y : ahead.x);
                }
                if (BUILD || r < rec_n) step((j & 1) ? cq[j / 2].y : cq[j / 2].x, pbq[j % PT_RING], elect_c, r < rec_n);
#ifdef FROG_SWEEP_TRACE
                if (base == 0 && j == 0) FROG_TR(3);
#endif
            }
        }
    };
    if (MODE != SWEEP_DEFORMABLE || elect) walk(std::true_type{});
    else walk(std::false_type{});
    FROG_TR(4);

    if constexpr (BUILD) {
        // Null records behind the list's last record, as far as a listed sweep's run-ahead gathers reach (its walk goes in
        // trips of BODY_STEPS steps and gathers PT_AHEAD steps ahead): what lies there otherwise are records of an EARLIER,
        // longer list -- left-out false matches whose partners are anywhere --, and every range would pay a few steps of
        // random gathers for them (measured: the deformable sweep 0.258 -> 0.294 ms behind the linear stage's lists).
        const uint32_t cap = (rec_n + REC_CHUNK - 1) / REC_CHUNK * REC_CHUNK;
        const uint32_t reach = (((built + 63u) / 64u + BODY_STEPS - 1u) / BODY_STEPS * BODY_STEPS + PT_AHEAD) * 64u;
        for (uint32_t k = built + lane; k < min(cap, reach); k += 64)
            reinterpret_cast<Rec *>(a.build_recs)[(size_t)rec_lo + (k / REC_CHUNK) * REC_CHUNK + (k % 64u) * 2u + (k % REC_CHUNK) / 64u] = (Rec)0;
    }
    if constexpr (MODE == SWEEP_LINEAR) {
        #pragma unroll
        for (int k = 0; k < LINEAR_SUMS; k++) {
            double v = wave_sum(s[k]);
            if (lane == 0 && live) a.tile_partial[((size_t)t * a.n_groups + grp) * LINEAR_SUMS + k] = v;
        }
        if constexpr (BUILD) {
            if (lane == 0 && live) a.build_cnt[(size_t)t * a.n_groups + grp] = built | CULL_DUP_BIT;    // the linear sweep has no election to skip
        }
    } else if constexpr (MODE == SWEEP_DEFORMABLE) {
        double v0 = wave_sum(s[0]), v1 = wave_sum(s[1]);
        if (lane == 0 && live) {
            a.tile_partial[((size_t)t * a.n_groups + grp) * 2] = v0;
            a.tile_partial[((size_t)t * a.n_groups + grp) * 2 + 1] = v1;
        }
        if constexpr (BUILD) {
            const bool any_dup = __ballot(dup) != 0ull;
            if (lane == 0 && live) a.build_cnt[(size_t)t * a.n_groups + grp] = built | (any_dup ? CULL_DUP_BIT : 0u);
        }
        __syncthreads();
        FROG_TR(5);
        typedef float v4f __attribute__((ext_vector_type(4)));
        if constexpr (FUSED) {
            // the point's sums = its eight group sums added in group order (combine_groups_kernel's order: same bits)
            for (uint32_t k = threadIdx.x; k < pt_count; k += 64 * WAVES) {
                float4 v = acc[k];
                #pragma unroll
                for (int g = 1; g < N_XCD; g++) {
                    const float4 u = acc[g * TILE_POINTS + k];
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                }
                a.point_sums[pt_begin + k] = v;
            }
#ifdef FROG_SWEEP_TRACE
            if constexpr (!BUILD) { if (lane == 0 && blockIdx.x < 16384) { g_sweep_trace[((size_t)blockIdx.x * 8 + wave) * 8 + 6] = wall_clock64(); g_sweep_trace[((size_t)blockIdx.x * 8 + wave) * 8 + 7] = rec_n; } }
#endif
        } else {
            for (uint32_t k = lane; k < pt_count; k += 64) {                 // written once, read once: non-temporal
                const float4 v = my[k];
                float4 *dst = a.group_sums + group_sum_index(xcd, pt_begin - a.own_pt_begin + k, a.own_points);
                if (SUMS_POINT_MAJOR) *dst = v;                              // pieces of a line other XCDs complete: let the caches merge them
                else __builtin_nontemporal_store((v4f){ v.x, v.y, v.z, v.w }, reinterpret_cast<v4f *>(dst));
            }
        }
    } else {
        long long v0 = wave_sum_ll(n_in), v1 = wave_sum_ll(n_out);
        if (lane == 0 && live) {
            a.tile_counts[((size_t)t * a.n_groups + grp) * 2] = v0;
            a.tile_counts[((size_t)t * a.n_groups + grp) * 2 + 1] = v1;
        }
    }
}

// (sDisp, sWeight) of every owned point = sum of its N_XCD partial sums, in a fixed order.
__global__ __launch_bounds__(256) void combine_groups_kernel(const float4 *group_sums, uint32_t own_points,
                                                             uint32_t own_pt_begin, float4 *point_sums)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= own_points) return;
    float4 s = group_sums[group_sum_index(0, i, own_points)];
    #pragma unroll
    for (int g = 1; g < N_XCD; g++) {
        const float4 v = group_sums[group_sum_index(g, i, own_points)];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    point_sums[own_pt_begin + i] = s;
}

// Landmark constraints (Point::hardLinks): after the regular links of a point, each hard link adds
// weight2 * (pB - pA) and weight2 to its f32 sums, in the order stored (imageGroup.cxx:280-295).  One
// thread per constrained point (a few hundred); its energy terms go to partial[t] = (sDistances, sWeights).
__global__ void hard_links_kernel(const P3 *pos2, float4 *point_sums, const uint32_t *point, const uint32_t *ptr,
                                  const uint32_t *partner, uint32_t n_hard, float w2, double *partial)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_hard) return;
    const uint32_t a = point[t];
    const P3 pA = pos2[a];
    float4 s = point_sums[a];
    double sd = 0, sw = 0;
    for (uint32_t l = ptr[t]; l < ptr[t + 1]; l++) {
        const P3 pB = pos2[partner[l]];
        const float dx = pB.x - pA.x, dy = pB.y - pA.y, dz = pB.z - pA.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        sd += (double)(w2 * d2);
        sw += (double)w2;
        s.x += w2 * dx; s.y += w2 * dy; s.z += w2 * dz; s.w += w2;
    }
    point_sums[a] = s;
    if (partial) { partial[2 * t] = sd; partial[2 * t + 1] = sw; }
}

// adds the constrained points' energy terms, in order, to energy[0..1]
__global__ void hard_energy_kernel(const double *partial, uint32_t n_hard, double *energy)
{
    if (blockIdx.x || threadIdx.x) return;
    double a = 0, b = 0;
    for (uint32_t t = 0; t < n_hard; t++) { a += partial[2 * t]; b += partial[2 * t + 1]; }
    energy[0] += a; energy[1] += b;
}

// Sum of the (sDistances, sWeights) tile partials -> energy[0..1], in two stages with a
// fixed tree (deterministic): ENERGY_BLOCKS blocks reduce contiguous slices, then one
// block adds their results in order.
constexpr int ENERGY_BLOCKS = 64;

// One launch: the block that finishes last (a ticket counter in memory) adds the block sums in block
// order, so the result does not depend on which block that is.
__global__ __launch_bounds__(256) void energy_reduce_kernel(const double *partial, uint32_t n, int stride, int off,
                                                            double *block_sums /*[ENERGY_BLOCKS][2]*/, unsigned int *ticket,
                                                            double *energy, const uint32_t *list_invalid = nullptr,
                                                            unsigned int *stray = nullptr)
{
    __shared__ double sh[2][256];
    __shared__ bool last;
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t b = min(n, blockIdx.x * per), e = min(n, b + per);
    double a0 = 0, a1 = 0;
    for (uint32_t t = b + threadIdx.x; t < e; t += 256) {
        a0 += partial[(size_t)t * stride + off];
        a1 += partial[(size_t)t * stride + off + 1];
    }
    sh[0][threadIdx.x] = a0; sh[1][threadIdx.x] = a1;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) { sh[0][threadIdx.x] += sh[0][threadIdx.x + h]; sh[1][threadIdx.x] += sh[1][threadIdx.x + h]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&block_sums[2 * blockIdx.x], sh[0][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&block_sums[2 * blockIdx.x + 1], sh[1][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    // the last block: all block sums fetched side by side (one thread each: 128 dependent loads by one thread took
    // longer than a second launch), then added in block order by one thread
    __threadfence();
    if (threadIdx.x < 2 * gridDim.x && threadIdx.x < 256)
        sh[0][threadIdx.x] = __hip_atomic_load(&block_sums[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
        double s0 = 0, s1 = 0;
        for (unsigned int k = 0; k < gridDim.x; k++) { s0 += sh[0][2 * k]; s1 += sh[0][2 * k + 1]; }
        energy[0] = s0; energy[1] = s1; energy[2] = 0.0;
        // fourth scalar of the per-iteration read-back: the culling list needs a rebuild (k_cull.hip.h); a sum over
        // ranks when the buffer is all-reduced, any non-zero value means the same
        energy[3] = list_invalid ? (double)list_invalid[0] : 0.0;
        if (stray) stray[0] = stray[1] = 0u;    // the scatters that follow count their stray points from 0 (k_grid.hip.h: [0], [1] = even / odd steps)
        *ticket = 0u;                                       // ready for the next launch (same stream: ordered)
    }
}

// K4: per-image scale / translation update (imageGroup.cxx:1123-1143).  One block per owned
// image: 8 slices x 18 sums add the image's (tile, group) partials in a fixed order, the
// slices are combined in order, then lanes 0..2 update one axis each.
// `img_energy` (null: the energy is reduced by energy_reduce_kernel): the block also leaves its image's (sDistances, sWeights)
// -- sums 16 and 17, which it forms anyway -- in img_energy[image - image_begin], and the block that finishes last adds them
// over the owned images in image order -> energy[0..1] (fixed order: deterministic), energy[2] = 0, energy[3] = the culling
// list's flag.  One launch and one dependent-launch gap less per linear iteration.
__global__ __launch_bounds__(256) void linear_update_kernel(const double *partial, const uint32_t *img_tile_ptr,
                                                            uint32_t n_groups, uint32_t image_begin, double *mat,
                                                            float linear_alpha, int use_scale,
                                                            double *img_energy = nullptr, unsigned int *ticket = nullptr,
                                                            double *energy = nullptr, const uint32_t *list_invalid = nullptr)
{
    __shared__ double part[8][32];
    __shared__ double sums[LINEAR_SUMS];
    __shared__ bool last_s;
    const uint32_t image = image_begin + blockIdx.x;
    const uint32_t t0 = img_tile_ptr[image] * n_groups, t1 = img_tile_ptr[image + 1] * n_groups;
    const int comp = threadIdx.x & 31, slice = threadIdx.x >> 5;
    if (comp < LINEAR_SUMS) {
        // a slice's partials in its order, eight loads at a time (a load per add made this a chain of ~80 memory round trips:
        // 25 us for a kernel of a hundred blocks); a slot past the end adds +0.0: the same sums
        double v = 0;
        for (uint32_t t = t0 + slice; t < t1; t += 64) {
            double w[8];
            #pragma unroll
            for (int u = 0; u < 8; u++) w[u] = (t + 8u * u < t1) ? partial[(size_t)(t + 8u * u) * LINEAR_SUMS + comp] : 0.0;
            #pragma unroll
            for (int u = 0; u < 8; u++) v += w[u];
        }
        part[slice][comp] = v;
    }
    __syncthreads();
    if (threadIdx.x < LINEAR_SUMS) {
        double v = 0;
        for (int sl = 0; sl < 8; sl++) v += part[sl][threadIdx.x];
        sums[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int k = threadIdx.x;
        double *M = mat + (size_t)image * 16;
        const double sDisp = sums[k], sPosA = sums[3 + k], sPosB = sums[6 + k];
        const double sPosA2 = sums[9 + k], sPosB2 = sums[12 + k], sWeight = sums[15];
        const float scale = (float)M[5 * k];
        float newScale = 1.0f;
        if (use_scale)
            newScale = (float)pow((sWeight * sPosB2 - sPosB * sPosB) / (sWeight * sPosA2 - sPosA * sPosA),
                                  0.5 * (double)linear_alpha);
        if (!isnan(newScale)) {
            M[5 * k] = (double)(scale * newScale);
            const float translation = (float)M[4 * k + 3];
            if (!isnan(translation))
                M[4 * k + 3] = (double)translation + (double)linear_alpha * sDisp / sWeight
                             + sPosA * (double)(1 - newScale) / sWeight;
        }
    }
    if (!img_energy) return;
    if (threadIdx.x == 0) {
        __hip_atomic_store(&img_energy[2 * blockIdx.x], sums[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&img_energy[2 * blockIdx.x + 1], sums[17], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last_s = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last_s) return;
    __threadfence();
    // the last block: every image's pair fetched side by side, then added in image order by one thread
    __shared__ double pairs_s[2][256];
    double s0 = 0, s1 = 0;
    for (uint32_t base = 0; base < gridDim.x; base += 256) {
        const uint32_t i = base + threadIdx.x;
        if (i < gridDim.x) {
            pairs_s[0][threadIdx.x] = __hip_atomic_load(&img_energy[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pairs_s[1][threadIdx.x] = __hip_atomic_load(&img_energy[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (uint32_t k = 0; k < min(256u, gridDim.x - base); k++) { s0 += pairs_s[0][k]; s1 += pairs_s[1][k]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        energy[0] = s0; energy[1] = s1; energy[2] = 0.0;
        energy[3] = list_invalid ? (double)list_invalid[0] : 0.0;
        *ticket = 0u;                                       // ready for the next launch (same stream: ordered)
    }
}

// per-image census from the tile counters (imageGroup.cxx:1033-1046)
__global__ void count_reduce_kernel(const long long *tile_counts, const uint32_t *img_tile_ptr, uint32_t n_groups,
                                    uint32_t image_begin, uint32_t n_owned, long long *out /*[n_owned][2]*/)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_owned) return;
    const uint32_t image = image_begin + i;
    long long a = 0, b = 0;
    for (uint32_t t = img_tile_ptr[image] * n_groups; t < img_tile_ptr[image + 1] * n_groups; t++) { a += tile_counts[(size_t)t * 2]; b += tile_counts[(size_t)t * 2 + 1]; }
    out[(size_t)i * 2] = a; out[(size_t)i * 2 + 1] = b;
}

} // namespace frog
