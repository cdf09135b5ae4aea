// k_stats.hip.h -- updateStats on the device (K1, K2 and the reservoir S1 of
// SURVEY.md): which half-links feed the EM fit, their distances, and the fit.
//
// Reservoir (stats.h:58-76, setupStats imageGroup.cxx:1151-1159): an image with
// more half-links than Stats::maxSize keeps a sample when
//     (float) rng() / rng.max()  <=  (float) capacity / virtualSize
// and stops drawing once the buffer is full; the mt19937 is seeded with 0 once
// and never reseeded, so the state carries over from refresh to refresh.  Which
// ordinals are kept therefore depends only on (virtualSize, capacity, number of
// earlier refreshes), not on the data: select_kernel replays the generator, one
// wavefront per image, and emits the kept ordinals bit-exactly.
#pragma once

#include "ctx.h"

namespace frog {

constexpr int MT_N = 624;
constexpr int MT_M = 397;
constexpr int MT_WORDS = MT_N + 1;      // state + index of the next unread word

__device__ __forceinline__ uint32_t mt_twist(uint32_t cur, uint32_t nxt, uint32_t far)
{
    uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// Regenerates all 624 state words in place (std::mt19937's _M_gen_rand), one block of SELECT_THREADS threads, state in
// LDS.  The recurrence has lag 227 (= N - M): new[k] needs old[k], old[k+1] and, for k >= 227, new[k-227].  So thread
// t < 227 forms the chain new[t] -> new[t+227] -> new[t+454] in its own registers from OLD words only (the last word,
// new[623] = f(old[623], new[0], new[396]), falls to thread 169, which holds new[396] and recomputes new[0]): ONE
// barrier between the reads and the writes instead of the four dependency phases [0,227) [227,454) [454,623) {623} with
// two barriers each that the first version took -- the replay is a latency chain of 1 600 regenerations per image and
// refresh, and barriers were most of it.
constexpr int SELECT_THREADS = 256;

__device__ __forceinline__ void mt_regenerate(uint32_t *x, int tid)
{
    constexpr int LAG = MT_N - MT_M;            // 227
    uint32_t a = 0, b = 0, c = 0;
    if (tid < LAG) {
        a = mt_twist(x[tid], x[tid + 1], x[tid + MT_M]);
        b = mt_twist(x[tid + LAG], x[tid + LAG + 1], a);
        if (tid + 2 * LAG < MT_N - 1) c = mt_twist(x[tid + 2 * LAG], x[tid + 2 * LAG + 1], b);
        else if (tid + 2 * LAG == MT_N - 1) c = mt_twist(x[MT_N - 1], mt_twist(x[0], x[1], x[MT_M]), b);
    }
    __syncthreads();
    if (tid < LAG) {
        x[tid] = a;
        x[tid + LAG] = b;
        if (tid + 2 * LAG < MT_N) x[tid + 2 * LAG] = c;
    }
    __syncthreads();
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// One block of 256 threads per owned image.
//   virtual_size[i] <= cap : every ordinal is kept, no draw (needsRandom false).
//   else                   : replay draws until the buffer is full or the
//                            image's half-links are exhausted.
// The replay of one image is a chain (every block of 624 outputs needs the previous one) and the
// kernel is pure latency: with a single wavefront per image a refresh of 10^6 draws took 8 ms,
// more than ten iterations last once the images are spread over several GPUs.  A round consumes ALL
// the words left in the state (up to 624: three per thread, word index = ordinal order), tests them,
// ranks the kept ones (ballots + a 12-entry prefix over (row, wavefront)) and stores them: two barriers
// per round and two per regeneration (2.4 ms per refresh with rounds of 256 draws and the four-phase
// regeneration; the kept ordinals are the same, bit for bit: the parity tests compare them).
constexpr int SELECT_ROWS = (MT_N + SELECT_THREADS - 1) / SELECT_THREADS;     // 3

__global__ __launch_bounds__(SELECT_THREADS) void select_kernel(uint32_t *mt_state, const uint32_t *virtual_size,
                                                                uint32_t cap, uint32_t *sample_ord, uint32_t *sample_count)
{
    __shared__ uint32_t x[MT_N];
    __shared__ uint32_t row_cnt[SELECT_ROWS][SELECT_THREADS / 64];
    __shared__ int jstar_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t img = blockIdx.x;
    const uint32_t vs = virtual_size[img];
    uint32_t *ord = sample_ord + (size_t)img * cap;

    if (vs <= cap) {
        for (uint32_t k = tid; k < vs; k += SELECT_THREADS) ord[k] = k;
        if (tid == 0) sample_count[img] = vs;
        return;
    }

    uint32_t *st = mt_state + (size_t)img * MT_WORDS;
    for (int k = tid; k < MT_N; k += SELECT_THREADS) x[k] = st[k];
    uint32_t idx = st[MT_N];                       // block-uniform, like everything that steers the loop
    if (tid == 0) jstar_s = -1;
    __syncthreads();
    if (idx >= (uint32_t)MT_N) { mt_regenerate(x, tid); idx = 0; }      // (a state saved exactly at its end)

    constexpr int LAG = MT_N - MT_M;
    const float thresh = (float)cap / (float)vs;   // (float) samples.size() / virtualSize
    uint32_t count = 0;                            // kept so far
    uint32_t ordinal = 0;                          // draws consumed this refresh
    bool done = false;
    // One round = the words [idx, 624) of the current state: tested and ranked -- and, in the same two barrier intervals, the
    // NEXT state formed in registers from the current one (mt_regenerate's chain), to be written only if the replay goes on
    // past this state.  Two barriers per 624 draws (four with the regeneration as a step of its own).
    while (!done && ordinal < vs) {
        const uint32_t n = min((uint32_t)MT_N - idx, vs - ordinal);
        uint32_t na = 0, nb = 0, nc = 0;           // the next state's words tid, tid + 227, tid + 454 (mt_regenerate)
        if (tid < LAG) {
            na = mt_twist(x[tid], x[tid + 1], x[tid + MT_M]);
            nb = mt_twist(x[tid + LAG], x[tid + LAG + 1], na);
            if (tid + 2 * LAG < MT_N - 1) nc = mt_twist(x[tid + 2 * LAG], x[tid + 2 * LAG + 1], nb);
            else if (tid + 2 * LAG == MT_N - 1) nc = mt_twist(x[MT_N - 1], mt_twist(x[0], x[1], x[MT_M]), nb);
        }
        // per row: kept? (bit r of keep_bits) and how many lower lanes of this wavefront kept theirs (byte r of below_pk)
        uint32_t keep_bits = 0, below_pk = 0;
        #pragma unroll
        for (int r = 0; r < SELECT_ROWS; r++) {
            const uint32_t q = (uint32_t)tid + (uint32_t)(SELECT_THREADS * r);
            bool keep = false;
            if (q < n) {
                const uint32_t y = mt_temper(x[idx + q]);
                const float u = (float)y / 4294967296.0f;   // (float) rng() / rng.max(); (float)0xFFFFFFFF == 2^32
                keep = !(u > thresh);
            }
            const unsigned long long mask = __ballot(keep);
            keep_bits |= (keep ? 1u : 0u) << r;
            below_pk |= (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)) << (8 * r);
            if (lane == 0) row_cnt[r][wave] = (uint32_t)__popcll(mask);
        }
        __syncthreads();                           // every read of x and every row_cnt of this round is done
        uint32_t kept = 0;
        #pragma unroll
        for (int r = 0; r < SELECT_ROWS; r++)
            #pragma unroll
            for (int w = 0; w < SELECT_THREADS / 64; w++) kept += row_cnt[r][w];
        // the buffer fills inside this round: keep the first (cap - count); the need-th kept draw is the last one consumed
        const bool fills = count + kept >= cap;
        const uint32_t need = fills ? cap - count : 0xFFFFFFFFu;
        uint32_t before = 0;                       // kept draws of the rows and wavefronts before this thread's (row, wavefront)
        #pragma unroll
        for (int r = 0; r < SELECT_ROWS; r++) {
            uint32_t rank = before + ((below_pk >> (8 * r)) & 0xFFu);
            #pragma unroll
            for (int w = 0; w < SELECT_THREADS / 64; w++) {
                const uint32_t c = row_cnt[r][w];
                if (w < wave) rank += c;
                before += c;
            }
            const uint32_t q = (uint32_t)tid + (uint32_t)(SELECT_THREADS * r);
            if (((keep_bits >> r) & 1u) && rank < need) ord[count + rank] = ordinal + q;
            if (((keep_bits >> r) & 1u) && fills && rank == need - 1) jstar_s = (int)q;
        }
        // the replay goes on past this state (block-uniform: `fills`, n, idx, ordinal and vs are): the next state replaces it
        const bool advance = !fills && idx + n == (uint32_t)MT_N && ordinal + n < vs;
        if (advance && tid < LAG) {
            x[tid] = na;
            x[tid + LAG] = nb;
            if (tid + 2 * LAG < MT_N) x[tid + 2 * LAG] = nc;
        }
        __syncthreads();                           // jstar_s, the new state; row_cnt may be written again
        if (fills) {
            const uint32_t jstar = (uint32_t)jstar_s;
            idx += jstar + 1;
            ordinal += jstar + 1;
            count = cap;
            done = true;
        } else {
            count += kept;
            ordinal += n;
            idx = advance ? 0u : idx + n;
        }
    }
    for (int k = tid; k < MT_N; k += SELECT_THREADS) st[k] = x[k];
    if (tid == 0) { st[MT_N] = idx; sample_count[img] = count; }
}

// K1: distance of every kept half-link (imageGroup.cxx:579-590), thread per slot, in two parts.  Which half-links a
// refresh keeps does not depend on the data, so neither do their end points: sample_resolve_kernel turns every kept
// ordinal into (own point, partner point) -- a binary search of 15 dependent loads in the reference-order row pointers --
// on the side stream, right behind the selection; what is left on the critical path is two gathers and a distance
// (62 -> 15 us per refresh).
__global__ __launch_bounds__(256) void sample_resolve_kernel(
    const uint32_t *sample_ord, const uint32_t *sample_count, uint32_t cap,
    const uint32_t *poff, uint32_t image_begin, uint32_t own_pt_begin,
    const uint64_t *ref_rowptr, const uint32_t *ref_link, const uint32_t *new_of_old, uint2 *sample_ends)
{
    const uint32_t img = blockIdx.y;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= sample_count[img]) return;
    const uint32_t image = image_begin + img;
    const uint32_t pb = poff[image] - own_pt_begin, pe = poff[image + 1] - own_pt_begin;
    const uint64_t l = ref_rowptr[pb] + sample_ord[(size_t)img * cap + slot];
    // largest p in [pb, pe) with rowptr[p] <= l
    uint32_t lo = pb, hi = pe;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (ref_rowptr[mid] <= l) lo = mid; else hi = mid;
    }
    // rows are in reference order, coordinates in internal order
    sample_ends[(size_t)img * cap + slot] = make_uint2(new_of_old[lo], ref_link[l]);
}

__global__ __launch_bounds__(256) void sample_distance_kernel(const uint2 *sample_ends, const uint32_t *sample_count, uint32_t cap,
                                                              const P3 *pos2, float *samples)
{
    const uint32_t img = blockIdx.y;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= sample_count[img]) return;
    const uint2 ends = sample_ends[(size_t)img * cap + slot];
    const P3 a = pos2[ends.x];
    const P3 b = pos2[ends.y];
    // vtkMath::Distance2BetweenPoints(pA, pB), f32
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    samples[(size_t)img * cap + slot] = sqrtf(dx * dx + dy * dy + dz * dz);
}

// chipdf (stats.h:10-16) with the reference's promotions: f32 c*x2, f64 exp.
__device__ __forceinline__ float chi_pdf_ref(float x)
{
    const float c = 0.797884560802865f;
    const float x2 = x * x;
    const float cx2 = c * x2;
    return (float)((double)cx2 * exp(-0.5 * (double)x2));
}

// K2: Stats::estimateDistribution (stats.cxx:14-70), one block of four wavefronts per owned image.
// The reference adds the per-sample terms SEQUENTIALLY into f32 accumulators
// (sum3/sum4 through an f64 add that is rounded back to f32 each step); the fit's
// stop test makes the result sensitive to that order at the 1e-5 level, which is
// amplified in the fine lattices.  So the order is kept, and the kernel is a chain of
// 10^4 dependent additions per sum and EM iteration: pure latency, on as many wavefronts
// as there are images.  What can run side by side does: wavefront 3 computes the
// membership t of the NEXT 64 samples (two f64 exps each, the expensive part) into LDS
// while wavefronts 0-2 each run their own chain(s) over the current 64 -- wavefront 0
// sum1 and sum2 (f32), wavefront 1 sum3, wavefront 2 sum4 (f64 add, rounded to f32 each
// step) -- reading term k of the batch with v_readlane.  Same terms, same order, same
// bits as one wavefront doing everything (which took 1.9x as long).
__device__ __forceinline__ float lane_f32(float v, int k)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k));
}
__device__ __forceinline__ double lane_f64(double v, int k)
{
    const long long bits = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), k);
    const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), k);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

__global__ __launch_bounds__(256) void em_kernel(const float *samples, const uint32_t *sample_count, uint32_t cap,
                                                 uint32_t image_begin, float4 *em, int max_iterations, float epsilon)
{
    __shared__ float t_s[2][64];
    __shared__ float sums_s[4];
    const uint32_t img = blockIdx.x;
    const uint32_t n = sample_count[img];
    const float *smp = samples + (size_t)img * cap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 e0 = em[image_begin + img];
    float c1 = e0.x, c2 = e0.y, ratio = e0.z;
    const float esp = 1.59576912160573f;
    const uint32_t n_batches = (n + 63) / 64;
    int iteration = 0;
    while (iteration++ < max_iterations) {
        // membership of sample i under the current parameters (stats.cxx:30-32)
        auto membership = [&](uint32_t i) {
            float t = 0.f;
            if (i < n) {
                const float x = smp[i];
                const float f1 = ratio * chi_pdf_ref(x / c1) / c1;
                const float f2 = (float)((1.0 - (double)ratio) * (double)chi_pdf_ref(x / c2) / (double)c2);
                t = (float)((double)f1 / ((double)(f1 + f2) + 1e-16));
            }
            return t;
        };
        float sum1 = 0, sum2 = 0, sum3 = 0, sum4 = 0;
        if (wave == 3 && n_batches) t_s[0][lane] = membership(lane);
        __syncthreads();
        for (uint32_t b = 0; b < n_batches; b++) {
            const uint32_t base = b * 64;
            if (wave == 3) {
                if (b + 1 < n_batches) t_s[(b + 1) & 1][lane] = membership(base + 64 + lane);
            } else {
                const uint32_t i = base + lane;
                const float t = t_s[b & 1][lane];
                const float p = (i < n ? smp[i] : 0.f) * 1.0f;      // weights are all 1 (addSample's default)
                const int cnt = (int)min(64u, n - base);
                if (wave == 0) {                                     // sum1 += t*p; sum2 += t*w   (:33-35)
                    const float a = t * p;
                    if (cnt == 64) {
                        #pragma unroll
                        for (int k = 0; k < 64; k++) { sum1 += lane_f32(a, k); sum2 += lane_f32(t, k); }
                    } else {
                        for (int k = 0; k < cnt; k++) { sum1 += lane_f32(a, k); sum2 += lane_f32(t, k); }
                    }
                } else if (wave == 1) {                              // sum3 += (1.0 - t) * p         (:36)
                    const double v = (1.0 - (double)t) * (double)p;
                    if (cnt == 64) {
                        #pragma unroll
                        for (int k = 0; k < 64; k++) sum3 = (float)((double)sum3 + lane_f64(v, k));
                    } else {
                        for (int k = 0; k < cnt; k++) sum3 = (float)((double)sum3 + lane_f64(v, k));
                    }
                } else {                                             // sum4 += (1.0 - t) * w         (:37)
                    const double v = (1.0 - (double)t) * 1.0;
                    if (cnt == 64) {
                        #pragma unroll
                        for (int k = 0; k < 64; k++) sum4 = (float)((double)sum4 + lane_f64(v, k));
                    } else {
                        for (int k = 0; k < cnt; k++) sum4 = (float)((double)sum4 + lane_f64(v, k));
                    }
                }
            }
            __syncthreads();
        }
        if (lane == 0) {
            if (wave == 0) { sums_s[0] = sum1; sums_s[1] = sum2; }
            if (wave == 1) sums_s[2] = sum3;
            if (wave == 2) sums_s[3] = sum4;
        }
        __syncthreads();
        sum1 = sums_s[0]; sum2 = sums_s[1]; sum3 = sums_s[2]; sum4 = sums_s[3];
        __syncthreads();                                // sums_s is rewritten by the next iteration
        float sum5 = (float)n;                          // n additions of 1.0f, exact below 2^24
        sum2 = fmaxf(sum2, epsilon);
        sum3 = fmaxf(sum3, epsilon);
        sum5 = fmaxf(sum5, epsilon);
        const float nc1 = fmaxf(epsilon, sum1 / sum2 / esp);
        const float nc2 = fmaxf(epsilon, sum3 / sum4 / esp);
        const float nr = fmaxf(epsilon, sum2 / sum5);
        const bool done = (double)fabsf((c1 - nc1) / nc1) < 0.001
                       && (double)fabsf((c2 - nc2) / nc2) < 0.001
                       && (double)fabsf((nr - ratio) / nr) < 0.001;
        c1 = nc1; c2 = nc2; ratio = nr;
        if (done) break;
    }
    if (threadIdx.x == 0) em[image_begin + img] = make_float4(c1, c2, ratio, 0.f);
}

// ---- K2, second form: the same sums, bit for bit, without walking them one term at a time -------------------
// Every accumulator of estimateDistribution is  s <- fl32(s + v)  with v >= 0 (sum1, sum2: f32 adds; sum3, sum4: an f64
// add rounded back to f32, stats.cxx:33-37).  While s stays inside one binade [2^e, 2^(e+1)) it is a multiple S u of
// u = 2^(e-23), and fl32(S u + v) = (S + r) u with r = v / u rounded to the nearest integer: the 64 terms of a batch
// become 64 integers, and the 64 dependent additions an integer prefix sum across the wavefront (6 DPP steps).  What
// that argument does not cover is handled by taking the one offending term through the real arithmetic and going on
// behind it:
//   * the term at which S + r reaches 2^24 (the sum leaves its binade: coarser grid from there on);
//   * a term whose fraction v/u - floor(v/u) is within 2^-20 of 1/2: a tie (round to even depends on S), or close
//     enough to one that the intermediate f64 rounding of sum3 / sum4 (at most 2^-28 u) could decide -- everywhere
//     else the two roundings of fl32(fl64(s + v)) and the single one of the integer form agree;
//   * s = 0 or denormal (the first non-zero term of every sum), non-finite values.
// About one batch in five stops once; the chain of a batch costs ~60 instructions instead of 64 dependent additions
// of 30+ cycles each.  The memberships t, which do not depend on each other, are computed by all sixteen wavefronts
// of the block for a chunk of samples at a time (LDS), then three wavefronts run the four chains over the chunk.
// Checked against em_kernel (the term-by-term form, kept: FROG_EM_SERIAL=1, frog_test_em_refit) bit for bit
// (tests/test_gpu_fullsize.py).
constexpr int EM_THREADS = 1024;
constexpr int EM_CHUNK = 8192;

__device__ __forceinline__ uint32_t wave_inclusive_scan_u32(uint32_t v, int lane)
{
    // row_shr:1,2,4,8 inside rows of 16 lanes, then row_bcast:15 / row_bcast:31 across rows (GCN / CDNA DPP)
    uint32_t t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false); if ((lane & 15) >= 1) v += t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false); if ((lane & 15) >= 2) v += t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false); if ((lane & 15) >= 4) v += t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false); if ((lane & 15) >= 8) v += t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xf, 0xf, false); if ((lane & 31) >= 16) v += t;
    t = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xf, 0xf, false); if (lane >= 32) v += t;
    return v;
}

// v / ulp(s) rounded to the nearest integer for a sum in the binade of exponent e (see above); `odd`: the term has to go
// through the real arithmetic (beyond the binade, a tie or close enough to one, NaN).  One definition for the chains and
// for the batch totals below: the same bits by construction.
__device__ __forceinline__ uint32_t em_term_f64(const double v, const int e, bool &odd)
{
    const double x = __builtin_ldexp(v, 23 - e);                // v / ulp(s), exact
    if (!(x < 16777216.0)) { odd = true; return 0u; }           // far beyond the binade (or NaN)
    const uint32_t f = (uint32_t)x;                             // x >= 0: truncation is floor
    const double frac = x - (double)f;
    odd = __builtin_fabs(frac - 0.5) < 9.5367431640625e-07;     // 2^-20
    return f + (frac > 0.5 ? 1u : 0u);
}
__device__ __forceinline__ uint32_t em_term_f32(const float v, const int e, bool &odd)
{
    const float x = __builtin_ldexpf(v, 23 - e);                // exact unless it overflows (-> inf: caught below)
    if (!(x < 16777216.0f)) { odd = true; return 0u; }
    const uint32_t f = (uint32_t)x;
    const float frac = x - (float)f;                            // exact: the low bits of x
    odd = frac == 0.5f;                                         // f32 terms: a single rounding, only the exact tie is special
    return f + (frac > 0.5f ? 1u : 0u);
}

// s <- fl32(fl64(s + v_0)), then v_1, ... v_{cnt-1} (v of lane k = term k), all lanes return the result
__device__ __forceinline__ float em_chain(float s, const double v, const int cnt, const int lane)
{
    int j = 0;
    while (j < cnt) {                                   // wave-uniform
        const uint32_t sb = __float_as_uint(s);
        const uint32_t ex = sb >> 23;                   // s >= 0: no sign bit
        const bool active = lane >= j && lane < cnt;
        int stop;
        if (ex == 0u || ex >= 255u) {
            // zero / denormal / non-finite sum: skip the terms that leave a zero sum alone, take the next one directly
            if (sb == 0u) {
                const unsigned long long nz = __ballot(active && v != 0.0);
                stop = nz ? (int)__ffsll((long long)nz) - 1 : cnt;
            } else {
                stop = j;
            }
        } else {
            const int e = (int)ex - 127;
            uint32_t r = 0;
            bool odd = false;
            if (active) r = em_term_f64(v, e, odd);
            const uint32_t P = wave_inclusive_scan_u32(r, lane);    // < 64 * 2^24
            const uint32_t S = (sb & 0x7FFFFFu) | 0x800000u;
            const unsigned long long halt = __ballot(active && (odd || S + P >= 0x1000000u));
            stop = halt ? (int)__ffsll((long long)halt) - 1 : cnt;
            if (stop > j) {
                const uint32_t Pp = (uint32_t)__builtin_amdgcn_readlane((int)P, stop - 1);
                s = __builtin_ldexpf((float)(S + Pp), e - 23);      // S + Pp < 2^24: exact
            }
        }
        if (stop < cnt) {
            const long long bits = __double_as_longlong(v);
            const int lo = __builtin_amdgcn_readlane((int)(bits & 0xffffffffll), stop);
            const int hi = __builtin_amdgcn_readlane((int)(bits >> 32), stop);
            const double vs = __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
            s = (float)((double)s + vs);                            // the reference's own step
        }
        j = stop + 1;
    }
    return s;
}

// the same for f32 terms (sum1, sum2): v / ulp(s), its floor and its fraction are exact in f32 too
__device__ __forceinline__ float em_chain_f32(float s, const float v, const int cnt, const int lane)
{
    int j = 0;
    while (j < cnt) {
        const uint32_t sb = __float_as_uint(s);
        const uint32_t ex = sb >> 23;
        const bool active = lane >= j && lane < cnt;
        int stop;
        if (ex == 0u || ex >= 255u) {
            if (sb == 0u) {
                const unsigned long long nz = __ballot(active && v != 0.0f);
                stop = nz ? (int)__ffsll((long long)nz) - 1 : cnt;
            } else {
                stop = j;
            }
        } else {
            const int e = (int)ex - 127;
            uint32_t r = 0;
            bool odd = false;
            if (active) r = em_term_f32(v, e, odd);
            const uint32_t P = wave_inclusive_scan_u32(r, lane);
            const uint32_t S = (sb & 0x7FFFFFu) | 0x800000u;
            const unsigned long long halt = __ballot(active && (odd || S + P >= 0x1000000u));
            stop = halt ? (int)__ffsll((long long)halt) - 1 : cnt;
            if (stop > j) {
                const uint32_t Pp = (uint32_t)__builtin_amdgcn_readlane((int)P, stop - 1);
                s = __builtin_ldexpf((float)(S + Pp), e - 23);
            }
        }
        if (stop < cnt) s += lane_f32(v, stop);                     // the reference's own step
        j = stop + 1;
    }
    return s;
}

// Batch totals.  A batch of 64 terms that neither leaves the binade nor contains a (near) tie changes the sum by the integer
// T = sum of its r: s' = (S + T) u, whatever the 64 prefix sums in between were -- and r depends on the sum only through
// its exponent.  The exponent the running sum had before batch b in the PREVIOUS EM iteration (same samples, parameters a
// little different) is almost always the one it has now, so the sixteen wavefronts that compute the memberships also
// compute, for that guess, every batch's T and whether it holds a term that needs the real arithmetic; the wavefront that
// runs a chain then takes a batch in a dozen SCALAR instructions (exponent as guessed, no flag, S + T < 2^24) or falls back
// to em_chain.  86 % of the kernel was those chains (measured by running them twice).  A steady-state fit converges in two
// iterations, so the guesses also persist from fit to fit (other samples, the same magnitudes): 91 % of all batches of
// the default schedule take the short way, 0.205 -> 0.155 ms per fit.  The guesses decide the way, never the result.
constexpr int EM_BATCHES = EM_CHUNK / 64;
constexpr int EM_GUESS_BATCHES = 1024;          // batches (of the whole sample set) that keep a guess: 65 536 samples

__global__ __launch_bounds__(EM_THREADS) void em_scan_kernel(const float *samples, const uint32_t *sample_count, uint32_t cap,
                                                             uint32_t image_begin, float4 *em, int max_iterations, float epsilon,
                                                             unsigned char *guess_g /* [images][4][EM_GUESS_BATCHES] or null */)
{
    __shared__ float t_s[EM_CHUNK];
    __shared__ float p_s[EM_CHUNK];                         // the chunk's samples: the chains' long way reads them here, not from memory
    __shared__ float sums_s[4];
    __shared__ uint32_t total_s[4][EM_BATCHES];             // T of the chunk's batches under the guessed exponent
    __shared__ unsigned char exact_s[4][EM_BATCHES];        // 1: no guess, or a term of the batch needs the real arithmetic
    __shared__ unsigned char guess_s[4][EM_GUESS_BATCHES];  // biased exponent of the running sum before the batch, last iteration (0: none)
    const uint32_t img = blockIdx.x;
    const uint32_t n = sample_count[img];
    const float *smp = samples + (size_t)img * cap;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4 e0 = em[image_begin + img];
    float c1 = e0.x, c2 = e0.y, ratio = e0.z;
    const float esp = 1.59576912160573f;
    // the guesses of a fit's first iteration are the previous fit's (a steady-state fit converges in two iterations: without
    // them half of all batches would go the long way); other samples, the same magnitudes
    unsigned char *guess_mine = guess_g ? guess_g + (size_t)img * 4 * EM_GUESS_BATCHES : nullptr;
    for (int k = threadIdx.x; k < 4 * EM_GUESS_BATCHES; k += EM_THREADS) (&guess_s[0][0])[k] = guess_mine ? guess_mine[k] : 0;
    __syncthreads();
    int iteration = 0;
    while (iteration++ < max_iterations) {
        float sum1 = 0, sum2 = 0, sum3 = 0, sum4 = 0;
        for (uint32_t c0 = 0; c0 < n; c0 += EM_CHUNK) {
            const uint32_t m = min((uint32_t)EM_CHUNK, n - c0);
            const uint32_t gb0 = c0 / 64u;                   // first batch of the chunk in the numbering of the whole set
            // membership of the chunk's samples under the current parameters (stats.cxx:30-32), all wavefronts; a wavefront's
            // 64 samples of one trip are one batch of the chains
            float x_next = (wave * 64u + lane) < m ? smp[c0 + wave * 64u + lane] : 0.f;        // one trip ahead of its use
            for (uint32_t i0 = wave * 64u; i0 < m; i0 += EM_THREADS) {
                const uint32_t i = i0 + lane;
                const bool in = i < m;
                const float x = x_next;
                x_next = (i + EM_THREADS) < m ? smp[c0 + i + EM_THREADS] : 0.f;
                float t = 0.f, p = 0.f;
                if (in) {
                    const float f1 = ratio * chi_pdf_ref(x / c1) / c1;
                    const float f2 = (float)((1.0 - (double)ratio) * (double)chi_pdf_ref(x / c2) / (double)c2);
                    t = (float)((double)f1 / ((double)(f1 + f2) + 1e-16));
                    t_s[i] = t;
                    p = x * 1.0f;                            // weights are all 1 (addSample's default)
                    p_s[i] = p;
                }
                const uint32_t lb = i0 / 64u, gb = gb0 + lb;
                #pragma unroll
                for (int a = 0; a < 4; a++) {
                    const uint32_t g = gb < (uint32_t)EM_GUESS_BATCHES ? guess_s[a][gb] : 0u;     // wave-uniform
                    uint32_t r = 0;
                    bool odd = false;
                    if (g - 1u < 254u && in) {
                        const int e = (int)g - 127;
                        // the four terms exactly as the chains form them below
                        if (a == 0) r = em_term_f32(t * p, e, odd);
                        else if (a == 1) r = em_term_f32(t, e, odd);
                        else if (a == 2) r = em_term_f64((1.0 - (double)t) * (double)p, e, odd);
                        else r = em_term_f64((1.0 - (double)t) * 1.0, e, odd);
                    }
                    const bool any_odd = __ballot(odd) != 0ull;
                    #pragma unroll
                    for (int off = 32; off > 0; off >>= 1) r += (uint32_t)__shfl_down((int)r, off, 64);      // < 64 * 2^24
                    if (lane == 0) {
                        total_s[a][lb] = r;
                        exact_s[a][lb] = (g - 1u < 254u && !any_odd) ? 0 : 1;
                    }
                }
            }
            __syncthreads();
            if (wave < 4) {                                  // one wavefront per accumulator
                // this wavefront's accumulator, by value (a reference picked at run time sends all four sums to scratch memory)
                float acc = wave == 0 ? sum1 : wave == 1 ? sum2 : wave == 2 ? sum3 : sum4;
                // what the batches of this chunk need from LDS, fetched once: lane l holds batches l and l + 64 (guess and
                // exact flag in one word, the total in another); inside the loop they are v_readlane's and everything the
                // short way does is scalar -- three dependent LDS round trips per batch made it a quarter microsecond
                const uint32_t n_b = (m + 63u) / 64u;
                uint32_t meta[2], tot[2], seen[2];
                #pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t lb = (uint32_t)lane + 64u * h, gb = gb0 + lb;
                    const bool have = lb < n_b;
                    const uint32_t g = (have && gb < (uint32_t)EM_GUESS_BATCHES) ? guess_s[wave][gb] : 0u;
                    meta[h] = g | ((have ? (uint32_t)exact_s[wave][lb] : 1u) << 8);
                    tot[h] = have ? total_s[wave][lb] : 0u;
                    seen[h] = g;                             // becomes the exponent seen this time (unchanged where no batch)
                }
                for (uint32_t b = 0; b < m; b += 64) {
                    const uint32_t lb = b / 64u;
                    const uint32_t mt = (uint32_t)__builtin_amdgcn_readlane((int)(lb < 64u ? meta[0] : meta[1]), (int)(lb & 63u));
                    const uint32_t T = (uint32_t)__builtin_amdgcn_readlane((int)(lb < 64u ? tot[0] : tot[1]), (int)(lb & 63u));
                    const uint32_t sb = (uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(acc));     // the same in every lane
                    const uint32_t ex = sb >> 23;            // acc >= 0: no sign bit
                    {
                        const uint32_t keep = ex < 255u ? ex : 0u;                  // for the next iteration / fit
                        if ((uint32_t)lane == (lb & 63u)) { if (lb < 64u) seen[0] = keep; else seen[1] = keep; }
                    }
                    if (mt == ex && ex - 1u < 254u) {        // guessed right, no flag (bit 8), a normal number
                        const uint32_t S = (sb & 0x7FFFFFu) | 0x800000u;
                        if (S + T < 0x1000000u) {            // the whole batch stays inside the binade: s' = (S + T) u, exactly
                            acc = __uint_as_float((ex << 23) | ((S + T) & 0x7FFFFFu));
                            continue;
                        }
                    }
                    const uint32_t i = b + lane;
                    const int cnt = (int)min(64u, m - b);
                    const float t = i < m ? t_s[i] : 0.f;
                    const float p = i < m ? p_s[i] : 0.f;
                    if (wave == 0) acc = em_chain_f32(acc, t * p, cnt, lane);                                    // sum1 += t*p         (:33)
                    else if (wave == 1) acc = em_chain_f32(acc, t, cnt, lane);                                   // sum2 += t*w         (:35)
                    else if (wave == 2) acc = em_chain(acc, (1.0 - (double)t) * (double)p, cnt, lane);           // sum3 += (1.0-t)*p   (:36)
                    else acc = em_chain(acc, (1.0 - (double)t) * 1.0, cnt, lane);                                // sum4 += (1.0-t)*w   (:37)
                }
                if (wave == 0) sum1 = acc; else if (wave == 1) sum2 = acc; else if (wave == 2) sum3 = acc; else sum4 = acc;
                #pragma unroll
                for (int h = 0; h < 2; h++) {
                    const uint32_t lb = (uint32_t)lane + 64u * h, gb = gb0 + lb;
                    if (lb < n_b && gb < (uint32_t)EM_GUESS_BATCHES) guess_s[wave][gb] = (unsigned char)seen[h];
                }
            }
            __syncthreads();                            // t_s is rewritten by the next chunk / iteration
        }
        if (lane == 0) {
            if (wave == 0) sums_s[0] = sum1;
            if (wave == 1) sums_s[1] = sum2;
            if (wave == 2) sums_s[2] = sum3;
            if (wave == 3) sums_s[3] = sum4;
        }
        __syncthreads();
        sum1 = sums_s[0]; sum2 = sums_s[1]; sum3 = sums_s[2]; sum4 = sums_s[3];
        __syncthreads();                                // sums_s is rewritten by the next iteration
        float sum5 = (float)n;                          // n additions of 1.0f, exact below 2^24
        sum2 = fmaxf(sum2, epsilon);
        sum3 = fmaxf(sum3, epsilon);
        sum5 = fmaxf(sum5, epsilon);
        const float nc1 = fmaxf(epsilon, sum1 / sum2 / esp);
        const float nc2 = fmaxf(epsilon, sum3 / sum4 / esp);
        const float nr = fmaxf(epsilon, sum2 / sum5);
        const bool done = (double)fabsf((c1 - nc1) / nc1) < 0.001
                       && (double)fabsf((c2 - nc2) / nc2) < 0.001
                       && (double)fabsf((nr - ratio) / nr) < 0.001;
        c1 = nc1; c2 = nc2; ratio = nr;
        if (done) break;
    }
    if (threadIdx.x == 0) em[image_begin + img] = make_float4(c1, c2, ratio, 0.f);
    if (guess_mine) {
        __syncthreads();
        for (int k = threadIdx.x; k < 4 * EM_GUESS_BATCHES; k += EM_THREADS) guess_mine[k] = (&guess_s[0][0])[k];
    }
}

// (c1,c2,ratio) -> constants of inlier_probability for ALL images (after the
// EM table has been made whole by the all-reduce in multi-rank runs).
__device__ __forceinline__ EmDerived em_derived_of(const float4 e)
{
    const float eps = 1e-10f;
    const float c = 0.797884560802865f;
    const float K = -0.72134752044448170368f;           // -log2(e) / 2
    EmDerived d;
    const float inv1 = 1.0f / (e.x + eps), inv2 = 1.0f / (e.y + eps);
    const float q1 = inv1 * inv1, q2 = inv2 * inv2;
    d.kq1 = e.z * c * inv1 * q1;
    d.kq2 = (1.0f - e.z) * c * inv2 * q2;
    d.s1 = q1 * K;
    d.s2 = q2 * K;
    return d;
}

// ---- the one-exponential form of the deformable sweeps' weight (ctx.h EmFast, k_links.hip.h inlier_weight_pair) ----
// Leaving the reference's `+ 1e-10` (stats.h:91) out of the denominator raises p = x1 / (x1 + x2 + eps) by
//     D = eps x1 / (S (S + eps)) <= eps p^2 / x1,      S = x1 + x2, p = x1 / S,
// and the form is used where D <= EM_FAST_EPS_DROP is certain.  Two sufficient conditions, both per image:
//   (i)  x1 >= EM_FAST_DENSITY = eps / EM_FAST_EPS_DROP (p <= 1): an interval [lo1, hi1] of d2 around the inlier component's
//        maximum (em_density_range);
//   (ii) beyond hi1, for c1 < c2 (p and x1 both fall there): on a stretch [a, b], D <= eps p(a)^2 / x1(b); the range is
//        extended stretch by stretch (0.25 % of d2 each) while that bound holds, up to the distance from which p stays below
//        `theta` -- the sweep never asks for the range of a weight below theta = inlierThreshold - THRESHOLD_BAND (such a link is
//        an outlier in either form: k_links.hip.h).
constexpr double EM_FAST_EPS_DROP = 7e-6;
constexpr double EM_FAST_DENSITY = 1e-10 / EM_FAST_EPS_DROP;
constexpr float EM_FAST_D2_MIN = 0.0100001f;    // above D2_FIX (k_links.hip.h): `d < 0.1 -> 1` is left to the general form

// [lo, hi] (in d2) on which x(d2) = kq d2 2^(s d2) >= T, for s < 0: x rises to its one maximum at d2 = -1 / (s ln 2) and falls.
// Guess and verify: the ends are bisected with the hardware's f32 exponential, moved inwards by 1e-3, rounded to f32, and
// ACCEPTED ONLY IF x, in f64, is >= T at both -- x has one maximum, so then it is >= T between them, whatever the quality of
// the guess (a bad guess costs range, never correctness).
__device__ inline bool em_density_range(double kq, double s, double T, float &lo_out, float &hi_out)
{
    if (!(kq > 0.0) || !(kq < 1e30) || !(s < 0.0) || !(s > -1e30)) return false;
    const float kqf = (float)kq, sf = (float)s, Tf = (float)T;
    const float peak = -1.0f / (sf * 0.69314718f);
    auto xf = [&](float d2) { return kqf * d2 * __builtin_amdgcn_exp2f(sf * d2); };
    if (!(peak > 0.0f) || !(peak < 1e30f) || !(xf(peak) >= Tf)) return false;
    float a = 0.0f, b = peak;                           // rising side: x(a) < T <= x(b)
    for (int k = 0; k < 26; k++) { const float m = 0.5f * (a + b); if (xf(m) >= Tf) b = m; else a = m; }    // f32: 24 bits
    const float lo = b * 1.001f;
    a = peak; b = peak;                                 // falling side: x(a) >= T > x(b)
    for (int k = 0; k < 12 && xf(b) >= Tf; k++) { a = b; b *= 4.0f; }      // 2^(s d2) is 2^-1.44 at the maximum: gone after a few steps
    if (xf(b) >= Tf) return false;
    for (int k = 0; k < 26; k++) { const float m = 0.5f * (a + b); if (xf(m) >= Tf) a = m; else b = m; }
    const float hi = a * 0.999f;
    auto xd = [&](double d2) { return kq * d2 * exp2(s * d2); };
    if (!(lo <= hi) || !(xd((double)lo) >= T) || !(xd((double)hi) >= T)) return false;
    lo_out = lo; hi_out = hi;
    return true;
}

// c1' = fl(c1 + eps), c2' = fl(c2 + eps) as stats.h:88-90 forms them;  x2 / x1 = ((1 - r) / r) (c1'/c2')^3 exp(d2 (1/c1'^2 - 1/c2'^2) / 2)
__device__ inline EmFast em_fast_of(const float4 e, float theta)
{
    EmFast f;
    // no range: the general form decides every link this image's exponent does not already rule out -- and with l = -inf it
    // rules out none (the pair's exponent is then the other image's: an outlier by that image alone is an outlier)
    f.l = -__builtin_inff(); f.ds = 0.f; f.lo = __builtin_inff(); f.hi = -__builtin_inff();
    const float eps = 1e-10f;
    const double c1 = (double)(e.x + eps), c2 = (double)(e.y + eps), r = (double)e.z;
    if (!(c1 > 0.0) || !(c2 > 0.0) || !(c1 < 1e18) || !(c2 < 1e18) || !(r > 0.0) || !(r < 1.0)) return f;
    const double c0 = 0.797884560802865, K = -0.72134752044448170368;
    const double i1 = 1.0 / c1, i2 = 1.0 / c2;
    const double kq1 = r * c0 * i1 * i1 * i1, kq2 = (1.0 - r) * c0 * i2 * i2 * i2;
    const double s1 = K * i1 * i1, s2 = K * i2 * i2;
    const double l = log2(kq2 / kq1), ds = s2 - s1;
    if (!(fabs(l) <= 66.0) || !(fabs(ds) < 1e30)) return f;          // |l| <= 66: what the rounding analysis at inlier_weight_pair assumes
    // `d < 0.1 -> 1` (stats.h:87) is the general form's: a link that close must reach it, i.e. must not be dropped for a
    // one-exponential value below theta -- E = l + ds d2 <= log2(1 / theta - 1) with a margin, at both ends of [0, D2_MIN]
    // (E is linear in d2).  Holds for every mixture with an inlier component worth the name (l = -19 on the benchmark group).
    const double e_theta = log2(1.0 / (double)theta - 1.0) - 0.01;
    if (!(l <= e_theta) || !(l + ds * (double)EM_FAST_D2_MIN <= e_theta)) return f;
    float lo = 0.f, hi = 0.f;
    if (!em_density_range(kq1, s1, EM_FAST_DENSITY, lo, hi)) return f;
    // (ii): p(a)^2 / x1(b) with the hardware's f32 exponentials (1e-6 relative) against a bound taken 3 % short
    if (ds > 0.0 && theta > 1e-3f && theta < 0.999f) {
        const float lf = (float)l, dsf = (float)ds, kqf = (float)kq1, sf = (float)s1;
        // p < theta 0.999 from here on: E = l + ds d2 > log2(1 / (0.999 theta) - 1)
        const float d2_theta = (__builtin_log2f(1.0f / (0.999f * theta) - 1.0f) - lf) / dsf * 1.0001f;
        float a = hi;
        for (int k = 0; k < 512 && a < d2_theta; k++) {
            const float b = a * 1.0025f;
            const float pa = 1.0f / (1.0f + __builtin_amdgcn_exp2f(__builtin_fmaf(dsf, a, lf)));
            const float xb = kqf * b * __builtin_amdgcn_exp2f(sf * b);
            if (!(1e-10f * pa * pa <= 0.97f * (float)EM_FAST_EPS_DROP * xb)) break;
            a = b;
        }
        if (a > hi) hi = a * 0.9999f;                   // the last stretch accepted ends at a
    }
    f.l = (float)l; f.ds = (float)ds;
    f.lo = fmaxf(lo, EM_FAST_D2_MIN); f.hi = hi;
    return f;
}

__global__ void em_derive_kernel(const float4 *em, EmDerived *emd, EmFast *emf, uint32_t n_images, float theta)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_images) return;
    emd[i] = em_derived_of(em[i]);
    emf[i] = em_fast_of(em[i], theta);
}

} // namespace frog
