// k_refchain.hip.h -- the B-spline scatter of reference-order mode (bin/frog -exact 1) as one chain per control point.
//
// imageGroup.cxx:301-338 adds, image by image and point by point in index order, (float)((double) g + w * (double) s) into the
// four components of the 64 control points of the point's stencil.  The ORDER only binds additions into the SAME control point:
// per (image, control point) the additions form one chain over the contributing points in index order, and chains of
// different control points are independent.  ref_scatter_kernel (k_reforder.hip.h, rounds 4-5) replayed the loop literally --
// one wavefront per image, a barrier per point, 20 000 dependent read-modify-writes of global memory in a row: 45 ms per
// iteration at cfg 3.  Here, once per lattice (the cell of a point depends on the re-based coordinates `pos`, which only
// transformPoints(apply) changes, and every apply is followed by a new lattice):
//   1. one 64-bit key per (point, tap): (image, control point) | point ordinal inside the image | tap;
//   2. a radix sort of the keys (hipCUB): every control point's contributions, in point order;
//   3. the chains laid out for the chain kernel: control points sorted by chain length, RC_GROUP to a wavefront, entry j of the
//      group's chains side by side (coalesced), padded to the group's longest (the sort keeps the padding to the length differences
//      inside a group); per entry the point (internal numbering) and the tap's f64 weight wx[i] wy[j] wz[k], formed by the same
//      expressions ref_scatter_kernel evaluated per point and iteration;
// and per iteration ref_chain_kernel: thread = (image, control point, component), running its chain with the reference's arithmetic.  Same
// values added in the same order into every control point: the gradient lattice has ref_scatter_kernel's bits (and those of the
// tests' CPU restatement of the reference: tests/test_gpu_reference_order.py compares after every step), FROG_REF_LITERAL=1 keeps the literal form
// for comparison (tests/test_gpu_round6.py).
#pragma once

#include <hipcub/hipcub.hpp>

#include "ctx.h"
#include "k_grid.hip.h"
#include "k_reforder.hip.h"

namespace frog {

constexpr uint32_t RC_PAD = 0xFFFFFFFFu;        // no entry (padding of a group to its longest chain)
constexpr int RC_UNROLL_MAX = 16;               // chain entries fetched together: 16 on lattices with long chains, else 8; group lengths are padded to a multiple
constexpr int RC_GROUP = 16;                    // control points per wavefront of the chain kernel (x 4 components = 64 lanes)

// Keys of one point: wavefront = point (owned row r, reference order), lane = tap i + 4 j + 16 k.  Cell as imageGroup.cxx:303-310
// (f64 quotient rounded to f32, floor).  key = ((gnode << rbits | ordinal in the image) << 6) | tap; taps outside the lattice
// (undefined upstream, never reached by in-group points: SURVEY App. D.4) get gnode = n_gnodes and sort behind everything.
__global__ __launch_bounds__(256) void ref_chain_keys_kernel(const float4 *pos, const uint32_t *new_of_old, const uint32_t *poff,
                                                             uint32_t n_rows, uint32_t image_begin, uint32_t own_pt_begin, const GeomDev g,
                                                             uint32_t n_gnodes, int rbits, uint64_t *keys)
{
    const uint32_t r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n_rows) return;
    const int lane = threadIdx.x & 63;
    const int ti = lane & 3, tj = (lane >> 2) & 3, tk = lane >> 4;
    const float4 v = pos[new_of_old[r]];
    const uint32_t image = (uint32_t)__float_as_int(v.w);
    const uint32_t ordinal = r - (poff[image] - own_pt_begin);
    const float in[3] = { v.x, v.y, v.z };
    int i0[3];
    #pragma unroll
    for (int k = 0; k < 3; k++) {
        const float coord = (float)(((double)in[k] - g.origin[k]) / g.spacing[k]);
        i0[k] = (int)floorf(coord) - 1;
    }
    const int x = i0[0] + ti, y = i0[1] + tj, z = i0[2] + tk;
    uint64_t gnode = n_gnodes;
    if (x >= 0 && y >= 0 && z >= 0 && x < g.dims[0] && y < g.dims[1] && z < g.dims[2])
        gnode = (uint64_t)(image - image_begin) * (uint64_t)g.n_cp + ((uint64_t)x + (uint64_t)g.dims[0] * ((uint64_t)y + (uint64_t)g.dims[1] * (uint64_t)z));
    keys[(size_t)r * 64 + lane] = (((gnode << rbits) | ordinal) << 6) | (uint64_t)lane;
}

// node_ptr[g] = first sorted key of control point g (g = n_gnodes: first key outside the lattice = number of entries)
__global__ __launch_bounds__(256) void ref_chain_bounds_kernel(const uint64_t *keys, uint64_t n_keys, int rbits, uint32_t n_gnodes, uint32_t *node_ptr)
{
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e > n_keys) return;
    const long long prev = e == 0 ? -1 : (long long)(keys[e - 1] >> (rbits + 6));
    const long long cur = e == n_keys ? (long long)n_gnodes : (long long)(keys[e] >> (rbits + 6));
    for (long long gn = prev + 1; gn <= cur; gn++) node_ptr[gn] = (uint32_t)e;
}

// Sort key of a chain: its length rounded up to the steps the kernel takes (`unit` entries each).  Chains that take the same
// number of steps keep their order (the sort is stable): neighbouring control points of an image stay together in a wavefront,
// and neighbours share three quarters of their points -- their gathers of a step fall on the same lines.
__global__ __launch_bounds__(256) void ref_chain_len_kernel(const uint32_t *node_ptr, uint32_t n_gnodes, uint32_t unit, uint32_t *len, uint32_t *iota)
{
    const uint32_t gn = blockIdx.x * blockDim.x + threadIdx.x;
    if (gn >= n_gnodes) return;
    len[gn] = (node_ptr[gn + 1] - node_ptr[gn] + unit - 1) / unit * unit;
    iota[gn] = gn;
}

// group q = slots RC_GROUP q .. RC_GROUP q + RC_GROUP - 1 of the length-sorted control points: padded length (a multiple of RC_UNROLL) and size.
// (The slots that fill the last group up carry no control point, slot_node = RC_PAD, and sort among the empty chains.)
__global__ __launch_bounds__(256) void ref_chain_group_kernel(const uint32_t *len_sorted, uint32_t n_slots, uint32_t n_groups, uint32_t RC_UNROLL, uint32_t *group_len,
                                                              uint64_t *group_size, const uint32_t *slot_node, uint32_t *slot_of_node)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_groups) {
        const uint32_t longest = len_sorted[(size_t)t * RC_GROUP];            // sorted descending: the group's first
        const uint32_t padded = (longest + RC_UNROLL - 1) / RC_UNROLL * RC_UNROLL;
        group_len[t] = padded;
        group_size[t] = (uint64_t)padded * RC_GROUP;
    }
    if (t < n_slots && slot_node[t] != RC_PAD) slot_of_node[slot_node[t]] = t;
}

// Debug (FROG_REF_TRACE=1): chains longer than their group's padded length, slots without a control point, the sum of the slots' control points
__global__ __launch_bounds__(256) void ref_chain_check_kernel(const uint32_t *node_ptr, const uint32_t *slot_node, const uint32_t *group_len, uint32_t n_slots,
                                                              unsigned long long *out)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_slots) return;
    const uint32_t gn = slot_node[t];
    if (gn == RC_PAD) { atomicAdd(&out[1], 1ull); return; }
    atomicAdd(&out[2], (unsigned long long)gn);
    if (node_ptr[gn + 1] - node_ptr[gn] > group_len[t / RC_GROUP]) atomicAdd(&out[0], 1ull);
}

// Entry e of the sorted keys -> its seat in the chain layout: the point (internal numbering) and the tap's weight
// w = wx[i] * wy[j] * wz[k] in f64 from the weights of the f32 fraction (imageGroup.cxx:311-322), as ref_scatter_kernel forms them.
__global__ __launch_bounds__(256) void ref_chain_fill_kernel(const uint64_t *keys, uint32_t n_entries, int rbits, const uint32_t *node_ptr,
                                                             const uint32_t *slot_of_node, const uint64_t *group_ptr, const float4 *pos,
                                                             const uint32_t *new_of_old, const uint32_t *poff, uint32_t image_begin,
                                                             uint32_t own_pt_begin, const GeomDev g, int by_row, uint32_t *ent, double *wt)
{
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    const uint64_t key = keys[e];
    const int tap = (int)(key & 63u);
    const uint32_t ordinal = (uint32_t)((key >> 6) & ((1ull << rbits) - 1ull));
    const uint32_t gnode = (uint32_t)(key >> (rbits + 6));
    const uint32_t image = image_begin + gnode / (uint32_t)g.n_cp;
    const uint32_t p = new_of_old[poff[image] - own_pt_begin + ordinal];
    const float4 v = pos[p];
    const float in[3] = { v.x, v.y, v.z };
    double W[3][4];
    #pragma unroll
    for (int k = 0; k < 3; k++) {
        const float coord = (float)(((double)in[k] - g.origin[k]) / g.spacing[k]);
        const float fl = floorf(coord);
        bspline_weights(W[k], (double)(coord - fl));
    }
    const double w = W[0][tap & 3] * W[1][(tap >> 2) & 3] * W[2][tap >> 4];
    const uint32_t slot = slot_of_node[gnode];
    const size_t seat = (size_t)group_ptr[slot / RC_GROUP] + (size_t)(e - node_ptr[gnode]) * RC_GROUP + (slot % RC_GROUP);
    // the point as the chain kernel will look its sums up: internal numbering (Morton order: the points of a fine lattice's
    // chain lie together), or -- coarse lattices, where a chain takes every third point of its image -- the owned row in reference
    // order, in which a chain's entries ascend: its gathers then walk through the image's sums instead of jumping about in them
    ent[seat] = by_row ? poff[image] - own_pt_begin + ordinal : p;
    wt[seat] = w;
}

// The same fill through LDS: block = (group of RC_GROUP chains, 64 consecutive entries of each).  The sorted keys of a chain are
// consecutive (a wavefront reads 64 of them side by side), the seats of a group's entry j are consecutive across its chains (the
// block writes the tile entry-major): both sides in whole lines.  A thread per sorted key wrote its seat RC_GROUP entries from its
// neighbour's -- 128 M partial-line writes, 9.6 ms per level-0 lattice.  Also writes the padding (RC_PAD), so nothing is cleared first.
__global__ __launch_bounds__(256) void ref_chain_fill_tiled_kernel(const uint64_t *keys, int rbits, const uint32_t *node_ptr, const uint32_t *slot_node,
                                                                   const uint64_t *group_ptr, const uint32_t *group_len, const float4 *pos,
                                                                   const uint32_t *new_of_old, const uint32_t *poff, uint32_t image_begin,
                                                                   uint32_t own_pt_begin, const GeomDev g, int by_row, uint32_t n_groups,
                                                                   uint32_t *ent, double *wt)
{
    __shared__ uint32_t s_ent[RC_GROUP][65];
    __shared__ double s_wt[RC_GROUP][65];
    const uint32_t grp = blockIdx.z * gridDim.x + blockIdx.x;   // rc_grid(): a launch holds < 2^32 work-items per dimension
    if (grp >= n_groups) return;
    const uint32_t j0 = blockIdx.y * 64u, n = group_len[grp];
    if (j0 >= n) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    #pragma unroll
    for (int c = wave; c < RC_GROUP; c += 4) {
        const uint32_t gnode = slot_node[(size_t)grp * RC_GROUP + c];
        uint32_t e_out = RC_PAD;
        double w_out = 0.0;
        if (gnode != RC_PAD) {
            const uint32_t b = node_ptr[gnode], len = node_ptr[gnode + 1] - b, j = j0 + (uint32_t)lane;
            if (j < len) {
                const uint64_t key = keys[(size_t)b + j];
                const int tap = (int)(key & 63u);
                const uint32_t ordinal = (uint32_t)((key >> 6) & ((1ull << rbits) - 1ull));
                const uint32_t image = image_begin + gnode / (uint32_t)g.n_cp;
                const uint32_t row = poff[image] - own_pt_begin + ordinal;
                const uint32_t p = new_of_old[row];
                const float4 v = pos[p];
                const float in[3] = { v.x, v.y, v.z };
                double W[3][4];
                #pragma unroll
                for (int k = 0; k < 3; k++) {
                    const float coord = (float)(((double)in[k] - g.origin[k]) / g.spacing[k]);
                    const float fl = floorf(coord);
                    bspline_weights(W[k], (double)(coord - fl));
                }
                w_out = W[0][tap & 3] * W[1][(tap >> 2) & 3] * W[2][tap >> 4];
                e_out = by_row ? row : p;
            }
        }
        s_ent[c][lane] = e_out;
        s_wt[c][lane] = w_out;
    }
    __syncthreads();
    const size_t base = (size_t)group_ptr[grp] + (size_t)j0 * RC_GROUP;
    for (uint32_t idx = threadIdx.x; idx < 64u * RC_GROUP; idx += 256u) {
        const uint32_t j = idx / RC_GROUP, c = idx % RC_GROUP;
        if (j0 + j < n) { ent[base + idx] = s_ent[c][j]; wt[base + idx] = s_wt[c][j]; }
    }
}

// The scatter: thread = one component of one control point of one image (slot order; lane = 4 slot + component), its chain in
// point order with the arithmetic of imageGroup.cxx:330-337 -- f64 product, f64 sum, rounded to f32 after every addition -- points
// without sums skipped (:299).  Every control point is written (an empty chain leaves the Fill(0) of :249).
// Why a thread per COMPONENT: an addition is v_cvt_f64_f32 (g) -> v_add_f64 -> v_cvt_f32_f64, each waiting for the one before
// (the conversions run at a quarter of the vector rate), so a thread that carries all four components issues ~256 cycles of
// vector work per entry; on a coarse lattice (cfg 3 level 0: 70 400 chains of up to 6 400 entries, one wavefront per SIMD with a
// thread per control point) the longest chains alone took 2.5 ms.  Four times the wavefronts, a quarter of the work in each.
// Three steps of RC_UNROLL entries are in flight per thread: while step k is added, the sums of step k + 1's points are being
// gathered and step k + 2's entries fetched -- nothing else hides the two dependent memory round trips of a step (a step of 8 on
// level 0 still took 2.9 us, one round trip; 16 there).
template <int RC_UNROLL>
__global__ __launch_bounds__(64) void ref_chain_kernel(const uint32_t *ent, const double *wt, const uint64_t *group_ptr, const uint32_t *group_len,
                                                       const uint32_t *slot_node, const float4 *point_sums, uint32_t n_groups, float4 *gradf)
{
    const int t = threadIdx.x >> 2, c = threadIdx.x & 3;
    const uint32_t grp = blockIdx.y * gridDim.x + blockIdx.x;   // rc_grid()
    if (grp >= n_groups) return;
    const uint32_t slot = grp * RC_GROUP + t;
    const uint32_t n = group_len[grp];                          // a multiple of RC_UNROLL
    const uint32_t *ge = ent + group_ptr[grp] + t;
    const double *gw = wt + group_ptr[grp] + t;
    const float *sums = reinterpret_cast<const float *>(point_sums) + c;
    float g = 0.f;
    if (n) {
        const uint32_t last = n - 1;
        uint32_t p_cur[RC_UNROLL], p_nxt[RC_UNROLL], p_far[RC_UNROLL];
        double w_cur[RC_UNROLL], w_nxt[RC_UNROLL], w_far[RC_UNROLL];
        float s_cur[RC_UNROLL], s_nxt[RC_UNROLL];
        #pragma unroll
        for (int u = 0; u < RC_UNROLL; u++) { p_cur[u] = ge[(size_t)u * RC_GROUP]; w_cur[u] = gw[(size_t)u * RC_GROUP]; }
        #pragma unroll
        for (int u = 0; u < RC_UNROLL; u++) { const uint32_t j = min((uint32_t)(RC_UNROLL + u), last); p_nxt[u] = ge[(size_t)j * RC_GROUP]; w_nxt[u] = gw[(size_t)j * RC_GROUP]; }
        #pragma unroll
        for (int u = 0; u < RC_UNROLL; u++) s_cur[u] = sums[4 * (size_t)(p_cur[u] == RC_PAD ? 0u : p_cur[u])];
        for (uint32_t j0 = 0; j0 < n; j0 += RC_UNROLL) {
            #pragma unroll
            for (int u = 0; u < RC_UNROLL; u++) s_nxt[u] = sums[4 * (size_t)(p_nxt[u] == RC_PAD ? 0u : p_nxt[u])];
            #pragma unroll
            for (int u = 0; u < RC_UNROLL; u++) { const uint32_t j = min(j0 + 2 * RC_UNROLL + u, last); p_far[u] = ge[(size_t)j * RC_GROUP]; w_far[u] = gw[(size_t)j * RC_GROUP]; }
            #pragma unroll
            for (int u = 0; u < RC_UNROLL; u++) {
                // the point's sWeight sits in the quad's fourth lane (quad_perm [3, 3, 3, 3])
                const float sw = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s_cur[u]), 0xFF, 0xF, 0xF, false));
                if (p_cur[u] == RC_PAD || sw == 0.0f) continue;
                g = (float)((double)g + w_cur[u] * (double)s_cur[u]);
            }
            #pragma unroll
            for (int u = 0; u < RC_UNROLL; u++) { p_cur[u] = p_nxt[u]; w_cur[u] = w_nxt[u]; s_cur[u] = s_nxt[u]; p_nxt[u] = p_far[u]; w_nxt[u] = w_far[u]; }
        }
    }
    const uint32_t gnode = slot_node[slot];
    if (gnode != RC_PAD) reinterpret_cast<float *>(gradf + gnode)[c] = g;
}

// ---- min(probA(dist), probB(dist)) with the reference's arithmetic, evaluated only where it is needed ------------------------------
// inlier_probability_exact (stats.h:84-92 with its promotions: two f64 exponentials, five f32 divisions) is ~250 instructions,
// and a half-link has two of them.  The f32 form of the product path (k_links.hip.h inlier_probability) is within
// INLIER_PROBABILITY_BOUND = 2^-16 of it (derived there, measured against the reference build of stats.cxx), so it can DECIDE
// without changing a bit of the result:
//   * a half-link whose f32 weight is below threshold - THRESHOLD_BAND (1e-4 = 6 bounds) is below the threshold in the reference's
//     arithmetic too: the deformable step skips it (imageGroup.cxx:274) and its value is never needed;
//   * where one image's f32 probability is more than REF_ORDER_BAND below the other's, it is the smaller one in the reference's
//     arithmetic too, and min() returns ITS exact value: one evaluation instead of two.
// The evaluations that remain are written as requests (distance, image) to a queue in LDS and worked off 64 at a time by the
// whole wavefront -- a lane's own requests under a divergent mask would cost the wavefront every branch any lane takes.
constexpr float REF_ORDER_BAND = 1e-4f;
template <int U> struct ExactQueue {
    float d[2 * U * 64];            // request: distance; afterwards: the probability
    uint32_t img[2 * U * 64];
};

// want[u]: the weight of link u is needed.  Returns w[u] = min(exact pA, exact pB) for those (2.0f stands for "not evaluated").
template <int U>
__device__ __forceinline__ void exact_min_weights(const float (&dist)[U], const float (&d2)[U], const bool (&want)[U], uint32_t imgA, const uint32_t (&imgB)[U],
                                                  const EmDerived emdA, const EmDerived (&emdB)[U], const float4 *em, ExactQueue<U> &q, float (&w)[U])
{
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    int pos[U][2];
    uint32_t total = 0;
    #pragma unroll
    for (int u = 0; u < U; u++) {
        const float pa = inlier_probability(d2[u], emdA), pb = inlier_probability(d2[u], emdB[u]);
        const bool need[2] = { want[u] && !(pb < pa - REF_ORDER_BAND), want[u] && !(pa < pb - REF_ORDER_BAND) };
        #pragma unroll
        for (int k = 0; k < 2; k++) {
            const unsigned long long m = __ballot(need[k]);
            pos[u][k] = need[k] ? (int)(total + (uint32_t)__popcll(m & below)) : -1;
            total += (uint32_t)__popcll(m);
            if (need[k]) { q.d[pos[u][k]] = dist[u]; q.img[pos[u][k]] = k == 0 ? imgA : imgB[u]; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t r = lane; r < total; r += 64) q.d[r] = inlier_probability_exact(q.d[r], em[q.img[r]]);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    #pragma unroll
    for (int u = 0; u < U; u++) {
        const float ea = pos[u][0] >= 0 ? q.d[pos[u][0]] : 2.0f, eb = pos[u][1] >= 0 ? q.d[pos[u][1]] : 2.0f;
        w[u] = ref_min(ea, eb);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();            // the queue is free for the next step
}

// ---- the rows of half-links side by side (deformable per-point sums) ------------------------------------------------------------
// imageGroup.cxx:252-278 is one f32 chain per point over its half-links in readPairs order; points are independent.
// ref_point_sums_kernel walked the reference-order CSR with a thread per point: lane t reads link[rowptr[t] + j], 64 different
// cache lines per step, and the partner's image from a second, dependent 16-byte gather -- 5.0 ms per launch at cfg 3 for 1 ms of
// arithmetic.  Built once per context (the CSR never changes): rows sorted by length, 64 to a wavefront, entry j of the 64 rows side
// by side and padded to the group's longest; per entry the partner point (internal numbering) and its image.  Same chain per
// point, same weights: identical bits.
constexpr int RR_UNROLL = 4;

__global__ __launch_bounds__(256) void ref_rows_len_kernel(const uint64_t *rowptr, uint32_t n_rows, uint32_t *len, uint32_t *iota)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    len[r] = (uint32_t)(rowptr[r + 1] - rowptr[r]);
    iota[r] = r;
}

__global__ __launch_bounds__(256) void ref_rows_group_kernel(const uint32_t *len_sorted, uint32_t n_groups, uint32_t *group_len, uint64_t *group_size)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_groups) return;
    const uint32_t padded = (len_sorted[(size_t)t * 64] + RR_UNROLL - 1) / RR_UNROLL * RR_UNROLL;
    group_len[t] = padded;
    group_size[t] = (uint64_t)padded * 64;
}

// thread = slot: its row's half-links into the seats of the layout (partner point, partner image)
__global__ __launch_bounds__(64) void ref_rows_fill_kernel(const uint64_t *rowptr, const uint32_t *link, const float4 *pos, const uint32_t *slot_row,
                                                           const uint64_t *group_ptr, uint32_t *ent, uint16_t *ent_img)
{
    const uint32_t r = slot_row[blockIdx.x * 64 + threadIdx.x];
    if (r == RC_PAD) return;
    const size_t base = (size_t)group_ptr[blockIdx.x] + threadIdx.x;
    const uint64_t l0 = rowptr[r], l1 = rowptr[r + 1];
    for (uint64_t l = l0; l < l1; l++) {
        const uint32_t b = link[l];
        ent[base + (size_t)(l - l0) * 64] = b;
        ent_img[base + (size_t)(l - l0) * 64] = (uint16_t)__float_as_int(pos[b].w);
    }
}

__global__ __launch_bounds__(64) void ref_point_sums_rows_kernel(const uint32_t *ent, const uint16_t *ent_img, const uint64_t *group_ptr,
                                                                 const uint32_t *group_len, const uint32_t *slot_row, const uint32_t *new_of_old,
                                                                 const float4 *pos, const P3 *pos2, const float4 *em, const EmDerived *emd,
                                                                 float threshold, float4 *point_sums, float4 *row_sums, double *pt_energy)
{
    __shared__ ExactQueue<RR_UNROLL> queue;
    const uint32_t r = slot_row[blockIdx.x * 64 + threadIdx.x];
    const uint32_t n = group_len[blockIdx.x];
    const uint32_t *ge = ent + group_ptr[blockIdx.x] + threadIdx.x;
    const uint16_t *gi = ent_img + group_ptr[blockIdx.x] + threadIdx.x;
    const uint32_t a = r == RC_PAD ? 0u : new_of_old[r];
    const P3 pA = pos2[a];
    const uint32_t imgA = (uint32_t)__float_as_int(pos[a].w);
    const EmDerived emdA = emd[imgA];
    float sx = 0, sy = 0, sz = 0, sw = 0;
    double ed = 0, ew = 0;
    for (uint32_t j0 = 0; j0 < n; j0 += RR_UNROLL) {
        uint32_t b[RR_UNROLL], im[RR_UNROLL];
        P3 pB[RR_UNROLL];
        EmDerived emdB[RR_UNROLL];
        float d2[RR_UNROLL], dist[RR_UNROLL], w[RR_UNROLL];
        bool want[RR_UNROLL];
        #pragma unroll
        for (int u = 0; u < RR_UNROLL; u++) { b[u] = ge[(size_t)(j0 + u) * 64]; im[u] = gi[(size_t)(j0 + u) * 64]; }
        #pragma unroll
        for (int u = 0; u < RR_UNROLL; u++) { pB[u] = pos2[b[u] == RC_PAD ? 0u : b[u]]; emdB[u] = emd[b[u] == RC_PAD ? 0u : im[u]]; }
        #pragma unroll
        for (int u = 0; u < RR_UNROLL; u++) {
            // vtkMath::Distance2BetweenPoints(pA, pB): (a - b)^2 summed x, y, z in f32
            const float ex = pA.x - pB[u].x, ey = pA.y - pB[u].y, ez = pA.z - pB[u].z;
            d2[u] = ex * ex + ey * ey + ez * ez;
            dist[u] = ref_sqrt(d2[u]);
            // certainly below the threshold (see exact_min_weights): skipped as the reference skips it, never evaluated
            want[u] = b[u] != RC_PAD && !(fminf(inlier_probability(d2[u], emdA), inlier_probability(d2[u], emdB[u])) < threshold - THRESHOLD_BAND);
        }
        exact_min_weights<RR_UNROLL>(dist, d2, want, imgA, im, emdA, emdB, em, queue, w);
        #pragma unroll
        for (int u = 0; u < RR_UNROLL; u++) {
            if (!want[u]) continue;
            const float w2 = w[u] * w[u];
            if (w[u] < threshold) continue;
            ew += (double)w2;
            ed += (double)(w2 * d2[u]);
            sx += w2 * (pB[u].x - pA.x); sy += w2 * (pB[u].y - pA.y); sz += w2 * (pB[u].z - pA.z);
            sw += w2;
        }
    }
    if (r == RC_PAD) return;
    point_sums[a] = make_float4(sx, sy, sz, sw);
    if (row_sums) row_sums[r] = make_float4(sx, sy, sz, sw);        // the same by owned row (reference order): ref_chain_fill_kernel
    if (pt_energy) { pt_energy[2 * (size_t)r] = ed; pt_energy[2 * (size_t)r + 1] = ew; }
}

// the points' energy terms of one image added in point order: 256 points per step, thread 0 adds sDistances and thread 1 sWeights
// from LDS while the next step's 256 points are on their way from memory
constexpr int RE_STEP = 256;
__global__ __launch_bounds__(RE_STEP) void ref_image_energy2_kernel(const double *pt_energy, const uint32_t *poff, uint32_t image_begin,
                                                                    uint32_t own_pt_begin, double *img_energy)
{
    __shared__ double buf[2][RE_STEP];
    const uint32_t r0 = poff[image_begin + blockIdx.x] - own_pt_begin, r1 = poff[image_begin + blockIdx.x + 1] - own_pt_begin;
    double acc = 0;
    double n0 = 0, n1 = 0;
    if (r0 + threadIdx.x < r1) { n0 = pt_energy[2 * (size_t)(r0 + threadIdx.x)]; n1 = pt_energy[2 * (size_t)(r0 + threadIdx.x) + 1]; }
    for (uint32_t base = r0; base < r1; base += RE_STEP) {
        buf[0][threadIdx.x] = n0; buf[1][threadIdx.x] = n1;
        __syncthreads();
        const uint32_t r = base + RE_STEP + threadIdx.x;
        if (r < r1) { n0 = pt_energy[2 * (size_t)r]; n1 = pt_energy[2 * (size_t)r + 1]; }
        if (threadIdx.x < 2) {
            const double *b = buf[threadIdx.x];
            const uint32_t n = min((uint32_t)RE_STEP, r1 - base);
            uint32_t j = 0;
            for (; j + 8 <= n; j += 8) {
                double v[8];
                #pragma unroll
                for (int u = 0; u < 8; u++) v[u] = b[j + u];
                #pragma unroll
                for (int u = 0; u < 8; u++) acc += v[u];
            }
            for (; j < n; j++) acc += b[j];
        }
        __syncthreads();
    }
    if (threadIdx.x < 2) img_energy[2 * blockIdx.x + threadIdx.x] = acc;
}

// ---- the linear step's chain per image, fed by producer wavefronts ------------------------------------------------------------------
// updateLinearTransforms (imageGroup.cxx:1080-1143) adds 18 f64 sums per image over its half-links in order: 1e6 dependent
// additions per sum at cfg 3, a floor of ~3.5 ms at 8 cycles each.  ref_linear_chain_kernel (one wavefront per image) formed 64
// half-links' terms, waited, added, waited: 24 ms, three quarters of it the memory round trips of the term stage, plus 8.8 ms for
// the weights in a kernel of their own.  Here the weights come from ref_link_weights_kernel (whole chip, rows of half-links side by
// side, the exact form evaluated by request), and the chain's block is 7 producer wavefronts and one consumer: while the consumer's
// 18 lanes add the 7 x 64 half-links of one LDS buffer in order, the producers fill the other with the next 7 x 64 half-links' terms.
constexpr int RL_PRODUCERS = 7;

__global__ __launch_bounds__(256) void ref_link_static_kernel(const uint64_t *rowptr, const uint32_t *link, const uint32_t *new_of_old, uint32_t n_rows,
                                                              const float4 *pos, uint32_t *own, uint16_t *link_img)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint32_t a = new_of_old[r];
    for (uint64_t l = rowptr[r]; l < rowptr[r + 1]; l++) { own[l] = a; link_img[l] = (uint16_t)__float_as_int(pos[link[l]].w); }
}

// dist = sqrt(|pB - pA|^2) and w = min(probA(dist), probB(dist)) of every half-link (imageGroup.cxx:1086-1099), reference order:
// block = one wavefront over 4 x 64 consecutive half-links of ONE image (grid.y), on the whole chip -- inside the chain kernel the
// weights kept the 4 SIMDs of each image's CU busy for 4 us per 448 half-links, twice what its consumer needs.
constexpr int RW_UNROLL = 4;
__global__ __launch_bounds__(64) void ref_link_weights_kernel(const uint64_t *img_link, const uint32_t *link, const uint32_t *own, const uint16_t *link_img,
                                                              const P3 *pos2, const float4 *em, const EmDerived *emd, uint32_t image_begin,
                                                              float *w_out, float *d_out)
{
    __shared__ ExactQueue<RW_UNROLL> queue;
    const uint64_t l0 = img_link[blockIdx.y], l1 = img_link[blockIdx.y + 1];
    const uint64_t base = l0 + (uint64_t)blockIdx.x * (64 * RW_UNROLL);
    if (base >= l1) return;
    const uint32_t imgA = image_begin + blockIdx.y;
    const EmDerived emdA = emd[imgA];
    uint64_t l[RW_UNROLL];
    uint32_t im[RW_UNROLL];
    EmDerived emdB[RW_UNROLL];
    float d2[RW_UNROLL], dist[RW_UNROLL], w[RW_UNROLL];
    bool want[RW_UNROLL];
    #pragma unroll
    for (int u = 0; u < RW_UNROLL; u++) {
        l[u] = base + (uint64_t)u * 64 + threadIdx.x;
        want[u] = l[u] < l1;
        const uint64_t lc = want[u] ? l[u] : l1 - 1;
        const P3 pA = pos2[own[lc]], pB = pos2[link[lc]];
        im[u] = link_img[lc];
        emdB[u] = emd[im[u]];
        const float dx = pB.x - pA.x, dy = pB.y - pA.y, dz = pB.z - pA.z;
        d2[u] = dx * dx + dy * dy + dz * dz;
        dist[u] = ref_sqrt(d2[u]);
    }
    exact_min_weights<RW_UNROLL>(dist, d2, want, imgA, im, emdA, emdB, em, queue, w);
    #pragma unroll
    for (int u = 0; u < RW_UNROLL; u++)
        if (want[u]) { w_out[l[u]] = w[u]; d_out[l[u]] = dist[u]; }
}

__global__ __launch_bounds__(64 * (RL_PRODUCERS + 1)) void ref_linear_chain2_kernel(const uint64_t *img_link, const uint32_t *link, const uint32_t *own,
                                                                                    const float *w_in, const float *d_in, const P3 *pos2,
                                                                                    uint32_t image_begin, double *mat, float linear_alpha, int use_scale,
                                                                                    double *img_energy)
{
    // The terms wait in LDS as DOUBLES: the consumer's addition then is ds_read_b64 + v_add_f64; with f32 terms every addition
    // dragged a v_cvt_f64_f32 (a quarter-rate instruction) through the consumer's one wavefront -- 23 cycles per half-link, 9.5 ms
    // per launch at cfg 3, where the dependent additions alone need ~8.
    __shared__ double terms[2][RL_PRODUCERS][64][LINEAR_SUMS + 1];
    __shared__ double sums[LINEAR_SUMS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t l0 = img_link[blockIdx.x], l1 = img_link[blockIdx.x + 1];
    const uint64_t n_chunks = (l1 - l0 + 63) / 64;
    const uint64_t n_phases = (n_chunks + RL_PRODUCERS - 1) / RL_PRODUCERS;
    double acc = 0.0;
    // A producer keeps three chunks in flight: the indices and weights of phase ph + 2, the gathered coordinates of phase ph + 1,
    // the arithmetic of phase ph.  (Indices past the image's last half-link are clamped to it; their terms are never written.)
    const uint64_t lmax = l1 > l0 ? l1 - 1 : l0;
    auto at = [&](uint64_t ph) { return min(l0 + (ph * RL_PRODUCERS + (uint64_t)(wave - 1)) * 64 + lane, lmax); };
    uint32_t ia = 0, ib = 0;                    // indices: own point, partner
    float iw = 0, id = 0, vw = 0, vd = 0;       // weight and distance riding with the indices / with the coordinates
    P3 pA{}, pB{};
    if (wave > 0 && n_phases) {
        uint64_t l = at(0);
        ia = own[l]; ib = link[l]; vw = w_in[l]; vd = d_in[l];
        pA = pos2[ia]; pB = pos2[ib];
        l = at(1);
        ia = own[l]; ib = link[l]; iw = w_in[l]; id = d_in[l];
    }
    // phase ph: producers fill buffer ph & 1 with chunks ph * 7 .. ph * 7 + 6; the consumer adds buffer (ph - 1) & 1
    for (uint64_t ph = 0; ph <= n_phases; ph++) {
        if (wave > 0) {
            if (ph < n_phases) {
                const P3 cA = pA, cB = pB;
                const float w = vw, dist = vd;
                pA = pos2[ia]; pB = pos2[ib]; vw = iw; vd = id;             // phase ph + 1
                const uint64_t lf = at(ph + 2);
                ia = own[lf]; ib = link[lf]; iw = w_in[lf]; id = d_in[lf];  // phase ph + 2
                const uint64_t l = l0 + (ph * RL_PRODUCERS + (uint64_t)(wave - 1)) * 64 + lane;
                if (l < l1) {
                    double *t = terms[ph & 1][wave - 1][lane];
                    const float a[3] = { cA.x, cA.y, cA.z }, b[3] = { cB.x, cB.y, cB.z };
                    #pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const float diff = b[k] - a[k];
                        t[k] = (double)(w * diff);
                        t[3 + k] = (double)(w * a[k]);
                        t[6 + k] = (double)(w * b[k]);
                        t[9 + k] = (double)(w * a[k] * a[k]);
                        t[12 + k] = (double)(w * b[k] * b[k]);
                    }
                    t[15] = (double)w;
                    t[16] = (double)(w * w * dist * dist);
                    t[17] = (double)(w * w);
                }
            }
        } else if (ph > 0 && lane < LINEAR_SUMS) {
            const uint64_t first = l0 + (ph - 1) * RL_PRODUCERS * 64;
            for (int ck = 0; ck < RL_PRODUCERS; ck++) {
                const uint64_t cb = first + (uint64_t)ck * 64;
                if (cb >= l1) break;
                const double (*tc)[LINEAR_SUMS + 1] = terms[(ph - 1) & 1][ck];
                if (l1 - cb >= 64) {
                    #pragma unroll 16
                    for (int j = 0; j < 64; j++) acc += tc[j][lane];
                } else {
                    const int n = (int)(l1 - cb);
                    for (int j = 0; j < n; j++) acc += tc[j][lane];
                }
            }
        }
        __syncthreads();
    }
    if (wave == 0 && lane < LINEAR_SUMS) sums[lane] = acc;
    __syncthreads();
    if (wave == 0 && lane < 3) ref_linear_update_axis(mat + (size_t)(image_begin + blockIdx.x) * 16, lane, sums, linear_alpha, use_scale);
    if (threadIdx.x == 0) { img_energy[2 * blockIdx.x] = sums[16]; img_energy[2 * blockIdx.x + 1] = sums[17]; }
}

} // namespace frog
