// ctx.h -- device context of libfrog_hip.so: buffers, layout tables, error plumbing.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <memory>
#include <vector>

#include "../../../include/frog_hip.h"

namespace frog {

extern thread_local std::string g_last_error;

inline int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

#define FROG_HIP_CHECK(expr)                                                            \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return ::frog::fail(FROG_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Owning device allocation.
// alloc() keeps the old block when it is large enough (lattices are re-created many
// times per run; hipFree/hipMalloc synchronise the device and cost milliseconds).
template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    size_t cap = 0;
    bool borrowed = false;      // p points into another DevBuf's block (borrow()): nothing to free
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p && !borrowed) (void)hipFree(p);
        p = nullptr; n = 0; cap = 0; borrowed = false;
    }
    void borrow(T *from, size_t count) { release(); p = from; n = count; cap = count; borrowed = true; }
    // `reserve` (>= count): capacity to allocate when a new block is needed at all
    hipError_t alloc(size_t count, size_t reserve = 0)
    {
        if (count <= cap && p) { n = count; return hipSuccess; }
        release();
        if (!count) return hipSuccess;
        const size_t want = reserve > count ? reserve : count;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess && want > count) {              // no room for the head-room: take what is needed
            (void)hipGetLastError();
            e = hipMalloc((void **)&p, count * sizeof(T));
            if (e == hipSuccess) { n = count; cap = count; }
            return e;
        }
        if (e == hipSuccess) { n = count; cap = want; }
        return e;
    }
    // stream-ordered allocation from the device's pool: no device synchronisation (used on the regrid path)
    hipError_t alloc_async(size_t count, hipStream_t s)
    {
        release();
        if (!count) return hipSuccess;
        hipError_t e = hipMallocAsync((void **)&p, count * sizeof(T), s);
        if (e != hipSuccess) { (void)hipGetLastError(); return alloc(count); }
        n = count; cap = count;
        return e;
    }
    template <class A> hipError_t upload(const std::vector<T, A> &v, hipStream_t s)
    {
        hipError_t e = alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s);
    }
    size_t bytes() const { return n * sizeof(T); }
};

// ---- link layout (built on the host in prep.h) --------------------------------

constexpr int TILE_POINTS = 256;     // points per sweep tile (one wavefront each)
constexpr int N_XCD = 8;             // blocks are dealt round-robin over the XCDs
constexpr int MAX_SUBPASS = 8;       // partner groups handled one after the other on each XCD: at most this many
constexpr int MAX_GROUPS = N_XCD * MAX_SUBPASS;
// Records are stored in chunks of REC_CHUNK = two sweep steps, transposed so that ONE 16-byte
// load per lane fetches the lane's records of both steps (8-byte loads run at 0.54-0.70x the rate
// of 16-byte loads on gfx950): record k of a (tile, group) range lies at
//   (k / 128) * 128 + (k % 64) * 2 + (k % 128) / 64.
// Every range is padded with null records to a whole number of chunks.
constexpr int REC_CHUNK = 128;
// Layout of the per-XCD partial sums of the deformable sweep (group_sums): false = [group][point] (a wavefront of the
// sweep stores 64 consecutive float4: 8 full lines; the scatter reads 8 lines per point, about half of each used);
// true = [point][group] (one 128-byte line per point for the scatter; the sweep's stores become 16-byte pieces of 64
// lines that eight XCDs complete).  Measured (cfg 3): the second form takes the sweep from 0.32 to 0.475 ms -- partial
// lines written from eight L2s -- and the scatter from 0.121 only to 0.113-0.120 ms.  Kept at false.
#ifndef FROG_SUMS_POINT_MAJOR
#define FROG_SUMS_POINT_MAJOR 0
#endif
constexpr bool SUMS_POINT_MAJOR = FROG_SUMS_POINT_MAJOR != 0;
__host__ __device__ inline size_t group_sum_index(uint32_t group, uint32_t point, uint32_t own_points)
{
    return SUMS_POINT_MAJOR ? (size_t)point * N_XCD + group : (size_t)group * own_points + point;
}
// Partner-image groups: n_groups = 8 * n_sub.  The sweep is launched n_sub times; launch `sub`
// lets XCD x read group sub*8 + x and continues XCD x's partial sums, so that a group's xyz2 slice
// can be made small enough to stay in one 4 MiB L2.  n_sub is 1 unless FROG_SUBPASSES sets it:
// splitting further has not paid anywhere it was measured --
//   100 images x 20 000 points (24 MB of xyz2, 8 slices of 3 MB): 16 groups 0.85 ms vs 0.74 ms;
//   400 images x 20 000 points, 4.8e8 half-links (96 MB, 8 slices of 12 MB): 32 groups in 4 launches
//   5.56 ms vs 5.16 ms (deformable), 4.43 vs 5.06 ms (linear).  There the sweep costs 2x per link
//   because an image pair holds 3 000 matches instead of 10 000: the 64 gathers of a step fall on
//   ~64 different lines whatever the slice size.

// xyz2 of a point, packed: the sweep gathers 12 bytes per end point.
struct P3 {
    float x, y, z;
};

// One sweep tile: a run of consecutive points of ONE image and all their
// half-links, sorted by partner image (then by point).
struct Tile {
    uint32_t pt_begin;      // global point index (internal numbering)
    uint32_t pt_count;
    uint32_t rec_begin;     // index into LinkRec array (a multiple of REC_CHUNK)
    uint32_t image;
    uint32_t group_off[MAX_GROUPS]; // records into partner group g start at rec_begin + off[g] (a multiple of REC_CHUNK)
    uint32_t group_cnt[MAX_GROUPS]; // ... and there are cnt[g] of them
};
static_assert(sizeof(Tile) % 16 == 0, "Tile is loaded with vector loads");

// One half-link.  Wide form, 8 bytes (the size of the reference's Link, point.h:11-16):
//   a = (partner image << 8) | index of the own point inside its tile (TILE_POINTS <= 256)
//   b = global index of the partner point
// The partner image rides in the record so that its EM constants can be fetched
// together with the coordinate gathers instead of after them.
// Narrow form, 4 bytes, used whenever it fits (RecFormat):
//   [ partner point inside its image | partner image inside its group | own point ]
//     pt_bits                          img_bits                         8
// The sweep is bound by the time its cache misses spend in flight (records come from HBM,
// ~5x the latency of the L2-resident coordinates), so halving the record stream pays directly.
struct LinkRec {
    uint32_t a;
    uint32_t b;
};
struct RecFormat {
    uint32_t narrow;        // 1: 4-byte records
    uint32_t img_bits;      // narrow only
};
static_assert(TILE_POINTS <= 256, "own-point index must fit 8 bits of LinkRec::a");

// Per-image constants derived from (c1, c2, ratio) for getInlierProbability (stats.h:84-92), k_links.hip.h: with
// inv_k = 1 / (c_k + eps), c0 = 0.797884560802865f:  kq1 = ratio c0 inv1^3, kq2 = (1 - ratio) c0 inv2^3 and
// s_k = -log2(e) inv_k^2 / 2, so that x_k = kq_k d2 2^(s_k d2) -- no square root and no division on the way.
struct EmDerived {
    float kq1, kq2, s1, s2;
};
// The deformable sweeps' weight (k_links.hip.h inlier_weight_pair): p = 1 / (1 + x2/x1 + eps/x1) with the eps term left out
// is 1 / (1 + 2^(l + ds d2)), l = log2(kq2 / kq1), ds = s2 - s1 -- ONE exponential -- and min(pA, pB) is that function of the
// LARGER of the two images' exponents: one exponential and one reciprocal per half-link instead of four and two.  Leaving
// eps = 1e-10 out changes p by p eps / (x1 + x2 + eps), so the form is used for lo <= d2 <= hi only, a range on which the
// image's mixture density x1 + x2 is certainly >= EM_FAST_DENSITY (k_stats.hip.h em_fast_of); outside it the sweep evaluates
// inlier_probability as before.  An empty range (lo = +inf, hi = -inf) for mixtures the form is not derived for.
struct EmFast {
    float l, ds, lo, hi;
};

struct GridGeom {
    int dims[3];
    int n_cp;               // dims[0]*dims[1]*dims[2]
    double origin[3];
    double spacing[3];
    int cells[3];           // dims - 3
    int brick;              // cells per brick edge (8 or 4)
    int nbricks[3];
    int n_bricks;           // per image
    // Layout of the lattices coeff / grad / gradf / grad_spare (k_grid.hip.h lat()): image-major [image][node], or -- `blocked`,
    // fine lattices of many images -- [node / 16][image][node % 16]: the lattice step works through ALL images of 16 nodes at a time,
    // and image-major those are 256-byte pieces 16 n_cp bytes apart (cfg 5 level 4: 500 pieces 14.7 MB apart per pass)
    bool blocked = false;
    uint32_t lat_images = 0;    // owned images the lattices are laid out for
    // `sparse`: entries are kept for ACTIVE (image, node) pairs only -- the nodes in the 4^3 stencil of any of the image's points
    // (frog_ctx::lat_mask: per owned image a bit per node, mask_words words).  `pos` stands still while a lattice stands,
    // so the set is fixed for the lattice's life; a pair outside it never receives a gradient and is never read by the transform
    // of the image's points: its coefficient is 0 minus the node's means so far, the SAME float for every such image of the node
    // (frog_ctx::ucoeff).  The lattice step then touches 43 % of the pairs of cfg 5's finest lattice instead of all of them.
    bool sparse = false;
    uint32_t mask_words = 0;
    size_t lat_entries() const { return blocked ? (size_t)((n_cp + 15) / 16) * 16 * lat_images : (size_t)lat_images * (size_t)n_cp; }
};

struct GridRecord {         // a finished or current lattice of the chain
    frog_grid_info info;
    bool blocked = false;       // layout of `kept` (GridGeom::blocked when the lattice was retired)
    uint32_t lat_images = 0;
    bool sparse = false;        // `kept` holds the active pairs only: kept_mask / kept_u say which, and what the others are
    uint32_t mask_words = 0;
    std::shared_ptr<DevBuf<uint32_t>> kept_mask;
    std::shared_ptr<DevBuf<float4>> kept_u;
    std::shared_ptr<DevBuf<float4>> kept;   // [owned images][G] coefficients, filled (device copy) when the lattice is retired
    bool retired = false;
};

} // namespace frog

struct frog_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    frog_options opt{};

    uint32_t nI = 0, ib = 0, ie = 0;          // images, owned range
    uint64_t P = 0;                           // all points
    uint32_t own_pt_begin = 0, own_pt_end = 0;
    uint64_t L_own = 0;                       // half-links of owned images
    uint64_t L_recs = 0;                      // records stored (ranges padded to whole chunks)
    std::vector<uint32_t> poff;               // host copy
    std::vector<uint64_t> img_link_begin;     // per image: first half-link ordinal base (ref-order CSR, owned rows only)

    // points
    frog::DevBuf<float4> pos;                 // xyz | image id
    frog::DevBuf<frog::P3> pos2;              // xyz2, packed
    frog::DevBuf<frog::P3> pos2_spec;         // xyz2 computed ahead of the caller's transformPoints(0) (owned rows)
    frog::DevBuf<uint32_t> d_poff;            // [nI+1]
    // reference-order CSR of the owned rows (for the reservoir ordinals)
    frog::DevBuf<uint64_t> ref_rowptr;        // [ownP + 1], relative to the first owned link
    frog::DevBuf<uint32_t> ref_link;          // [L_own] partner global index
    // sweep layout
    std::vector<uint32_t> h_old_of_new, h_new_of_old;   // internal (Morton) <-> reference point numbering
    frog::DevBuf<uint32_t> new_of_old;        // [ownP] for the owned rows (reservoir: ordinal -> point)
    frog::DevBuf<frog::Tile> tiles;
    frog::DevBuf<frog::LinkRec> recs;         // wide records, or ...
    frog::DevBuf<uint32_t> recs32;            // ... narrow records (exactly one of the two is filled)
    frog::RecFormat rec_format{};
    uint32_t n_tiles = 0;
    frog::DevBuf<uint32_t> img_tile_ptr;      // [nI+1] tiles of image (owned only non-empty)
    std::vector<uint32_t> h_img_tile_ptr;
    uint32_t n_sub = 1, n_groups = frog::N_XCD;      // sweep launches per pass, partner groups (= 8 * n_sub)
    uint32_t group_begin[frog::MAX_GROUPS + 1] = {}; // partner-image groups (prep.h)
    frog::DevBuf<double> tile_partial;        // [n_tiles][n_groups][18]
    frog::DevBuf<long long> tile_counts;      // [n_tiles][n_groups][2]
    frog::DevBuf<float4> group_sums;          // [N_XCD][ownP] per-point partial sums (one buffer per XCD, continued across sub-passes)
    frog::DevBuf<float4> point_sums;          // [P] (owned rows used)
    // fused deformable sweep (k_links.hip.h FUSED): block -> tile table and whether this context uses it
    frog::DevBuf<uint32_t> tile_order;        // [n_order_blocks] tile of every block, 0xFFFFFFFF = none; block % 8 = the tile's eighth of its image
    frog::DevBuf<frog::Tile> tiles_bo;        // the tiles in block order (zero tiles where tile_order has none)
    uint32_t n_order_blocks = 0;
    bool fused_sweep = false;
    bool fused_forced = false;                // FROG_SWEEP_FUSED=1 / 2: also without a culling list

    // statistics
    frog::DevBuf<float4> em;                  // [nI] c1,c2,ratio,0
    frog::DevBuf<frog::EmDerived> emd;        // [nI]
    frog::DevBuf<frog::EmFast> emf;           // [nI] one-exponential form of the deformable sweeps' weight
    frog::DevBuf<float> samples;              // [nOwned][cap]
    frog::DevBuf<unsigned char> em_guess;     // [nOwned][4][EM_GUESS_BATCHES] exponents of the EM sums' trajectories, last fit (k_stats.hip.h)
    // Which ordinals a refresh keeps does not depend on the data (k_stats.hip.h: only on virtualSize, the capacity and the
    // number of earlier refreshes), so the selections of the NEXT refreshes -- and the end points of the kept half-links --
    // are computed ahead of time on a side stream into a ring of sel_ring buffers: sel_ring - 1 of them are filled when
    // the context is created, every refresh consumes one and queues the production of one more.  (Two buffers, i.e. one
    // refresh ahead, put the 2.4 ms replay of the generator on the critical path of a context whose ten iterations take
    // less than that: one rank of eight.)
    // Round 4: the ring holds, as a rule, the selections of a WHOLE run (80 buffers: the reference's default schedule has 65
    // refreshes, a few more with the guard's regrids; 12 bytes per kept half-link and buffer, up to 1 GB), all replayed at
    // frog_create, and a refresh queues a new replay only when fewer than SEL_LOW_WATER selections are left -- the 1.5 ms
    // replay otherwise overlapped the sweeps of every refresh's iteration and cost each a resident block per CU (0.278
    // instead of 0.250 ms).
    static constexpr int SEL_RING_MAX = 96;
    static constexpr int SEL_LOW_WATER = 3;
    int sel_ring = 2;
    frog::DevBuf<uint32_t> sample_ord[SEL_RING_MAX];     // [nOwned][cap]
    frog::DevBuf<uint32_t> sample_count[SEL_RING_MAX];   // [nOwned]
    frog::DevBuf<uint2> sample_ends[SEL_RING_MAX];       // [nOwned][cap] (own point, partner point) of every kept half-link, internal numbering
    uint64_t sel_consumed = 0;                // refreshes done
    uint64_t sel_produced = 0;                // selections queued on the side stream (refresh k lives in buffer k % sel_ring)
    int sel_used = 0;                         // buffer the last refresh consumed
    hipStream_t side = nullptr;
    // The device work of a lattice set-up (zeroing, the sort of the points by (image, brick, cell), the scatter's block table)
    // runs on a stream of its own: nothing needs its products before the first scatter on the new lattice, so it overlaps
    // with the statistics refresh and the half-link sweep that open the level (frog_deformable_setup_bounds, join_setup)
    hipStream_t setup_stream = nullptr;
    hipEvent_t setup_fork = nullptr, setup_join = nullptr;
    bool setup_pending = false;               // host side: setup_join not yet waited for on `stream`
    bool setup_deferred = false;              // host side: the set-up's kernels are not queued yet (join_setup queues them)
    bool setup_async = true;                  // FROG_SETUP_STREAM=0: everything on `stream`
    hipEvent_t energy_copied = nullptr;       // the four scalars of the last step are in h_energy
    bool xyz2_exported = false;               // frog_comm_buffer handed out pos2: its address must not change
    hipEvent_t sel_done[SEL_RING_MAX] = {};   // selection in buffer b complete (side stream)
    hipEvent_t ord_read[SEL_RING_MAX] = {};   // last reader of buffer b done (main stream)
    frog::DevBuf<uint32_t> mt_state;          // [nOwned][625] (624 words + index)
    std::vector<uint32_t> h_virtual;          // per owned image: virtualSize (clamped to 2^32-1)
    frog::DevBuf<uint32_t> d_virtual;         // [nOwned]
    int sample_cap = 0;

    // linear
    frog::DevBuf<double> mat;                 // [nI][16]; owned rows live
    // energy / counters
    frog::DevBuf<double> energy;              // [4]
    frog::DevBuf<double> energy_blocks;       // [ENERGY_BLOCKS][2] stage-1 sums
    frog::DevBuf<double> img_energy;          // [nOwned][2] per-image (sDistances, sWeights) of the last linear step
    frog::DevBuf<unsigned int> energy_ticket; // [0] blocks done in energy_reduce_kernel
    double *h_energy = nullptr;               // pinned [8]
    double *h_energy_dev = nullptr;           // the same memory as the device sees it (null: scalars come by copy)
    uint64_t scalar_seq = 0;                  // steps whose scalars were handed over through h_energy[7]

    // deformable
    bool deformable = false;
    frog::GridGeom geom{};
    frog::DevBuf<float4> coeff;               // [nOwned][G]
    frog::DevBuf<float4> grad;                // [nOwned][G] proposed coefficients (xyz), gradient weight (w)
    frog::DevBuf<float4> gradf;               // [nOwned][G] gradient lattice: sum w*sDisp xyz, sum w*sWeight
    frog::DevBuf<float4> grad_spare;          // a third lattice: the proposals of a step queued before the previous one's decision (frog_step_speculate)
    // sparse lattices (GridGeom::sparse): the active pairs, and per node the value of its inactive pairs -- companions of coeff / grad /
    // grad_spare, exchanged with them
    frog::DevBuf<uint32_t> lat_mask;          // [owned images][mask_words]: a bit per node
    frog::DevBuf<uint32_t> lat_inactive;      // [G] owned images for which the node is inactive
    frog::DevBuf<float4> ucoeff, ugrad, ugrad_spare;   // [G]
    frog::DevBuf<double> gridsum;             // [3G] (+ 4: frog_comm_mode)
    frog::DevBuf<uint32_t> perm;              // owned points sorted by (image, brick)
    frog::DevBuf<uint32_t> key_ptr;           // [nOwned*n_bricks*B^3 + 1] (image, brick, cell) -> perm range
    frog::DevBuf<uint32_t> key_cursor;
    frog::DevBuf<uint32_t> key_counts, brick_ptr_scratch, scan_sums;   // set-up scratch, kept between lattices
    frog::DevBuf<unsigned char> scatter_blocks; // ScatterBlock[...] longest first (k_grid.hip.h); the count is on the device
    frog::DevBuf<unsigned char> scatter_blocks_tmp; // the same blocks in brick order
    frog::DevBuf<uint32_t> len_hist;          // [2][SCATTER_CHUNK + 1] block-length histogram, cursors
    uint32_t n_scatter_blocks = 0;            // launch grid of the scatter: an upper bound of the block count
    uint32_t scatter_chunk = 384;             // points per scatter block of this context (<= SCATTER_CHUNK; frog_create)
    frog::DevBuf<float> bounds_scratch;       // [BOUNDS_BLOCKS][6] per-block min xyz, max xyz
    frog::DevBuf<unsigned int> stray;         // [0], [1]: points the scatter of an even / odd step found outside every brick (their taps went to
                                              // gradf), [2]: running total
    uint32_t stray_parity = 0;                // which of the two the current step uses
    bool centered_in_a = false;               // phase A of the current step also did phase B's work (whole-group context)
    frog::DevBuf<long long> img_counts;       // [nOwned][2] inliers, outliers
    std::vector<double> h_img_bbox;           // [nI][6] bbox of the model xyz per image (min xyz, max xyz)
    std::vector<frog::GridRecord> grids;
    // hard links (landmark constraints) of owned points, internal numbering: hl_point[n_hard], CSR hl_ptr, partners
    uint32_t n_hard = 0;
    float hard_weight2 = 0;
    frog::DevBuf<uint32_t> hl_point, hl_ptr, hl_partner;
    frog::DevBuf<double> hl_partial;          // [n_hard][2]
    frog::DevBuf<uint32_t> perm_tmp;          // second buffer of the cell-order pass
    frog::DevBuf<uint32_t> perm_key;          // (image, brick, cell) key of every sorted slot
    frog::DevBuf<float4> scatter_stage;       // [scatter blocks][(B+3)^3] tiles of the last scatter
    frog::DevBuf<uint32_t> brick_slot_ptr;    // [owned images * bricks + 1] staging slots per (image, brick)
    frog::DevBuf<float4> extract_tmp;         // one image's lattice, contiguous (frog_get_grid / frog_get_gradient)
    frog::DevBuf<uint32_t> subset_idx;        // frog_get_points2_subset scratch
    frog::DevBuf<float> subset_out;
    std::vector<float4> h_res_sums, h_res_pos; // frog_residual_sums: owned rows, internal numbering
    bool res_valid = false;
    bool xyz2_fresh = false;                  // pos2_spec holds transformPoints(apply=0) of the current state
    float pending_alpha = 0;
    int phase = 0;                            // 0 idle, 1 after phase_a, 2 after phase_b

    // FROG_REFERENCE_ORDER=1 (test hook, k_reforder.hip.h): every solver loop in the reference's own order and arithmetic --
    // no culling list, no fast weight, no re-associated sum; results are bit-comparable with the tests' CPU restatement of the reference
    bool ref_order = false;
    double create_s[3] = { 0, 0, 0 };   // frog_create: host layout build, allocations + uploads + first kernels, reservoir selections replayed ahead
    int create_selections = 0;
    bool created = false;               // frog_create has returned: allocations of lattice buffers from here on are counted
    int lattice_reallocs = 0;
    bool two_collectives = false;   // frog_comm_mode: the energy sums ride on the all-reduce of the proposal sums, the oversize count on the coordinate gather
    bool finish_deformable = false; // ... and they decide a deformable step (frog_step_finish commits or rejects it)
    bool speculated = false;        // frog_step_speculate has exchanged the lattices' roles ahead of the decision
    bool spec_coeff_zero = false;
    bool scalars_by_copy = false;   // the scalars of the pending step come by hipMemcpyAsync + energy_copied (no device-visible pinned block)
    double pending_seq = 0.0;       // sequence number of the scalars frog_comm_unpack_slab_step published and frog_step_finish has not read yet
    bool k11_f64 = false;           // FROG_K11_F64=1: the B-spline transform's weights and sums in f64 (rounds 1-4), for comparison
    frog::DevBuf<uint32_t> ref_own;           // [L_own] own point (internal numbering) of every half-link, reference order
    frog::DevBuf<float> ref_w, ref_d;         // [L_own] weight and distance of every half-link (linear step)
    frog::DevBuf<uint64_t> ref_img_link;      // [nOwned + 1] first half-link of every owned image (relative, reference order)
    frog::DevBuf<double> ref_pt_energy;       // [ownP][2] energy terms of every owned point (deformable step)
    // reference-order scatter as one chain per (image, control point) (k_refchain.hip.h), built once per lattice
    frog::DevBuf<uint16_t> ref_link_img;      // [L_own] image of every half-link's partner (static; ref_own is static too since round 6)
    // ... the rows of half-links side by side for the deformable per-point sums (k_refchain.hip.h), built once per context
    bool rr_valid = false;
    uint32_t rr_n_groups = 0;
    frog::DevBuf<uint32_t> rr_slot_row, rr_group_len, rr_ent;
    frog::DevBuf<uint16_t> rr_ent_img;
    frog::DevBuf<uint64_t> rr_group_ptr;
    bool rc_valid = false;                    // the chains below belong to the current lattice and the current `pos`
    bool ref_literal = false;                   // FROG_REF_LITERAL=1: the literal form (ref_scatter_kernel), for comparison
    uint32_t rc_n_groups = 0, rc_n_gnodes = 0;
    hipStream_t ref_stream = nullptr;         // the energy chains of a deformable step run here, beside the scatter
    hipEvent_t ref_fork = nullptr, ref_join = nullptr;
    bool ref_join_pending = false;
    bool rc_by_row = false;                   // the chains' entries name owned rows (ref_row_sums) instead of points (point_sums)
    frog::DevBuf<float4> ref_row_sums;        // [ownP] the per-point sums once more, by owned row in reference order
    int rc_unroll = 8;                        // entries per step of the chain kernel on this lattice (8 or 16)
    frog::DevBuf<uint64_t> rc_keys, rc_keys_alt;      // [64 ownP] sort buffers
    frog::DevBuf<unsigned char> rc_temp;              // hipCUB scratch
    frog::DevBuf<uint32_t> rc_node_ptr;               // [gnodes + 1] first sorted entry of every (image, control point)
    frog::DevBuf<uint32_t> rc_len, rc_len_sorted, rc_iota, rc_slot_node, rc_slot_of_node;   // chain lengths; slot <-> control point
    frog::DevBuf<uint32_t> rc_group_len;              // [groups] padded chain length of the group
    frog::DevBuf<uint64_t> rc_group_size, rc_group_ptr;   // [groups + 1] seats of the group, their exclusive sum
    frog::DevBuf<uint32_t> rc_ent;                    // [seats] point (internal numbering) or RC_PAD
    frog::DevBuf<double> rc_wt;                       // [seats] the tap's f64 weight

    // certified outlier culling of the deformable sweep (k_cull.hip.h)
    bool exact_weights = false;               // FROG_WEIGHT_EXACT=1 (test hook): inlier_probability_exact for every weight
    bool general_weights = false;             // FROG_WEIGHT_GENERAL=1 (test hook): no image gets a range for the one-exponential form
                                              // (k_stats.hip.h em_fast_of with theta = NaN), the deformable sweeps evaluate inlier_probability twice
    float fast_theta() const { return general_weights ? __builtin_nanf("") : opt.inlier_threshold - 1e-4f; }
    bool cull_enabled = true;                 // FROG_CULL=0 turns it off (every sweep walks all records)
    bool cull_need_build = true;              // host side: (re)build the list before the next deformable sweep
    float cull_scale = 2.0f, cull_pad = 25.0f; // list cutoff = scale * certified cutoff + pad (the skin)
    // the same machinery for the LINEAR stage (weights that are exactly zero, k_cull.hip.h cull_cutoff_linear_of): on unless
    // FROG_CULL_LINEAR=0; its own skin (the cutoffs are 14.5 c1: far out, where a tighter skin still lasts for iterations)
    bool cull_linear = true;
    float cull_lin_scale = 1.25f, cull_lin_pad = 10.0f;
    uint64_t cull_lin_builds = 0;             // statistics: lists built during the linear stage
    frog::DevBuf<unsigned long long> lin_listed;   // [0] half-links in the last list of the linear stage
    frog::DevBuf<uint32_t> act_recs32;        // listed records (narrow form), or ...
    frog::DevBuf<frog::LinkRec> act_recs;     // ... wide form; same offsets as recs32 / recs
    frog::DevBuf<uint32_t> act_cnt;           // [n_tiles][n_groups]
    bool pos_b_stale = true;                  // pos was written (`apply`) since pos_b was gathered
    frog::DevBuf<float4> pos_b;               // pos in the order of perm (brick, cell, index): written by every lattice set-up,
                                              // read (coalesced) by the scatter and the B-spline transforms of that lattice --
                                              // pos itself does not change while a lattice stands (only `apply` writes it)
    frog::DevBuf<frog::P3> pos2_snap;         // xyz2 of every point when the list was built
    frog::DevBuf<float> cut_now, cut_list;    // [nI] certified cutoff of the current mixtures / list cutoff at build time
    frog::DevBuf<uint32_t> disp_part;         // per-block maxima of the points' displacement since the build (f32 bits)
    uint32_t disp_n = 0;                      // entries of disp_part the last producer wrote
    uint32_t disp_own_n = 0;                  // ... of them by the transform of the OWNED points (the rest: frog_comm_unpack_slab)
    bool disp_others = false;                 // disp_part also covers the other ranks' rows (measured while they were unpacked)
    frog::DevBuf<uint32_t> cull_state;        // [0] 1: list not valid for the current coordinates
    frog::DevBuf<float> disp_allow;           // [0] displacement up to which the list stays good (cull_allow_kernel)
    frog::DevBuf<float4> retired_arena;       // finished lattices are copied here, one after the other (retire_current_grid)
    size_t retired_used = 0;
    bool coeff_zero = false;                  // host side: the current lattice has not taken a step yet (all coefficients 0)
    bool build_in_sweep = false;              // host side: the sweep of this step also writes the list (k_links.hip.h BUILD)
    bool cull_check_due = true;               // host side: cutoffs or list changed since the stand-alone check last ran
    uint64_t cull_builds = 0;                 // statistics: lists built
    bool disp_current = false;                // disp_part holds the displacement of the CURRENT xyz2 from the snapshot
    bool disp_spec = false;                   // ... of pos2_spec (becomes current when it is published)

    // live timing
    bool point_sums_stale = false;             // the last deformable step left the per-point sums as N_XCD partial sums
    int profiling = 0;                         // 0 off, 1 every kernel group, 2 the half-link sweeps only (3 = 2 with every fourth steady launch timed)
    int profile_stride = 1;                    // mode 2: every launch of a steady sweep carries events; mode 3: one in four
    uint64_t span_seen[FROG_K_COUNT_] = {};    // launches of the group since frog_profile_read(reset), timed or not
    double timed_ms[FROG_K_COUNT_] = {};       // ... the timed ones: their sum and count
    uint64_t timed_n[FROG_K_COUNT_] = {};
    struct TimedSpan { hipEvent_t a, b; int slot; };
    std::vector<TimedSpan> spans;             // recorded, not yet read
    std::vector<std::pair<hipEvent_t, hipEvent_t>> free_events;
    frog_kernel_time ktime[FROG_K_COUNT_] = {};

    uint32_t n_owned() const { return ie - ib; }
    // -fi: the first nf images are fixed.  This context then owns [nf, nI) and `helper` (stats only,
    // same stream) owns [0, nf): updateStats refreshes every image's mixture (imageGroup.cxx:569-598
    // loops from 0), everything else loops from numberOfFixedImages.
    uint32_t nf = 0;
    frog_ctx *helper = nullptr;
    bool whole_group() const { return ib == nf && ie == nI; }
};
