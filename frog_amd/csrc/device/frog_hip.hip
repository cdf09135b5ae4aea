// frog_hip.hip -- C ABI of libfrog_hip.so (include/frog_hip.h): context life cycle,
// launch sequences of the kernels in k_*.hip.h, read-back.  gfx950 only; there is
// no CPU path in this library.

#include "ctx.h"
#include <hip/hip_ext.h>
#include "prep.h"
#include "k_links.hip.h"
#include "k_stats.hip.h"
#include "k_grid.hip.h"
#include "k_cull.hip.h"
#include "k_ransac.hip.h"
#include "k_reforder.hip.h"
#include "k_refchain.hip.h"
#include "similarity.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <random>
#include <limits>
#include <new>

namespace frog {
thread_local std::string g_last_error;
constexpr int BOUNDS_BLOCKS = 256;
void set_last_error(const std::string &s) { g_last_error = s; }     // for the library's other translation units
}

using namespace frog;

#define CTX_GUARD(ctx)                                                   \
    do {                                                                 \
        if (!(ctx)) return fail(FROG_E_INVALID, "null context");         \
        FROG_HIP_CHECK(hipSetDevice((ctx)->device));                      \
    } while (0)

// This library's host loops (prep.h, the RANSAC batches) run on the LLVM OpenMP runtime.  At its first use that runtime maps the
// machine's topology for its affinity interface -- 40-120 ms on a 256-CPU host, measured as "numbering" taking 0.04-0.12 s instead
// of 0.005 -- which nothing here uses (no thread is ever bound).  Switched off when the library is loaded, before the runtime's
// first call, unless the user has said something about KMP_AFFINITY themselves.
// And its workers spin for 200 ms after a parallel region before they go to sleep (KMP_BLOCKTIME): the regions here are set-up
// work followed at once by a timed device loop whose host side (the caller's thread, the matcher's collectors) then shares a
// container's CPU quota with fifteen spinning threads -- the matcher ran 5 % slower after its create had become a parallel region
// (10 100-10 500 against 11 100-11 700 image pairs/s).  Workers sleep at once, again unless the user said otherwise.
__attribute__((constructor)) static void frog_openmp_defaults()
{
    setenv("KMP_AFFINITY", "disabled", 0);
    setenv("KMP_BLOCKTIME", "0", 0);
}

static inline unsigned div_up(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// FROG_ROCTX=1: a roctx range around every kernel group (the names of FROG_K_*), for `rocprofv3 --marker-trace`.  The
// library is looked up at run time (librocprofiler-sdk-roctx.so, else libroctx64.so: both export roctxRangePushA / roctxRangePop),
// so that nothing links against a profiler.
#include <dlfcn.h>
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *e = getenv("FROG_ROCTX");
        if (!e || !atoi(e)) return;
        for (const char *name : { "librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4" }) {
            if (void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
const Roctx &roctx() { static Roctx r; return r; }
const char *const K_GROUP_NAMES[FROG_K_COUNT_] = { "frog:sweep_linear", "frog:sweep_deformable", "frog:scatter", "frog:lattice_step", "frog:transform_points",
                                                    "frog:update_stats", "frog:combine_energy", "frog:cull_check", "frog:sweep_deformable_build_list",
                                                    "frog:sweep_linear_build_list" };
}

// RAII bracket: records a HIP event pair around the launches issued in its scope (frog_profile_enable) and, with FROG_ROCTX=1,
// a roctx range around their enqueueing.
struct Span {
    frog_ctx *c; int slot; hipEvent_t a = nullptr, b = nullptr;
    bool attached;          // the events ride on ONE kernel's own dispatch packet (hipExtLaunchKernelGGL) instead of being
                            // recorded around it: no marker packets, so no bubbles before and after the kernel
    bool range = false;
    Span(frog_ctx *ctx, int s, bool attach = false) : c(ctx), slot(s), attached(attach)
    {
        if (roctx().push) { roctx().push(K_GROUP_NAMES[s]); range = true; }
        if (!c->profiling || (c->profiling == 2 && s > FROG_K_SWEEP_DEFORMABLE && s != FROG_K_SWEEP_BUILD && s != FROG_K_SWEEP_LINEAR_BUILD)) return;
        // sampled timing (frog_profile_enable(ctx, 3)): every launch is counted, one steady sweep in profile_stride carries events
        // -- a launch with events on its dispatch starts ~6 us late and holds its successor back ~5.
        // The list-writing launches, a few per run and three times as long, are all timed.
        const uint64_t seen = c->span_seen[s]++;
        if (c->profiling == 2 && c->profile_stride > 1 && (s == FROG_K_SWEEP_DEFORMABLE || s == FROG_K_SWEEP_LINEAR) && seen % (uint64_t)c->profile_stride != 0) return;
        if (!c->free_events.empty()) { a = c->free_events.back().first; b = c->free_events.back().second; c->free_events.pop_back(); }
        else if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        if (!attached) (void)hipEventRecord(a, c->stream);
    }
    ~Span()
    {
        if (a) {
            if (!attached) (void)hipEventRecord(b, c->stream);
            c->spans.push_back({ a, b, slot });
        }
        if (range) roctx().pop();
    }
};

template <class T> static int download_points(frog_ctx *ctx, const T *src, float *dst)
{
    std::vector<T> h(ctx->P);
    FROG_HIP_CHECK(hipMemcpyAsync(h.data(), src, h.size() * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (size_t p = 0; p < ctx->P; p++) {          // p: reference numbering
        const T &v = h[ctx->h_new_of_old[p]];
        dst[3 * p] = v.x; dst[3 * p + 1] = v.y; dst[3 * p + 2] = v.z;
    }
    return FROG_OK;
}

static unsigned sweep_blocks(const frog_ctx *ctx) { return div_up(ctx->n_tiles, 4) * N_XCD; }     // per sub-pass

// Entries of the sweep's LDS tables (EM constants and first point of the partner group's images; dynamic LDS, 20 bytes
// each): narrow records index them with an img_bits-wide field -- also the prefetched records past a range, which may
// hold any value -- so 2^img_bits; wide records with the largest group's size; 0 when a group is too large for LDS.
static_assert(THRESHOLD_BAND == 1e-4f, "ctx.h frog_ctx::fast_theta spells the band out");

static uint32_t sweep_lds_images(const frog_ctx *ctx)
{
    if (ctx->rec_format.narrow) return 1u << ctx->rec_format.img_bits;
    uint32_t widest = 0;
    for (uint32_t g = 0; g < ctx->n_groups; g++) widest = std::max(widest, ctx->group_begin[g + 1] - ctx->group_begin[g]);
    return widest <= (uint32_t)EMD_LDS_IMAGES ? widest : 0u;
}

static SweepArgs sweep_args(frog_ctx *ctx, uint32_t sub)
{
    SweepArgs a;
    a.sub = sub;
    a.n_groups = ctx->n_groups;
    for (uint32_t g = 0; g <= ctx->n_groups; g++) a.group_begin[g] = ctx->group_begin[g];
    a.tiles = ctx->tiles.p; a.recs = ctx->rec_format.narrow ? (const void *)ctx->recs32.p : (const void *)ctx->recs.p; a.pos2 = ctx->pos2.p; a.emd = ctx->emd.p; a.emf = ctx->emf.p; a.em = ctx->em.p;
    a.rec2_last = (uint32_t)(ctx->L_recs / 2 - 1);
    a.img_bits = ctx->rec_format.img_bits; a.lds_images = sweep_lds_images(ctx); a.poff = ctx->d_poff.p; a.point_last = (uint32_t)(ctx->P ? ctx->P - 1 : 0);
    a.n_tiles = ctx->n_tiles; a.threshold = ctx->opt.inlier_threshold;
    a.band = ctx->exact_weights ? __builtin_inff() : THRESHOLD_BAND;
    a.tile_partial = ctx->tile_partial.p; a.tile_counts = ctx->tile_counts.p; a.group_sums = ctx->group_sums.p;
    a.own_pt_begin = ctx->own_pt_begin; a.own_points = ctx->own_pt_end - ctx->own_pt_begin;
    a.act_recs = nullptr; a.act_cnt = nullptr; a.cull_state = nullptr;
    a.cut_list = nullptr; a.build_recs = nullptr; a.build_cnt = nullptr;
    a.point_sums = ctx->point_sums.p; a.tile_order = ctx->tile_order.p; a.tiles_bo = ctx->tiles_bo.p;
    return a;
}

// culling applies to the deformable sweeps of a context whose threshold leaves room for the certified margin
static bool cull_active(const frog_ctx *ctx)
{
    return ctx->cull_enabled && ctx->deformable && ctx->opt.inlier_threshold >= 1e-3f && ctx->n_tiles > 0;
}

// ... and to the linear sweeps (weights that are exactly zero: k_cull.hip.h cull_cutoff_linear_of), whose zero set belongs
// to the fast weight: not with FROG_WEIGHT_EXACT
static bool cull_active_linear(const frog_ctx *ctx)
{
    return ctx->cull_enabled && ctx->cull_linear && !ctx->exact_weights && !ctx->deformable && ctx->n_tiles > 0 && ctx->n_sub == 1;
}

// the narrow deformable sweep can write the culling list while it walks every record (k_links.hip.h BUILD)
static bool sweep_builds_list(const frog_ctx *ctx) { return ctx->rec_format.narrow != 0; }

// Does this deformable sweep run in its fused form (k_links.hip.h FUSED)?  The one launch per list that walks EVERY record
// (and writes the list) does not: its false matches gather from uniformly random places of their partner image, which the
// per-group form keeps inside one XCD's 3 MB slice of the coordinates and the fused form does not (measured on cfg 3:
// 0.62 against 0.76 ms; the steady-state launches 0.300 against 0.254 ms).  FROG_SWEEP_FUSED=2 fuses that launch too.
static bool sweep_fused_now(const frog_ctx *ctx, bool build)
{
    static const bool fuse_build = [] { const char *e = getenv("FROG_SWEEP_FUSED"); return e && e[0] == '2'; }();
    return ctx->fused_sweep && (!build || fuse_build);
}

template <int MODE>
static void launch_sweep(frog_ctx *ctx, uint32_t sub, hipStream_t s, hipEvent_t ea = nullptr, hipEvent_t eb = nullptr,
                         bool use_list = false, bool build_list = false)
{
    uint32_t widest = 0;
    for (uint32_t g = 0; g < ctx->n_groups; g++) widest = std::max(widest, ctx->group_begin[g + 1] - ctx->group_begin[g]);
    // a context without tiles (every owned image is empty): nothing to launch -- a grid of 0 blocks is an invalid
    // configuration; the reductions that follow read no tile partials and write zero sums
    if (ctx->n_tiles == 0) return;
    SweepArgs args = sweep_args(ctx, sub);
    if (use_list) {         // the caller has run cull_check() for the coordinates this launch reads
        args.act_recs = ctx->rec_format.narrow ? (const void *)ctx->act_recs32.p : (const void *)ctx->act_recs.p;
        args.act_cnt = ctx->act_cnt.p;
        args.cull_state = ctx->cull_state.p;
    }
    const dim3 grid(sweep_blocks(ctx)), block(256);
    const size_t lds = (size_t)args.lds_images * (sizeof(EmDerived) + sizeof(uint32_t) + sizeof(float));
    if constexpr (MODE == SWEEP_LINEAR) {
        if (build_list && sweep_builds_list(ctx)) {
            args.act_recs = nullptr; args.act_cnt = nullptr;         // walks every record, writes the list
            args.cut_list = ctx->cut_list.p; args.build_recs = ctx->act_recs32.p; args.build_cnt = ctx->act_cnt.p;
            hipExtLaunchKernelGGL((sweep_kernel<SWEEP_LINEAR, true, false, true>), grid, block, lds, s, ea, eb, 0, args);
            return;
        }
    }
    if constexpr (MODE == SWEEP_DEFORMABLE) {
        const bool build = build_list && sweep_builds_list(ctx);
        if (build) {
            args.act_recs = nullptr; args.act_cnt = nullptr;         // walks every record
            args.cut_list = ctx->cut_list.p; args.build_recs = ctx->act_recs32.p; args.build_cnt = ctx->act_cnt.p;
        }
        if (sweep_fused_now(ctx, build)) {
            // one block per tile, wavefront = partner group, ONE float4 of sums per point out (k_links.hip.h FUSED)
            const dim3 fgrid(ctx->n_order_blocks), fblock(512);
            const size_t flds = lds * N_XCD;
            if (build) hipExtLaunchKernelGGL((sweep_kernel<SWEEP_DEFORMABLE, true, false, true, true>), fgrid, fblock, flds, s, ea, eb, 0, args);
            else if (ctx->rec_format.narrow) hipExtLaunchKernelGGL((sweep_kernel<SWEEP_DEFORMABLE, true, false, false, true>), fgrid, fblock, flds, s, ea, eb, 0, args);
            else hipExtLaunchKernelGGL((sweep_kernel<SWEEP_DEFORMABLE, true, true, false, true>), fgrid, fblock, flds, s, ea, eb, 0, args);
            return;
        }
        if (build) {
            hipExtLaunchKernelGGL((sweep_kernel<SWEEP_DEFORMABLE, true, false, true>), grid, block, lds, s, ea, eb, 0, args);
            return;
        }
    }
    if (ctx->rec_format.narrow)                 // implies a group of at most 2^img_bits <= EMD_LDS_IMAGES images
        hipExtLaunchKernelGGL((sweep_kernel<MODE, true, false>), grid, block, lds, s, ea, eb, 0, args);
    else if (widest <= (uint32_t)EMD_LDS_IMAGES)
        hipExtLaunchKernelGGL((sweep_kernel<MODE, true, true>), grid, block, lds, s, ea, eb, 0, args);
    else
        hipExtLaunchKernelGGL((sweep_kernel<MODE, false, true>), grid, block, 0, s, ea, eb, 0, args);
}

static int cull_allocate(frog_ctx *ctx);
static int cull_prepare(frog_ctx *ctx);
static int make_geometry(const frog_ctx *ctx, int level, const double mins[3], const double maxs[3], GridGeom &g, frog_grid_info &info);
static int lattice_alloc(frog_ctx *ctx, const GridGeom &g);

// The runtime gives a queue its scratch memory at the first dispatch that asks for any -- a stall of 0.13 ms in front of the
// first tiled B-spline transform of a run (the one kernel here with spills: 12 bytes per lane; rocprofv3 kernel trace).
// This kernel asks for more than that, once, on every stream a context is put on.
__global__ void scratch_warm_kernel(unsigned int *out, int n)
{
    volatile unsigned int a[16];
    for (int k = 0; k < 16; k++) a[k] = (unsigned int)(k * n);
    unsigned int v = 0;
    for (int k = 0; k < 16; k++) v += a[(k + n) & 15];
    if (n < 0) out[0] = v;              // never: n >= 0
}

// upper bound of the scatter's block count: every non-empty brick ends with at most one partial block
static uint32_t scatter_max_blocks(uint32_t n_bricks_total, uint32_t n_points, uint32_t chunk)
{
    return std::min(n_bricks_total, n_points) + n_points / chunk;
}

// Device work of a lattice set-up (frog_deformable_setup_bounds) for the geometry in ctx->geom, on stream s.
static int queue_setup_kernels(frog_ctx *ctx, hipStream_t s)
{
    const GridGeom &g = ctx->geom;
    const uint32_t nO = ctx->n_owned();
    const uint32_t nPts = ctx->own_pt_end - ctx->own_pt_begin;
    const size_t nb = (size_t)g.n_bricks;
    int rc = FROG_OK;
    {
        // gradient lattice, coefficients, proposals, their sums, the sort's counters, the block-length histogram: one launch
        ZeroList z{};
        auto add = [&](void *p, size_t bytes) {
            if (!p || !bytes) return;
            z.p[z.n] = (uint32_t *)p; z.bytes[z.n] = bytes;
            z.first_block[z.n + 1] = z.first_block[z.n] + (unsigned)((bytes + ZERO_BLOCK_BYTES - 1) / ZERO_BLOCK_BYTES);
            z.n++;
        };
        add(ctx->gradf.p, ctx->gradf.bytes()); add(ctx->coeff.p, ctx->coeff.bytes()); add(ctx->grad.p, ctx->grad.bytes());
        add(ctx->gridsum.p, ctx->gridsum.bytes()); add(ctx->key_counts.p, ctx->key_counts.bytes());
        add(ctx->len_hist.p, ctx->len_hist.bytes());
        if (g.sparse) {
            add(ctx->lat_mask.p, ctx->lat_mask.bytes()); add(ctx->ucoeff.p, ctx->ucoeff.bytes()); add(ctx->ugrad.p, ctx->ugrad.bytes());
            add(ctx->ugrad_spare.p, ctx->ugrad_spare.bytes());
        }
        if (z.n) {
            zero_buffers_kernel<<<z.first_block[z.n], 256, 0, s>>>(z);
            FROG_HIP_CHECK(hipGetLastError());
        }
    }

    // sort the owned points by (image, brick, cell) and build the scatter's block table -- all on the device, nothing
    // comes back to the host (the first version read the brick sizes back, built and sorted the table on the host and
    // uploaded it: 1.0-2.5 ms per lattice, most of it two round trips and the host loop)
    const uint32_t keys_per_brick = (uint32_t)(g.brick * g.brick * g.brick);
    const uint32_t n_keys = (uint32_t)((size_t)nO * nb * keys_per_brick);
    const uint32_t n_bricks_total = nO * (uint32_t)nb;
    frog::DevBuf<uint32_t> &counts = ctx->key_counts, &chunks = ctx->brick_ptr_scratch;
    const GeomDev gd = to_dev(g);
    uint32_t max_img_pts = 0;
    for (uint32_t i = ctx->ib; i < ctx->ie; i++) max_img_pts = std::max(max_img_pts, ctx->poff[i + 1] - ctx->poff[i]);
    const dim3 bgrid(std::max(1u, div_up(max_img_pts, BRICK_BLOCK_POINTS)), nO);
    if (nPts) {
        brick_count_kernel<<<bgrid, 256, 0, s>>>(ctx->pos.p, ctx->d_poff.p, ctx->ib, gd, counts.p);
        FROG_HIP_CHECK(hipGetLastError());
    }
    auto exclusive_scan = [&](const uint32_t *in, uint32_t n, uint32_t *ptr, uint32_t *cursor) -> int {
        const uint32_t n_scan_blocks = div_up(n, SCAN_BLOCK_ITEMS);
        frog::DevBuf<uint32_t> &bsums = ctx->scan_sums;      // sized for the longer of the two scans by lattice_alloc
        scan_block_sums_kernel<<<n_scan_blocks, 1024, 0, s>>>(in, n, bsums.p);
        scan_of_sums_kernel<<<1, 1024, 0, s>>>(bsums.p, n_scan_blocks, bsums.p + n_scan_blocks);
        scan_apply_kernel<<<n_scan_blocks, 1024, 0, s>>>(in, n, bsums.p, bsums.p + n_scan_blocks, ptr, cursor);
        FROG_HIP_CHECK(hipGetLastError());
        return FROG_OK;
    };
    rc = exclusive_scan(counts.p, n_keys, ctx->key_ptr.p, ctx->key_cursor.p);
    if (rc) return rc;
    if (nPts) {
        brick_place_kernel<<<bgrid, 256, 0, s>>>(ctx->pos.p, ctx->d_poff.p, ctx->ib, gd, ctx->key_cursor.p, ctx->perm.p, ctx->perm_key.p);
        FROG_HIP_CHECK(hipGetLastError());
        // canonical order inside every cell (the placement's atomics make it arbitrary): reproducible sums
        cell_order_kernel<<<div_up(nPts, 256), 256, 0, s>>>(ctx->key_ptr.p, ctx->perm_key.p, nPts, ctx->perm.p, ctx->perm_tmp.p);
        FROG_HIP_CHECK(hipGetLastError());
        std::swap(ctx->perm.p, ctx->perm_tmp.p);
        std::swap(ctx->perm.cap, ctx->perm_tmp.cap);
        std::swap(ctx->perm.n, ctx->perm_tmp.n);
        gather_positions_kernel<<<div_up(nPts, 256), 256, 0, s>>>(ctx->pos.p, ctx->perm.p, nPts, ctx->pos_b.p);
        FROG_HIP_CHECK(hipGetLastError());
        ctx->pos_b_stale = false;
    }
    // block table (k_grid.hip.h): blocks per brick -> staging slots (scan) -> blocks in brick order -> longest first.
    // Its length stays on the device (brick_slot_ptr[n_bricks_total]); the scatter is launched with an upper bound:
    // every non-empty brick ends with at most one partial block
    const uint32_t max_blocks = scatter_max_blocks(n_bricks_total, nPts, ctx->scatter_chunk);
    brick_chunks_kernel<<<div_up(n_bricks_total, 256), 256, 0, s>>>(ctx->key_ptr.p, n_bricks_total, keys_per_brick, ctx->scatter_chunk, chunks.p);
    FROG_HIP_CHECK(hipGetLastError());
    rc = exclusive_scan(chunks.p, n_bricks_total, ctx->brick_slot_ptr.p, ctx->key_cursor.p /* scratch: the placement is done */);
    if (rc) return rc;
    ScatterBlock *blk_tmp = reinterpret_cast<ScatterBlock *>(ctx->scatter_blocks_tmp.p);
    ScatterBlock *blk = reinterpret_cast<ScatterBlock *>(ctx->scatter_blocks.p);
    uint32_t *len_hist = ctx->len_hist.p, *len_cursor = ctx->len_hist.p + (SCATTER_CHUNK + 1);
    const uint32_t *n_blocks_dev = ctx->brick_slot_ptr.p + n_bricks_total;
    block_fill_kernel<<<div_up(n_bricks_total, 256), 256, 0, s>>>(ctx->key_ptr.p, ctx->brick_slot_ptr.p, n_bricks_total, keys_per_brick,
                                                                 ctx->scatter_chunk, blk_tmp, len_hist);
    static_assert(SCATTER_CHUNK + 1 <= 1024, "block_len_base_kernel: one thread per block length");
    block_len_base_kernel<<<1, (SCATTER_CHUNK + 1 + 63) / 64 * 64, 0, s>>>(len_hist, len_cursor);
    if (max_blocks)
        block_sort_kernel<<<div_up(max_blocks, 256), 256, 0, s>>>(blk_tmp, n_blocks_dev, ctx->scatter_chunk, len_cursor, blk);
    if (g.sparse) {                             // the active (image, node) pairs of the new lattice, block by block; per node how many images lack it
        if (nPts && max_blocks) lattice_mask_kernel<<<max_blocks, 64, 0, s>>>(ctx->pos_b.p, blk, n_blocks_dev, gd, ctx->lat_mask.p);
        lattice_inactive_kernel<<<div_up((size_t)gd.n_cp, 256), 256, 0, s>>>(ctx->lat_mask.p, nO, gd, ctx->lat_inactive.p);
    }
    FROG_HIP_CHECK(hipGetLastError());
    return FROG_OK;
}

// Makes `stream` wait for the lattice set-up (once per set-up; a no-op otherwise).  A set-up whose device work has not
// been queued yet (frog_deformable_setup_bounds) is queued now, on the set-up stream, behind the fork event.
static int join_setup(frog_ctx *ctx)
{
    if (ctx->setup_deferred) {
        ctx->setup_deferred = false;
        FROG_HIP_CHECK(hipStreamWaitEvent(ctx->setup_stream, ctx->setup_fork, 0));
        const int rc = queue_setup_kernels(ctx, ctx->setup_stream);
        if (rc) return rc;
        FROG_HIP_CHECK(hipEventRecord(ctx->setup_join, ctx->setup_stream));
        ctx->setup_pending = true;
    }
    if (!ctx->setup_pending) return FROG_OK;
    FROG_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->setup_join, 0));
    ctx->setup_pending = false;
    return FROG_OK;
}

// join_setup + the positions in perm's order brought up to date (an `apply` that no set-up followed: the C ABI allows it)
static int join_setup_positions(frog_ctx *ctx)
{
    const int rc = join_setup(ctx);
    if (rc) return rc;
    const uint32_t nPts = ctx->own_pt_end - ctx->own_pt_begin;
    if (ctx->pos_b_stale && nPts && ctx->pos_b.p && ctx->perm.p) {
        gather_positions_kernel<<<div_up(nPts, 256), 256, 0, ctx->stream>>>(ctx->pos.p, ctx->perm.p, nPts, ctx->pos_b.p);
        FROG_HIP_CHECK(hipGetLastError());
        ctx->pos_b_stale = false;
    }
    return FROG_OK;
}

// Spins until the kernel that carries a step's scalars (publish_step_scalars) has written sequence number `seq` to pinned memory.
static int wait_step_scalars(frog_ctx *ctx, double seq)
{
    volatile double *h = ctx->h_energy;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (h[7] != seq) {
        __builtin_ia32_pause();
        if ((++spins & 0xFFFFu) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
            // the kernel never ran (a launch failure shows here): wait for the stream and report
            FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (h[7] != seq) return fail(FROG_E_HIP, "the step's scalars never arrived");
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return FROG_OK;
}

// Queues, on the side stream, the selection of the next refresh not yet selected (k_stats.hip.h select_kernel: the
// generator's state carries over, so selections are produced in refresh order) and the end points of its half-links.
static int produce_selection(frog_ctx *c)
{
    const int b = (int)(c->sel_produced % (uint64_t)c->sel_ring);
    const uint32_t nO = c->n_owned(), cap = (uint32_t)c->sample_cap;
    if (c->sel_produced >= (uint64_t)c->sel_ring)                       // the buffer has been consumed before: wait for its readers
        FROG_HIP_CHECK(hipStreamWaitEvent(c->side, c->ord_read[b], 0));
    select_kernel<<<nO, SELECT_THREADS, 0, c->side>>>(c->mt_state.p, c->d_virtual.p, cap, c->sample_ord[b].p, c->sample_count[b].p);
    sample_resolve_kernel<<<dim3(div_up(cap, 256), nO), 256, 0, c->side>>>(
        c->sample_ord[b].p, c->sample_count[b].p, cap, c->d_poff.p, c->ib, c->own_pt_begin,
        c->ref_rowptr.p, c->ref_link.p, c->new_of_old.p, c->sample_ends[b].p);
    FROG_HIP_CHECK(hipGetLastError());
    FROG_HIP_CHECK(hipEventRecord(c->sel_done[b], c->side));
    c->sel_produced++;
    return FROG_OK;
}

// ---- FROG_REFERENCE_ORDER=1 (k_reforder.hip.h): launch sequences ---------------------------------------------
static int ref_rows_build(frog_ctx *ctx);
static int ref_alloc(frog_ctx *ctx)
{
    if (ctx->ref_img_link.p) return FROG_OK;
    const uint32_t nO = ctx->n_owned(), nRows = ctx->own_pt_end - ctx->own_pt_begin;
    std::vector<uint64_t> il(nO + 1);
    for (uint32_t i = 0; i <= nO; i++) il[i] = ctx->img_link_begin[ctx->ib + i];
    FROG_HIP_CHECK(ctx->ref_img_link.upload(il, ctx->stream));
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    FROG_HIP_CHECK(ctx->ref_own.alloc(std::max<size_t>(1, ctx->L_own)));
    FROG_HIP_CHECK(ctx->ref_link_img.alloc(std::max<size_t>(1, ctx->L_own)));
    FROG_HIP_CHECK(ctx->ref_pt_energy.alloc(2 * (size_t)std::max(1u, nRows)));
    FROG_HIP_CHECK(ctx->ref_row_sums.alloc(std::max(1u, nRows)));
    if (!ctx->ref_stream && !getenv("FROG_REF_ONE_STREAM")) {
        FROG_HIP_CHECK(hipStreamCreateWithFlags(&ctx->ref_stream, hipStreamNonBlocking));
        FROG_HIP_CHECK(hipEventCreateWithFlags(&ctx->ref_fork, hipEventDisableTiming));
        FROG_HIP_CHECK(hipEventCreateWithFlags(&ctx->ref_join, hipEventDisableTiming));
    }
    FROG_HIP_CHECK(ctx->ref_w.alloc(std::max<size_t>(1, ctx->L_own)));
    FROG_HIP_CHECK(ctx->ref_d.alloc(std::max<size_t>(1, ctx->L_own)));
    // per half-link, static: the own point and the partner's image (pos.w never changes)
    if (nRows)
        ref_link_static_kernel<<<div_up(nRows, 256), 256, 0, ctx->stream>>>(ctx->ref_rowptr.p, ctx->ref_link.p, ctx->new_of_old.p, nRows, ctx->pos.p,
                                                                          ctx->ref_own.p, ctx->ref_link_img.p);
    FROG_HIP_CHECK(hipGetLastError());
    // the other table that depends on the model alone: the rows of half-links side by side (the deformable stage's per-point sums)
    return ctx->ref_literal ? FROG_OK : ref_rows_build(ctx);
}

// The rows of half-links side by side (k_refchain.hip.h), once per context.
static int ref_rows_build(frog_ctx *ctx)
{
    if (ctx->rr_valid) return FROG_OK;
    hipStream_t s = ctx->stream;
    const uint32_t nRows = ctx->own_pt_end - ctx->own_pt_begin;
    const uint32_t n_groups = div_up(std::max(1u, nRows), 64u);
    const size_t n_slots = (size_t)n_groups * 64;
    DevBuf<uint32_t> len, len_sorted, iota;
    DevBuf<uint64_t> group_size;
    DevBuf<unsigned char> temp;
    FROG_HIP_CHECK(len.alloc(n_slots)); FROG_HIP_CHECK(len_sorted.alloc(n_slots)); FROG_HIP_CHECK(iota.alloc(n_slots));
    FROG_HIP_CHECK(ctx->rr_slot_row.alloc(n_slots)); FROG_HIP_CHECK(ctx->rr_group_len.alloc(n_groups));
    FROG_HIP_CHECK(group_size.alloc((size_t)n_groups + 1)); FROG_HIP_CHECK(ctx->rr_group_ptr.alloc((size_t)n_groups + 1));
    size_t t1 = 0, t2 = 0;
    FROG_HIP_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, t1, len.p, len_sorted.p, iota.p, ctx->rr_slot_row.p, n_slots, 0, 32, s));
    FROG_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, t2, group_size.p, ctx->rr_group_ptr.p, (size_t)n_groups + 1, s));
    FROG_HIP_CHECK(temp.alloc(std::max<size_t>(16, std::max(t1, t2))));
    FROG_HIP_CHECK(hipMemsetAsync(len.p, 0, n_slots * sizeof(uint32_t), s));
    FROG_HIP_CHECK(hipMemsetAsync(iota.p, 0xFF, n_slots * sizeof(uint32_t), s));          // slots past the last row: RC_PAD
    if (nRows) ref_rows_len_kernel<<<div_up(nRows, 256), 256, 0, s>>>(ctx->ref_rowptr.p, nRows, len.p, iota.p);
    size_t tb = temp.n;
    FROG_HIP_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(temp.p, tb, len.p, len_sorted.p, iota.p, ctx->rr_slot_row.p, n_slots, 0, 32, s));
    FROG_HIP_CHECK(hipMemsetAsync(group_size.p, 0, ((size_t)n_groups + 1) * sizeof(uint64_t), s));
    ref_rows_group_kernel<<<div_up(n_groups, 256), 256, 0, s>>>(len_sorted.p, n_groups, ctx->rr_group_len.p, group_size.p);
    tb = temp.n;
    FROG_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp.p, tb, group_size.p, ctx->rr_group_ptr.p, (size_t)n_groups + 1, s));
    uint64_t seats = 0;
    FROG_HIP_CHECK(hipMemcpyAsync(&seats, ctx->rr_group_ptr.p + n_groups, sizeof seats, hipMemcpyDeviceToHost, s));
    FROG_HIP_CHECK(hipStreamSynchronize(s));
    FROG_HIP_CHECK(ctx->rr_ent.alloc(std::max<uint64_t>(1, seats)));
    FROG_HIP_CHECK(ctx->rr_ent_img.alloc(std::max<uint64_t>(1, seats)));
    FROG_HIP_CHECK(hipMemsetAsync(ctx->rr_ent.p, 0xFF, std::max<uint64_t>(1, seats) * sizeof(uint32_t), s));
    FROG_HIP_CHECK(hipMemsetAsync(ctx->rr_ent_img.p, 0, std::max<uint64_t>(1, seats) * sizeof(uint16_t), s));
    ref_rows_fill_kernel<<<n_groups, 64, 0, s>>>(ctx->ref_rowptr.p, ctx->ref_link.p, ctx->pos.p, ctx->rr_slot_row.p, ctx->rr_group_ptr.p,
                                                ctx->rr_ent.p, ctx->rr_ent_img.p);
    FROG_HIP_CHECK(hipGetLastError());
    FROG_HIP_CHECK(hipStreamSynchronize(s));            // the scratch buffers above die with this scope
    ctx->rr_n_groups = n_groups;
    ctx->rr_valid = true;
    return FROG_OK;
}

// updateLinearTransforms (imageGroup.cxx:1063-1149) in the reference's order
static int ref_linear_step(frog_ctx *ctx)
{
    { const int rc = ref_alloc(ctx); if (rc) return rc; }
    hipStream_t s = ctx->stream;
    const uint32_t nO = ctx->n_owned(), nRows = ctx->own_pt_end - ctx->own_pt_begin;
    if (ctx->ref_literal) {
        if (nRows)
            ref_link_terms_kernel<<<div_up(nRows, 256), 256, 0, s>>>(ctx->ref_rowptr.p, ctx->ref_link.p, ctx->new_of_old.p, nRows, ctx->pos.p,
                                                                     ctx->pos2.p, ctx->em.p, ctx->ref_own.p, ctx->ref_w.p, ctx->ref_d.p);
        ref_linear_chain_kernel<<<nO, 64, 0, s>>>(ctx->ref_img_link.p, ctx->ref_link.p, ctx->ref_own.p, ctx->ref_w.p, ctx->ref_d.p, ctx->pos2.p,
                                                  ctx->ib, ctx->mat.p, ctx->opt.linear_alpha, ctx->opt.use_scale, ctx->img_energy.p);
    } else if (nO) {
        uint64_t longest = 0;
        for (uint32_t i = ctx->ib; i < ctx->ie; i++) longest = std::max(longest, ctx->img_link_begin[i + 1] - ctx->img_link_begin[i]);
        if (longest)
            ref_link_weights_kernel<<<dim3(div_up(longest, 64 * RW_UNROLL), nO), 64, 0, s>>>(ctx->ref_img_link.p, ctx->ref_link.p, ctx->ref_own.p,
                                                                                            ctx->ref_link_img.p, ctx->pos2.p, ctx->em.p, ctx->emd.p,
                                                                                            ctx->ib, ctx->ref_w.p, ctx->ref_d.p);
        ref_linear_chain2_kernel<<<nO, 64 * (RL_PRODUCERS + 1), 0, s>>>(ctx->ref_img_link.p, ctx->ref_link.p, ctx->ref_own.p, ctx->ref_w.p, ctx->ref_d.p,
                                                                       ctx->pos2.p, ctx->ib, ctx->mat.p, ctx->opt.linear_alpha,
                                                                       ctx->opt.use_scale, ctx->img_energy.p);
    }
    ref_energy_total_kernel<<<1, 1, 0, s>>>(ctx->img_energy.p, nO, ctx->energy.p);
    FROG_HIP_CHECK(hipGetLastError());
    return FROG_OK;
}

// per-point sums of the deformable step / the error maps (imageGroup.cxx:252-299, :493-533) in the reference's order
static int ref_point_sums(frog_ctx *ctx, bool with_energy)
{
    { const int rc = ref_alloc(ctx); if (rc) return rc; }
    hipStream_t s = ctx->stream;
    const uint32_t nO = ctx->n_owned(), nRows = ctx->own_pt_end - ctx->own_pt_begin;
    if (nRows && ctx->ref_literal) {
        ref_point_sums_kernel<<<div_up(nRows, 256), 256, 0, s>>>(ctx->ref_rowptr.p, ctx->ref_link.p, ctx->new_of_old.p, nRows, ctx->pos.p,
                                                                 ctx->pos2.p, ctx->em.p, ctx->opt.inlier_threshold, ctx->point_sums.p,
                                                                 with_energy ? ctx->ref_pt_energy.p : nullptr);
    } else if (nRows) {
        const int rc = ref_rows_build(ctx);
        if (rc) return rc;
        ref_point_sums_rows_kernel<<<ctx->rr_n_groups, 64, 0, s>>>(ctx->rr_ent.p, ctx->rr_ent_img.p, ctx->rr_group_ptr.p, ctx->rr_group_len.p,
                                                                  ctx->rr_slot_row.p, ctx->new_of_old.p, ctx->pos.p, ctx->pos2.p, ctx->em.p,
                                                                  ctx->emd.p, ctx->opt.inlier_threshold, ctx->point_sums.p, ctx->n_hard ? nullptr : ctx->ref_row_sums.p,
                                                                  with_energy ? ctx->ref_pt_energy.p : nullptr);
    }
    ctx->point_sums_stale = false;
    if (with_energy) {
        // The energy's two chains per image (20 000 dependent f64 additions each at cfg 3: 0.19 ms on 100 of 256 CUs) need nothing
        // the scatter produces and produce nothing it needs: on a stream of their own beside it, joined by ref_deformable_phase_a.
        hipStream_t se = s;
        if (!ctx->ref_literal && !ctx->n_hard && ctx->ref_stream) {
            FROG_HIP_CHECK(hipEventRecord(ctx->ref_fork, s));
            FROG_HIP_CHECK(hipStreamWaitEvent(ctx->ref_stream, ctx->ref_fork, 0));
            se = ctx->ref_stream;
        }
        if (ctx->ref_literal) ref_image_energy_kernel<<<nO, 64, 0, se>>>(ctx->ref_pt_energy.p, ctx->d_poff.p, ctx->ib, ctx->own_pt_begin, ctx->img_energy.p);
        else ref_image_energy2_kernel<<<nO, RE_STEP, 0, se>>>(ctx->ref_pt_energy.p, ctx->d_poff.p, ctx->ib, ctx->own_pt_begin, ctx->img_energy.p);
        ref_energy_total_kernel<<<1, 1, 0, se>>>(ctx->img_energy.p, nO, ctx->energy.p);
        if (se != s) { FROG_HIP_CHECK(hipEventRecord(ctx->ref_join, se)); ctx->ref_join_pending = true; }
    }
    if (ctx->n_hard) {                                          // landmark constraints, imageGroup.cxx:280-295, :520-533
        hard_links_kernel<<<div_up(ctx->n_hard, 64), 64, 0, s>>>(ctx->pos2.p, ctx->point_sums.p, ctx->hl_point.p, ctx->hl_ptr.p,
                                                                ctx->hl_partner.p, ctx->n_hard, ctx->hard_weight2,
                                                                with_energy ? ctx->hl_partial.p : nullptr);
        if (with_energy) hard_energy_kernel<<<1, 1, 0, s>>>(ctx->hl_partial.p, ctx->n_hard, ctx->energy.p);
    }
    FROG_HIP_CHECK(hipGetLastError());
    return FROG_OK;
}

// The chains of the reference-order scatter for the current lattice (k_refchain.hip.h): once per lattice, on `stream`.
// One workgroup per group of RC_GROUP chains.  A launch is one AQL packet whose grid is 32-bit WORK-ITEMS per dimension: cfg 5's
// finest lattice has 3.4e7 groups, x 256 threads = 8.8e9 -- the launch returned no error and the groups past 2^32 / 256 were
// never filled (round 6: found by scripts/diag_exact_forms.py at full size).  So: at most 2^22 workgroups in x, the rest in the
// second dimension; the kernels bound-check blockIdx.y * gridDim.x + blockIdx.x against the group count.
static inline dim3 rc_grid(uint32_t n_groups)
{
    static const uint32_t max_x = getenv("FROG_RC_GRID_X") ? (uint32_t)std::max(1, atoi(getenv("FROG_RC_GRID_X"))) : 1u << 22;      // (test hook: the fold on small groups)
    const uint32_t gx = std::min(std::max(1u, n_groups), max_x);
    return dim3(gx, div_up(std::max(1u, n_groups), gx));
}

static int ref_chain_build(frog_ctx *ctx)
{
    if (ctx->rc_valid) return FROG_OK;
    hipStream_t s = ctx->stream;
    static const bool trace = getenv("FROG_REF_TRACE") != nullptr;
    const auto t_in = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (trace) std::fprintf(stderr, "[ref_chain_build] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_in).count());
    };
    const GeomDev gd = to_dev(ctx->geom);
    const uint32_t nO = ctx->n_owned(), nRows = ctx->own_pt_end - ctx->own_pt_begin;
    const uint64_t n_gnodes64 = (uint64_t)nO * (uint64_t)gd.n_cp, n_keys = (uint64_t)nRows * 64;
    if (n_gnodes64 >= 0xFFFF0000ull || n_keys >= 0xFFFF0000ull) return fail(FROG_E_INVALID, "reference-order scatter: lattice or group too large for 32-bit chain indices");
    const uint32_t n_gnodes = (uint32_t)n_gnodes64;
    const uint32_t n_groups = div_up(std::max(1u, n_gnodes), (uint32_t)RC_GROUP);
    uint32_t max_img_pts = 1;
    for (uint32_t i = ctx->ib; i < ctx->ie; i++) max_img_pts = std::max(max_img_pts, ctx->poff[i + 1] - ctx->poff[i]);
    auto bits_for = [](uint64_t v) { int b = 1; while ((v >> b) != 0) b++; return b; };      // values 0 .. v
    const int rbits = bits_for(max_img_pts - 1), gbits = bits_for(n_gnodes);
    if (6 + rbits + gbits > 64) return fail(FROG_E_INVALID, "reference-order scatter: key does not fit 64 bits");
    ctx->rc_n_groups = n_groups; ctx->rc_n_gnodes = n_gnodes;
    FROG_HIP_CHECK(ctx->rc_keys.alloc(std::max<uint64_t>(1, n_keys)));
    FROG_HIP_CHECK(ctx->rc_keys_alt.alloc(std::max<uint64_t>(1, n_keys)));
    FROG_HIP_CHECK(ctx->rc_node_ptr.alloc((size_t)n_gnodes + 2));
    const size_t n_slots = (size_t)n_groups * RC_GROUP;
    FROG_HIP_CHECK(ctx->rc_len.alloc(n_slots)); FROG_HIP_CHECK(ctx->rc_len_sorted.alloc(n_slots));
    FROG_HIP_CHECK(ctx->rc_iota.alloc(n_slots)); FROG_HIP_CHECK(ctx->rc_slot_node.alloc(n_slots));
    FROG_HIP_CHECK(ctx->rc_slot_of_node.alloc(n_slots));
    FROG_HIP_CHECK(ctx->rc_group_len.alloc(n_groups));
    FROG_HIP_CHECK(ctx->rc_group_size.alloc((size_t)n_groups + 1)); FROG_HIP_CHECK(ctx->rc_group_ptr.alloc((size_t)n_groups + 1));
    size_t t1 = 0, t2 = 0, t3 = 0;
    FROG_HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(nullptr, t1, ctx->rc_keys.p, ctx->rc_keys_alt.p, (size_t)n_keys, 6, 6 + rbits + gbits, s));
    FROG_HIP_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(nullptr, t2, ctx->rc_len.p, ctx->rc_len_sorted.p, ctx->rc_iota.p, ctx->rc_slot_node.p, n_slots, 0, 32, s));
    FROG_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, t3, ctx->rc_group_size.p, ctx->rc_group_ptr.p, (size_t)n_groups + 1, s));
    FROG_HIP_CHECK(ctx->rc_temp.alloc(std::max<size_t>(16, std::max(t1, std::max(t2, t3)))));
    mark("allocations + size queries");
    size_t tb = ctx->rc_temp.n;
    if (nRows)
        ref_chain_keys_kernel<<<div_up(nRows, 4), 256, 0, s>>>(ctx->pos.p, ctx->new_of_old.p, ctx->d_poff.p, nRows, ctx->ib, ctx->own_pt_begin, gd,
                                                              n_gnodes, rbits, ctx->rc_keys.p);
    FROG_HIP_CHECK(hipGetLastError());
    if (n_keys) FROG_HIP_CHECK(hipcub::DeviceRadixSort::SortKeys(ctx->rc_temp.p, tb, ctx->rc_keys.p, ctx->rc_keys_alt.p, (size_t)n_keys, 6, 6 + rbits + gbits, s));
    const uint64_t *sorted = ctx->rc_keys_alt.p;
    ref_chain_bounds_kernel<<<(unsigned)((n_keys + 1 + 255) / 256), 256, 0, s>>>(sorted, n_keys, rbits, n_gnodes, ctx->rc_node_ptr.p);
    // slots past the last control point: length 0, pointing nowhere (the chain kernel does not write them)
    FROG_HIP_CHECK(hipMemsetAsync(ctx->rc_len.p, 0, n_slots * sizeof(uint32_t), s));
    FROG_HIP_CHECK(hipMemsetAsync(ctx->rc_iota.p, 0xFF, n_slots * sizeof(uint32_t), s));
    // long chains (coarse lattices: few control points, thousands of entries each) fetch 16 entries per step, others 8
    ctx->rc_unroll = n_keys / std::max<uint64_t>(1, n_gnodes) >= 256 ? 16 : 8;
    if (n_gnodes) ref_chain_len_kernel<<<div_up(n_gnodes, 256), 256, 0, s>>>(ctx->rc_node_ptr.p, n_gnodes, (uint32_t)ctx->rc_unroll, ctx->rc_len.p, ctx->rc_iota.p);
    tb = ctx->rc_temp.n;
    FROG_HIP_CHECK(hipcub::DeviceRadixSort::SortPairsDescending(ctx->rc_temp.p, tb, ctx->rc_len.p, ctx->rc_len_sorted.p, ctx->rc_iota.p, ctx->rc_slot_node.p, n_slots, 0, 32, s));
    FROG_HIP_CHECK(hipMemsetAsync(ctx->rc_group_size.p, 0, ((size_t)n_groups + 1) * sizeof(uint64_t), s));
    // ... and look the sums up by owned row (see ref_chain_fill_kernel); landmark constraints edit the sums by point afterwards: by point then
    static const int by_row_min = getenv("FROG_REF_BY_ROW_MIN") ? atoi(getenv("FROG_REF_BY_ROW_MIN")) : 300;
    ctx->rc_by_row = !ctx->n_hard && !ctx->ref_literal && n_keys / std::max<uint64_t>(1, n_gnodes) >= (uint64_t)by_row_min;
    ref_chain_group_kernel<<<div_up(n_slots, 256), 256, 0, s>>>(ctx->rc_len_sorted.p, (uint32_t)n_slots, n_groups, (uint32_t)ctx->rc_unroll, ctx->rc_group_len.p,
                                                                                  ctx->rc_group_size.p, ctx->rc_slot_node.p, ctx->rc_slot_of_node.p);
    tb = ctx->rc_temp.n;
    FROG_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(ctx->rc_temp.p, tb, ctx->rc_group_size.p, ctx->rc_group_ptr.p, (size_t)n_groups + 1, s));
    FROG_HIP_CHECK(hipGetLastError());
    // the layout's size decides two allocations and the fill's grid: one round trip per lattice
    uint64_t seats = 0;
    uint32_t n_entries = 0, longest = 0;
    FROG_HIP_CHECK(hipMemcpyAsync(&seats, ctx->rc_group_ptr.p + n_groups, sizeof seats, hipMemcpyDeviceToHost, s));
    FROG_HIP_CHECK(hipMemcpyAsync(&n_entries, ctx->rc_node_ptr.p + n_gnodes, sizeof n_entries, hipMemcpyDeviceToHost, s));
    FROG_HIP_CHECK(hipMemcpyAsync(&longest, ctx->rc_group_len.p, sizeof longest, hipMemcpyDeviceToHost, s));      // the first group's: sorted descending
    mark("kernels queued");
    FROG_HIP_CHECK(hipStreamSynchronize(s));
    mark("sorted, sizes on the host");
    if (trace) {
        DevBuf<unsigned long long> chk;
        FROG_HIP_CHECK(chk.alloc(4));
        FROG_HIP_CHECK(hipMemsetAsync(chk.p, 0, 4 * sizeof(unsigned long long), s));
        ref_chain_check_kernel<<<div_up(n_slots, 256), 256, 0, s>>>(ctx->rc_node_ptr.p, ctx->rc_slot_node.p, ctx->rc_group_len.p, (uint32_t)n_slots, chk.p);
        unsigned long long h[4];
        FROG_HIP_CHECK(hipMemcpyAsync(h, chk.p, sizeof h, hipMemcpyDeviceToHost, s));
        FROG_HIP_CHECK(hipStreamSynchronize(s));
        std::fprintf(stderr, "[ref_chain_build] check: %llu chains longer than their group, %llu empty slots (expected %llu), control points sum %llu (expected %llu)\n",
                     h[0], h[1], (unsigned long long)(n_slots - n_gnodes), h[2], (unsigned long long)n_gnodes * (n_gnodes - 1) / 2);
    }
    FROG_HIP_CHECK(ctx->rc_ent.alloc(std::max<uint64_t>(1, seats), std::max<uint64_t>(1, seats + seats / 4)));
    FROG_HIP_CHECK(ctx->rc_wt.alloc(std::max<uint64_t>(1, seats), std::max<uint64_t>(1, seats + seats / 4)));
    const unsigned j_tiles = div_up(std::max(1u, longest), 64u);
    if (n_entries && !ctx->ref_literal && j_tiles <= 65535u) {
        const dim3 gg = rc_grid(n_groups);
        ref_chain_fill_tiled_kernel<<<dim3(gg.x, j_tiles, gg.y), 256, 0, s>>>(sorted, rbits, ctx->rc_node_ptr.p, ctx->rc_slot_node.p, ctx->rc_group_ptr.p,
                                                                             ctx->rc_group_len.p, ctx->pos.p, ctx->new_of_old.p, ctx->d_poff.p, ctx->ib,
                                                                             ctx->own_pt_begin, gd, ctx->rc_by_row ? 1 : 0, n_groups, ctx->rc_ent.p, ctx->rc_wt.p);
    } else {
        FROG_HIP_CHECK(hipMemsetAsync(ctx->rc_ent.p, 0xFF, std::max<uint64_t>(1, seats) * sizeof(uint32_t), s));
        if (n_entries)
            ref_chain_fill_kernel<<<div_up(n_entries, 256), 256, 0, s>>>(sorted, n_entries, rbits, ctx->rc_node_ptr.p, ctx->rc_slot_of_node.p, ctx->rc_group_ptr.p,
                                                                        ctx->pos.p, ctx->new_of_old.p, ctx->d_poff.p, ctx->ib, ctx->own_pt_begin, gd,
                                                                        ctx->rc_by_row ? 1 : 0, ctx->rc_ent.p, ctx->rc_wt.p);
    }
    FROG_HIP_CHECK(hipGetLastError());
    mark("layout allocated, fill queued");
    if (trace) std::fprintf(stderr, "[ref_chain_build] %u control points, %u entries, %llu seats, %d entries per step\n", n_gnodes, n_entries, (unsigned long long)seats, ctx->rc_unroll);
    ctx->rc_valid = true;
    return FROG_OK;
}

// phase A of updateDeformableTransforms (imageGroup.cxx:239-377 and the proposal sums of :411-415) in the reference's order
__global__ void energy_fold_kernel(const double *energy, double *tail) { tail[0] = energy[0]; tail[1] = energy[1]; tail[2] = 0.0; tail[3] = energy[3]; }

static int ref_deformable_phase_a(frog_ctx *ctx, float alpha)
{
    hipStream_t s = ctx->stream;
    const GeomDev gd = to_dev(ctx->geom);
    const uint32_t nO = ctx->n_owned();
    int rc = ref_point_sums(ctx, true);
    if (rc) return rc;
    rc = join_setup(ctx);                       // the set-up's stream zeroes and sorts: wait before gradf is touched
    if (rc) return rc;
    if (ctx->ref_literal) {
        FROG_HIP_CHECK(hipMemsetAsync(ctx->gradf.p, 0, (size_t)nO * gd.n_cp * sizeof(float4), s));       // Fill(0), :249
        ref_scatter_kernel<<<nO, 64, 0, s>>>(ctx->pos.p, ctx->point_sums.p, ctx->new_of_old.p, ctx->d_poff.p, ctx->ib, ctx->own_pt_begin, gd,
                                             ctx->gradf.p);
    } else {
        rc = ref_chain_build(ctx);
        if (rc) return rc;
        const float4 *sums = ctx->rc_by_row ? ctx->ref_row_sums.p : ctx->point_sums.p;
        if (ctx->rc_unroll == 16)
            ref_chain_kernel<16><<<rc_grid(ctx->rc_n_groups), 64, 0, s>>>(ctx->rc_ent.p, ctx->rc_wt.p, ctx->rc_group_ptr.p, ctx->rc_group_len.p, ctx->rc_slot_node.p,
                                                                         sums, ctx->rc_n_groups, ctx->gradf.p);
        else
            ref_chain_kernel<8><<<rc_grid(ctx->rc_n_groups), 64, 0, s>>>(ctx->rc_ent.p, ctx->rc_wt.p, ctx->rc_group_ptr.p, ctx->rc_group_len.p, ctx->rc_slot_node.p,
                                                                        sums, ctx->rc_n_groups, ctx->gradf.p);
    }
    ref_cp_step_kernel<<<div_up(gd.n_cp, 256), 256, 0, s>>>(ctx->gradf.p, ctx->coeff.p, ctx->grad.p, nO, gd.n_cp, alpha, ctx->gridsum.p);
    if (ctx->ref_join_pending) { FROG_HIP_CHECK(hipStreamWaitEvent(s, ctx->ref_join, 0)); ctx->ref_join_pending = false; }      // the energy sums (ref_point_sums)
    if (ctx->two_collectives) energy_fold_kernel<<<1, 1, 0, s>>>(ctx->energy.p, ctx->gridsum.p + 3 * (size_t)gd.n_cp);      // as lattice_step_kernel
    FROG_HIP_CHECK(hipGetLastError());
    ctx->centered_in_a = false;                 // phase B subtracts the mean (cp_center_kernel: :417-428 as written)
    ctx->pending_alpha = alpha;
    ctx->phase = 1;
    return FROG_OK;
}

extern "C" {

int frog_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int frog_device_warm(int device)
{
    if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return FROG_E_NODEVICE; }
    (void)hipFree(nullptr);
    return FROG_OK;
}

const char *frog_last_error(void) { return g_last_error.c_str(); }

void frog_destroy(frog_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->helper) { frog_destroy(ctx->helper); ctx->helper = nullptr; }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
#ifdef FROG_W1_COUNT
    {
        unsigned long long h[8];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_w1_count), sizeof h);
        fprintf(stderr, "W1 steps %llu with-general %llu  lanes %llu general %llu skipped %llu general-below-lo %llu\n", h[0], h[1], h[2], h[3], h[4], h[5]);
        std::vector<float4> em(ctx->nI); std::vector<frog::EmFast> ef(ctx->nI);
        (void)hipMemcpy(em.data(), ctx->em.p, em.size() * 16, hipMemcpyDeviceToHost); (void)hipMemcpy(ef.data(), ctx->emf.p, ef.size() * 16, hipMemcpyDeviceToHost);
        for (uint32_t i = 0; i < ctx->nI && i < 6; i++) fprintf(stderr, "  image %u c1 %g c2 %g r %g   l %g ds %g lo %g hi %g (d_hi %g)\n", i, em[i].x, em[i].y, em[i].z, ef[i].l, ef[i].ds, ef[i].lo, ef[i].hi, sqrt(ef[i].hi));
    }
#endif
#ifdef FROG_SWEEP_TRACE
    if (const char *path = getenv("FROG_SWEEP_TRACE_FILE")) {
        std::vector<unsigned long long> h(8 * 8 * 16384);
        (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_sweep_trace), h.size() * 8);
        if (FILE *fp = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), fp); fclose(fp); }
    }
#endif
#ifdef FROG_SCATTER_TRACE
    if (const char *path = getenv("FROG_SCATTER_TRACE_FILE")) {
        std::vector<unsigned long long> h(4 * 65536);
        (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_scatter_trace), h.size() * 8);
        if (FILE *fp = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), fp); fclose(fp); }
    }
#endif
    if (ctx->side) { (void)hipStreamSynchronize(ctx->side); (void)hipStreamDestroy(ctx->side); }
    if (ctx->ref_stream) { (void)hipStreamSynchronize(ctx->ref_stream); (void)hipStreamDestroy(ctx->ref_stream); }
    if (ctx->ref_fork) (void)hipEventDestroy(ctx->ref_fork);
    if (ctx->ref_join) (void)hipEventDestroy(ctx->ref_join);
    if (ctx->setup_stream) { (void)hipStreamSynchronize(ctx->setup_stream); (void)hipStreamDestroy(ctx->setup_stream); }
    if (ctx->setup_fork) (void)hipEventDestroy(ctx->setup_fork);
    if (ctx->setup_join) (void)hipEventDestroy(ctx->setup_join);
    for (hipEvent_t e : ctx->sel_done) if (e) (void)hipEventDestroy(e);
    if (ctx->energy_copied) (void)hipEventDestroy(ctx->energy_copied);
    for (hipEvent_t e : ctx->ord_read) if (e) (void)hipEventDestroy(e);
    for (auto &sp : ctx->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto &ev : ctx->free_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (ctx->h_energy) (void)hipHostFree(ctx->h_energy);
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int frog_create(const frog_model *m, const frog_options *o, int device,
                uint32_t image_begin, uint32_t image_end, frog_ctx **out)
{
    if (!m || !o || !out) return fail(FROG_E_INVALID, "null argument");
    *out = nullptr;
    if (m->n_images < 1 || image_begin >= image_end || image_end > m->n_images)
        return fail(FROG_E_INVALID, "bad image range");
    for (size_t i = 0; i < sizeof(o->reserved) / sizeof(o->reserved[0]); i++)
        if (o->reserved[i]) return fail(FROG_E_INVALID, "reserved option fields must be 0");
    if (o->stats_max_size < 1) return fail(FROG_E_INVALID, "stats_max_size < 1");
    if (o->n_fixed_images) {
        // fixed images (-fi, imageGroup.cxx:34): a context for the moving images plus a stats-only one
        const uint32_t nf = (uint32_t)o->n_fixed_images;
        if (o->n_fixed_images < 0 || nf >= m->n_images) return fail(FROG_E_INVALID, "n_fixed_images must leave at least one moving image");
        if (image_begin != 0 || image_end != m->n_images)
            return fail(FROG_E_INVALID, "n_fixed_images needs the whole group in one context");
        frog_options o2 = *o;
        o2.n_fixed_images = 0;
        frog_ctx *moving = nullptr, *fixed = nullptr;
        int rc = frog_create(m, &o2, device, nf, m->n_images, &moving);
        if (rc) return rc;
        frog_options o3 = o2;
        o3.max_levels_hint = 0;                                     // a statistics-only context never holds a lattice
        rc = frog_create(m, &o3, device, 0, nf, &fixed);
        if (rc) { frog_destroy(moving); return rc; }
        fixed->cull_enabled = false;                                // a statistics-only context never sweeps a lattice step
        fixed->act_recs32.release(); fixed->act_recs.release(); fixed->act_cnt.release(); fixed->pos2_snap.release();
        rc = frog_set_stream(fixed, moving->stream);
        if (rc) { frog_destroy(fixed); frog_destroy(moving); return rc; }
        moving->helper = fixed;
        moving->nf = nf;
        moving->opt.n_fixed_images = o->n_fixed_images;
        *out = moving;
        return FROG_OK;
    }
    // (the first HIP call of a process waits for the runtime to come up -- 0.05-0.2 s; bin/frog starts that on a thread of its own
    // beside readPairs (frog_device_warm), and the host-side layout build below needs no device: the device is asked for after it)
    frog_ctx *c = new (std::nothrow) frog_ctx;
    if (!c) return fail(FROG_E_NOMEM, "out of host memory");
    c->device = device;
    c->opt = *o;
    c->nI = m->n_images; c->ib = image_begin; c->ie = image_end;
    c->poff.assign(m->point_offset, m->point_offset + c->nI + 1);
    c->P = c->poff[c->nI];
    c->own_pt_begin = c->poff[c->ib]; c->own_pt_end = c->poff[c->ie];
    {
        // Points per scatter block (k_grid.hip.h): SCATTER_CHUNK for every context.  Round 6 tried shorter blocks for contexts that
        // own few points (a rank of eight of the benchmark group has 625 blocks of 384 for 1 024 SIMDs, its scatter lasts one block's
        // 31 us): with 128 the scatter took 18 us -- and the lattice step, which adds a brick's staged tiles slot by slot, 40 instead
        // of 21 (level 0; 30 / 19 and 33 / 26 on levels 1 and 2): 7 287 against 7 365 rank-iterations/s, and a brick's points are summed
        // per block, so shardings would no longer agree to the bit.  The switch stays for experiments.
        uint32_t chunk = (uint32_t)SCATTER_CHUNK;
        if (const char *e = getenv("FROG_SCATTER_CHUNK_POINTS")) chunk = std::min<uint32_t>((uint32_t)SCATTER_CHUNK, std::max<uint32_t>(64u, (uint32_t)atoi(e) / 64u * 64u));
        c->scatter_chunk = chunk;
    }
    if (c->P >= 0x7FFFFFFFull) { delete c; return fail(FROG_E_INVALID, "more than 2^31-1 points"); }

    Layout lay;
    std::string err;
    // FROG_WIDE_RECORDS=1 keeps the 8-byte record form where the 4-byte one would fit (test hook)
    const char *wide_env = getenv("FROG_WIDE_RECORDS");
    // partner groups: 8 (one sweep launch per pass) unless FROG_SUBPASSES asks for 8 * n launches-worth
    // (ctx.h: measured no gain from keeping the slices L2-sized, so it is not automatic)
    c->n_sub = 1;
    if (const char *e = getenv("FROG_SUBPASSES")) c->n_sub = (uint32_t)std::min(MAX_SUBPASS, std::max(1, atoi(e)));
    c->n_groups = N_XCD * c->n_sub;
    const auto t_create0 = std::chrono::steady_clock::now();
    int rc = build_layout(*m, c->ib, c->ie, wide_env && wide_env[0] == '1', (int)c->n_groups, lay, err);
    if (rc) { delete c; return fail(rc, err); }
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { (void)hipGetLastError(); delete c; return fail(FROG_E_NODEVICE, "no HIP device: libfrog_hip has no CPU fallback"); }
        if (device < 0 || device >= ndev) { delete c; return fail(FROG_E_INVALID, "device index out of range"); }
        if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); delete c; return fail(FROG_E_HIP, "hipSetDevice failed"); }
    }
    const auto t_layout = std::chrono::steady_clock::now();
    c->create_s[0] = std::chrono::duration<double>(t_layout - t_create0).count();
    c->L_own = lay.ref_link.size();
    c->rec_format = lay.format;
    for (uint32_t g = 0; lay.format.narrow && g < c->n_groups; g++)
        if (lay.group_begin[g + 1] - lay.group_begin[g] > (uint32_t)EMD_LDS_IMAGES) {
            delete c;
            return fail(FROG_E_INVALID, "internal: narrow link records chosen for a partner group of more than 256 images");
        }
    c->L_recs = lay.format.narrow ? lay.recs32.size() : lay.recs.size();
    c->n_tiles = (uint32_t)lay.tiles.size();
    for (uint32_t g = 0; g <= c->n_groups; g++) c->group_begin[g] = lay.group_begin[g];
    c->h_old_of_new = lay.old_of_new;
    c->h_new_of_old = lay.new_of_old;
    c->h_img_tile_ptr = lay.img_tile_ptr;
    c->img_link_begin = lay.img_link_begin;

#define CREATE_CHECK(expr)                                                                           \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            std::string msg_ = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            frog_destroy(c);                                                                         \
            return fail(e_ == hipErrorOutOfMemory ? FROG_E_NOMEM : FROG_E_HIP, msg_);                \
        }                                                                                            \
    } while (0)

    auto create_lap = [&](const char *what) {
        if (getenv("FROG_TIMING")) std::printf("[timing] frog_create device part, %s : %gs\n", what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_layout).count());
    };
    CREATE_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    c->own_stream = true;
    {
        // retired lattices are allocated stream-ordered (hipMallocAsync): keep freed blocks in the pool instead of
        // returning them to the driver at the next synchronisation
        hipMemPool_t pool = nullptr;
        if (hipDeviceGetDefaultMemPool(&pool, device) == hipSuccess && pool) {
            uint64_t keep = ~0ull;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
        }
        (void)hipGetLastError();
    }
    create_lap("stream, memory pool");
    // [0..3] energy scalars, [4..6] bounding box (6 floats), [7] sequence number of the step whose scalars a kernel wrote here
    CREATE_CHECK(hipHostMalloc((void **)&c->h_energy, 8 * sizeof(double), hipHostMallocMapped));
    std::memset(c->h_energy, 0, 8 * sizeof(double));
    if (hipHostGetDevicePointer((void **)&c->h_energy_dev, c->h_energy, 0) != hipSuccess) { (void)hipGetLastError(); c->h_energy_dev = nullptr; }
    if (getenv("FROG_SCALARS_COPY")) c->h_energy_dev = nullptr;      // the copy + event hand-off (A/B, fallback)
    hipStream_t s = c->stream;

    create_lap("+ pinned scalar block");
    // points: xyz | image id
    std::vector<float4> hp(c->P);
    c->h_img_bbox.assign((size_t)c->nI * 6, 0.0);
    #pragma omp parallel for schedule(dynamic, 1) num_threads(host_threads())
    for (int64_t i = 0; i < (int64_t)c->nI; i++) {
        double mn[3] = { std::numeric_limits<double>::max(), std::numeric_limits<double>::max(), std::numeric_limits<double>::max() };
        double mx[3] = { -mn[0], -mn[1], -mn[2] };
        for (uint32_t p = c->poff[i]; p < c->poff[i + 1]; p++) {
            float4 v;
            v.x = m->xyz[3 * (size_t)p]; v.y = m->xyz[3 * (size_t)p + 1]; v.z = m->xyz[3 * (size_t)p + 2];
            int id = (int)i;
            std::memcpy(&v.w, &id, 4);
            hp[lay.new_of_old[p]] = v;
            const double q[3] = { v.x, v.y, v.z };
            for (int k = 0; k < 3; k++) { if (q[k] < mn[k]) mn[k] = q[k]; if (q[k] > mx[k]) mx[k] = q[k]; }
        }
        for (int k = 0; k < 3; k++) { c->h_img_bbox[(size_t)i * 6 + k] = mn[k]; c->h_img_bbox[(size_t)i * 6 + 3 + k] = mx[k]; }
    }
    create_lap("+ points renumbered on the host");
    CREATE_CHECK(c->pos.upload(hp, s));
    create_lap("+ first upload queued");
    {
        std::vector<P3> hp2(c->P);
        #pragma omp parallel for num_threads(host_threads())
        for (int64_t p = 0; p < (int64_t)c->P; p++) hp2[p] = P3{ hp[p].x, hp[p].y, hp[p].z };
        CREATE_CHECK(c->pos2.upload(hp2, s));
        CREATE_CHECK(hipStreamSynchronize(s));
    }
    create_lap("+ points uploaded"); 
    CREATE_CHECK(c->d_poff.upload(c->poff, s));
    CREATE_CHECK(c->point_sums.alloc(c->P));
    CREATE_CHECK(hipMemsetAsync(c->point_sums.p, 0, c->point_sums.bytes(), s));

    CREATE_CHECK(c->ref_rowptr.upload(lay.ref_rowptr, s));
    CREATE_CHECK(c->ref_link.upload(lay.ref_link, s));
    {
        std::vector<uint32_t> own(lay.new_of_old.begin() + c->own_pt_begin, lay.new_of_old.begin() + c->own_pt_end);
        CREATE_CHECK(c->new_of_old.upload(own, s));
        CREATE_CHECK(hipStreamSynchronize(s));
    }
    create_lap("+ reference-order links uploaded");
    CREATE_CHECK(c->tiles.upload(lay.tiles, s));
    if (lay.format.narrow) { CREATE_CHECK(c->recs32.upload(lay.recs32, s)); }
    else { CREATE_CHECK(c->recs.upload(lay.recs, s)); }
    CREATE_CHECK(c->img_tile_ptr.upload(lay.img_tile_ptr, s));
    create_lap("+ records queued");
    {
        // Fused deformable sweep (k_links.hip.h FUSED): block b works on tile order[b], and b % 8 -- the XCD the block lands on
        // under round-robin dispatch -- is the tile's eighth of its image along the Morton curve, so that an XCD's L2 sees the
        // same eighth of every image: its own tiles' and, since true matches are spatial neighbours, nearly all their partners'.
        // Within an XCD's list the tiles go slice by slice, image by image inside a slice (a slice = one of FROG_TILE_SLICES
        // equal parts of the eighth, along the curve): the blocks resident on an XCD at any time then gather from 1/16 of every
        // partner image instead of 1/8 (1.5 MB of coordinates instead of 3 MB beside the record stream in a 4 MB L2).
        // Measured on cfg 3: 0.2486 -> 0.2446 ms with two slices; 3, 4 and 8 slices the same as two (0.2451-0.2460).  A context
        // that owns an eighth of that group (12-13 images, 118 tiles per list: little more than one round of resident blocks)
        // keeps gaining: 0.0702 / 0.0663 / 0.0651 / 0.0641 ms with 1 / 2 / 4 / 8 slices -- hence eight below 64 owned images.
        std::vector<uint32_t> lists[N_XCD];
        const int n_slices = [&] {
            const char *e = getenv("FROG_TILE_SLICES");
            const int v = e ? atoi(e) : (c->ie - c->ib >= 64u ? 2 : 8);
            return v < 1 ? 1 : v > 64 ? 64 : v;
        }();
        for (int sl = 0; sl < n_slices; sl++)
            for (uint32_t i = c->ib; i < c->ie; i++) {
                const uint32_t tb = lay.img_tile_ptr[i], nt = lay.img_tile_ptr[i + 1] - tb;
                for (uint32_t j = 0; j < nt; j++)
                    if ((int)(((size_t)j * N_XCD * n_slices / nt) % n_slices) == sl) lists[(size_t)j * N_XCD / nt].push_back(tb + j);
            }
        size_t rounds = 0;
        for (auto &l : lists) rounds = std::max(rounds, l.size());
        std::vector<uint32_t> order(std::max<size_t>(1, rounds * N_XCD), 0xFFFFFFFFu);
        for (int x = 0; x < N_XCD; x++)
            for (size_t r = 0; r < lists[x].size(); r++) order[r * N_XCD + x] = lists[x][r];
        c->n_order_blocks = (uint32_t)(rounds * N_XCD);
        CREATE_CHECK(c->tile_order.upload(order, s));
        {
            std::vector<Tile> bo(order.size(), Tile{});
            for (size_t b = 0; b < order.size(); b++) if (order[b] != 0xFFFFFFFFu) bo[b] = lay.tiles[order[b]];
            CREATE_CHECK(c->tiles_bo.upload(bo, s));
            CREATE_CHECK(hipStreamSynchronize(s));
        }
        CREATE_CHECK(hipStreamSynchronize(s));
        // static + dynamic LDS of the fused block must stay under the 64 KB a block gets without asking for more
        uint32_t widest = 0;
        for (uint32_t g = 0; g < c->n_groups; g++) widest = std::max(widest, c->group_begin[g + 1] - c->group_begin[g]);
        const char *fe = getenv("FROG_SWEEP_FUSED");
        c->fused_sweep = c->n_sub == 1 && c->n_tiles > 0 && widest <= (uint32_t)EMD_LDS_IMAGES && sweep_lds_images(c) <= 64u
                         && !(fe && fe[0] == '0');
        c->fused_forced = fe && (fe[0] == '1' || fe[0] == '2');
    }
    CREATE_CHECK(c->tile_partial.alloc((size_t)std::max(1u, c->n_tiles) * c->n_groups * LINEAR_SUMS));
    CREATE_CHECK(c->tile_counts.alloc((size_t)std::max(1u, c->n_tiles) * c->n_groups * 2));
    CREATE_CHECK(c->group_sums.alloc((size_t)N_XCD * std::max(1u, c->own_pt_end - c->own_pt_begin)));
    CREATE_CHECK(c->img_counts.alloc((size_t)c->n_owned() * 2));

    // statistics: Stats ctor (stats.h:94-99) + setupStats (imageGroup.cxx:1151-1159)
    std::vector<float4> hem(c->nI, make_float4(10.f, 300.f, 0.5f, 0.f));
    CREATE_CHECK(c->em.upload(hem, s));
    CREATE_CHECK(c->emd.alloc(c->nI));
    CREATE_CHECK(c->emf.alloc(c->nI));
    c->h_virtual.resize(c->n_owned());
    uint32_t cap = 1;
    for (uint32_t i = c->ib; i < c->ie; i++) {
        uint64_t v = lay.img_link_begin[i + 1] - lay.img_link_begin[i];
        c->h_virtual[i - c->ib] = (uint32_t)v;
        cap = std::max<uint32_t>(cap, (uint32_t)std::min<uint64_t>(v, (uint64_t)o->stats_max_size));
    }
    c->sample_cap = (int)cap;
    CREATE_CHECK(c->d_virtual.upload(c->h_virtual, s));
    CREATE_CHECK(c->samples.alloc((size_t)c->n_owned() * cap));
    CREATE_CHECK(c->em_guess.alloc((size_t)c->n_owned() * 4 * EM_GUESS_BATCHES));
    CREATE_CHECK(hipMemsetAsync(c->em_guess.p, 0, c->em_guess.bytes(), s));
    // ring of pre-computed selections (ctx.h): FROG_SELECT_RING buffers (default 6: five refreshes ahead), fewer when they
    // would take more than 1 GB
    c->sel_ring = 80;
    if (const char *e = getenv("FROG_SELECT_RING")) c->sel_ring = atoi(e);
    c->sel_ring = std::max(2, std::min(c->sel_ring, (int)frog_ctx::SEL_RING_MAX));
    while (c->sel_ring > 2 && (size_t)c->sel_ring * c->n_owned() * cap * 12 > ((size_t)1 << 30)) c->sel_ring--;
    for (int b = 0; b < c->sel_ring; b++) {
        CREATE_CHECK(c->sample_ord[b].alloc((size_t)c->n_owned() * cap));
        CREATE_CHECK(c->sample_ends[b].alloc((size_t)c->n_owned() * cap));
        CREATE_CHECK(c->sample_count[b].alloc(c->n_owned()));
        CREATE_CHECK(hipMemsetAsync(c->sample_count[b].p, 0, c->sample_count[b].bytes(), s));
        CREATE_CHECK(hipEventCreateWithFlags(&c->ord_read[b], hipEventDisableTiming));
        CREATE_CHECK(hipEventCreateWithFlags(&c->sel_done[b], hipEventDisableTiming));
    }
    {
        // The replay is a latency chain on a few wavefronts per image.  With one selection ahead it had to finish within ten
        // iterations and ran at high priority -- where its wavefronts take issue slots from the kernels of the iteration
        // (measured on cfg 3 while it runs: sweeps +4 %, the one-block-per-CU lattice step +60 %).  With a ring it has
        // (sel_ring - 1) x ten iterations and runs at LOW priority (FROG_SELECT_PRIORITY=high restores the old choice).
        int lo = 0, hi = 0;
        CREATE_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        const char *pe = getenv("FROG_SELECT_PRIORITY");
        const bool high = pe ? pe[0] == 'h' : c->sel_ring < 3;
        CREATE_CHECK(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, high ? hi : lo));
    }
    CREATE_CHECK(hipEventCreateWithFlags(&c->energy_copied, hipEventDisableTiming));
    CREATE_CHECK(hipStreamCreateWithFlags(&c->setup_stream, hipStreamNonBlocking));
    CREATE_CHECK(hipEventCreateWithFlags(&c->setup_fork, hipEventDisableTiming));
    CREATE_CHECK(hipEventCreateWithFlags(&c->setup_join, hipEventDisableTiming));
    if (const char *e = getenv("FROG_SETUP_STREAM")) c->setup_async = atoi(e) != 0;
    {
        // std::mt19937::seed(0); index 624 forces a regeneration at the first draw
        std::vector<uint32_t> st((size_t)c->n_owned() * MT_WORDS);
        uint32_t x[MT_WORDS];
        x[0] = 0u;
        for (int k = 1; k < MT_N; k++) x[k] = 1812433253u * (x[k - 1] ^ (x[k - 1] >> 30)) + (uint32_t)k;
        x[MT_N] = MT_N;
        for (uint32_t i = 0; i < c->n_owned(); i++) std::memcpy(&st[(size_t)i * MT_WORDS], x, sizeof x);
        CREATE_CHECK(c->mt_state.upload(st, s));
    }

    std::vector<double> hm((size_t)c->nI * 16, 0.0);
    for (uint32_t i = 0; i < c->nI; i++) for (int k = 0; k < 4; k++) hm[(size_t)i * 16 + 5 * k] = 1.0;
    CREATE_CHECK(c->mat.upload(hm, s));
    CREATE_CHECK(c->energy.alloc(4));
    CREATE_CHECK(c->energy_blocks.alloc(2 * ENERGY_BLOCKS));
    CREATE_CHECK(c->img_energy.alloc(2 * (size_t)std::max(1u, c->n_owned())));
    CREATE_CHECK(c->energy_ticket.alloc(2));
    CREATE_CHECK(hipMemsetAsync(c->energy_ticket.p, 0, 2 * sizeof(unsigned int), s));
    CREATE_CHECK(hipMemsetAsync(c->energy.p, 0, c->energy.bytes(), s));
    CREATE_CHECK(c->stray.alloc(3));
    CREATE_CHECK(hipMemsetAsync(c->stray.p, 0, 3 * sizeof(unsigned int), s));
    CREATE_CHECK(c->bounds_scratch.alloc((size_t)BOUNDS_BLOCKS * 6 + 6));
    em_derive_kernel<<<div_up(c->nI, 256), 256, 0, s>>>(c->em.p, c->emd.p, c->emf.p, c->nI, c->fast_theta());
    CREATE_CHECK(hipGetLastError());
    // certified outlier culling (k_cull.hip.h): FROG_CULL=0 off; FROG_CULL_SKIN="scale,pad" sets the list cutoff
    if (const char *e = getenv("FROG_CULL")) c->cull_enabled = atoi(e) != 0;
    // without the culling list a third of a typical group's gathers are false matches with random partners: the fused sweep's
    // spatial XCD mapping does nothing for them, the per-group form keeps them in L2 (unless FROG_SWEEP_FUSED=1 insists)
    if (!(c->cull_enabled && c->opt.inlier_threshold >= 1e-3f) && !c->fused_forced) c->fused_sweep = false;
    // test hook: every inlier weight through the form with the reference's own promotions (ten times the arithmetic)
    if (const char *e = getenv("FROG_WEIGHT_EXACT")) c->exact_weights = atoi(e) != 0;
    if (const char *e = getenv("FROG_WEIGHT_GENERAL")) c->general_weights = atoi(e) != 0;
    // test hook: the solver loops in the reference's own order and arithmetic (k_reforder.hip.h); no list, no fast weight
    c->ref_order = o->reference_order != 0;                    // frog_options::reference_order (bin/frog -exact 1)
    if (const char *e = getenv("FROG_REFERENCE_ORDER")) c->ref_order = atoi(e) != 0;        // the tests' switch, overrides
    if (const char *e = getenv("FROG_K11_F64")) c->k11_f64 = atoi(e) != 0;
    if (c->ref_order) { c->cull_enabled = false; c->exact_weights = true; c->fused_sweep = false; }
    c->ref_literal = getenv("FROG_REF_LITERAL") != nullptr;
    if (const char *e = getenv("FROG_CULL_LINEAR")) c->cull_linear = atoi(e) != 0;
    if (const char *e = getenv("FROG_CULL_SKIN_LINEAR")) {
        float a = 0, b = 0;
        if (sscanf(e, "%f,%f", &a, &b) == 2 && a >= 1.0f && b >= 0.0f) { c->cull_lin_scale = a; c->cull_lin_pad = b; }
    }
    if (const char *e = getenv("FROG_CULL_SKIN")) {
        float a = 0, b = 0;
        if (sscanf(e, "%f,%f", &a, &b) == 2 && a >= 1.0f && b >= 0.0f) { c->cull_scale = a; c->cull_pad = b; }
    }
    CREATE_CHECK(c->cut_now.alloc(c->nI));
    CREATE_CHECK(c->lin_listed.alloc(1));
    CREATE_CHECK(hipMemsetAsync(c->lin_listed.p, 0, sizeof(unsigned long long), s));
    if (c->cull_enabled && (c->opt.inlier_threshold >= 1e-3f || c->cull_linear) && c->n_tiles > 0 && !getenv("FROG_CULL_LAZY")) {
        if (int rc_ = cull_allocate(c)) { frog_destroy(c); return rc_; }
    }
    stats_publish_kernel<<<dim3(div_up(c->nI, 64), 2), 64, 0, s>>>(c->em.p, c->emd.p, c->emf.p, c->nI, c->opt.inlier_threshold, c->fast_theta(), c->cut_now.p, 1);
    CREATE_CHECK(hipGetLastError());
    CREATE_CHECK(hipStreamSynchronize(s));      // host staging vectors die here
    create_lap("+ everything else allocated, uploads done");
    if (!getenv("FROG_LATTICE_LAZY")) {
        // the lattice buffers of level 0 (with their head-room) for the box of the model as it is: close enough to what
        // the first frog_deformable_setup will ask for that it finds them allocated
        double mn[3] = { 1e300, 1e300, 1e300 }, mx[3] = { -1e300, -1e300, -1e300 };
        for (uint32_t i = c->ib; i < c->ie; i++)
            for (int k = 0; k < 3; k++) {
                if (c->poff[i + 1] == c->poff[i]) continue;
                mn[k] = std::min(mn[k], c->h_img_bbox[(size_t)i * 6 + k]); mx[k] = std::max(mx[k], c->h_img_bbox[(size_t)i * 6 + 3 + k]);
            }
        GridGeom g0{};
        frog_grid_info i0{};
        if (mn[0] <= mx[0] && make_geometry(c, 0, mn, mx, g0, i0) == FROG_OK) {
            g0.n_cp = (int)std::min<size_t>(0x7FFFFFFF, (size_t)g0.n_cp * 3 / 2);      // the registered box differs a little
            if (int rc_ = lattice_alloc(c, g0)) { (void)rc_; (void)hipGetLastError(); }  // best effort: the set-up allocates again
            // Finished lattices stay on the device (retire_current_grid).  A stream-ordered allocation per lattice cost 0.25 ms
            // of host time with the GPU idle at the set-up of level 2 (the pool grows by a driver call whenever it is asked
            // for a size it has not served yet, however much it holds): one block for them, about what three levels with their
            // regrids take (1 + 2 x 8 + 3 x 64 lattices of level 0), best effort.
            const size_t lattice0 = (size_t)c->n_owned() * (size_t)g0.n_cp;
            size_t arena = std::min<size_t>(((size_t)2 << 30) / sizeof(float4), 224 * lattice0);
            // The caller said how many levels it will run (frog_options::max_levels_hint): the buffers of the FINEST of them now,
            // and an arena for three finished lattices per level.  A group of 500 images reaches 7.7 GB per lattice buffer at its
            // fifth level; hipMalloc of such blocks took between 5 and 1 500 ms on the test boxes, inside the timed loops (the
            // 130-step figure of BASELINE configs[4] moved between 45 and 166 it/s with it).  Best effort, and only while it leaves
            // half of the free memory alone.
            if (o->max_levels_hint > 1) {
                const int finest = std::min(o->max_levels_hint, 12) - 1;
                GridGeom gh{};
                frog_grid_info ih{};
                size_t free_b = 0, total_b = 0;
                // The box the finest level will see is the REGISTERED group's, not the model's: frog_linear_init moves every image's
                // anchor (the centre of its box, by default) onto the images' mean anchor, so the union of the images' boxes centred
                // on one point is what is left of, say, +-100 mm of translation between them.  Estimated here from the per-image
                // boxes; the union of the boxes where they lie (mn, mx) overstated cfg 5's finest lattice 3.8 times (4.4e6 control
                // points against 0.92e6), `need` came to 320 GB of 293 and nothing was reserved: the set-up of level 4 then paid three
                // hipMalloc of 7.4 GB inside the loops -- 17 ms on one box, 730-1 230 ms on another (round 6; FROG_SETUP_TRACE=1).
                double rmn[3], rmx[3];
                for (int k = 0; k < 3; k++) {
                    double centre = 0, half = 0;
                    uint32_t n_img = 0;
                    for (uint32_t i = 0; i < c->nI; i++) {
                        if (c->poff[i + 1] == c->poff[i]) continue;
                        const double lo = c->h_img_bbox[(size_t)i * 6 + k], hi = c->h_img_bbox[(size_t)i * 6 + 3 + k];
                        centre += 0.5 * (lo + hi); half = std::max(half, 0.5 * (hi - lo)); n_img++;
                    }
                    centre /= std::max(1u, n_img);
                    rmn[k] = std::max(mn[k], centre - half); rmx[k] = std::min(mx[k], centre + half);
                    if (!(rmn[k] < rmx[k])) { rmn[k] = mn[k]; rmx[k] = mx[k]; }
                }
                if (make_geometry(c, finest, rmn, rmx, gh, ih) == FROG_OK && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                    gh.n_cp = (int)std::min<size_t>(0x7FFFFFFF, (size_t)gh.n_cp * 3 / 2);       // the linear stage rescales the images; another anchor
                    const size_t nO = c->n_owned(), nPts = c->own_pt_end - c->own_pt_begin;
                    const size_t blocks = scatter_max_blocks((uint32_t)std::min<size_t>(0xFFFFFFFFu, nO * (size_t)gh.n_bricks), (uint32_t)nPts, c->scatter_chunk);
                    const size_t E = (size_t)gh.brick + 3;
                    size_t arena_h = 0;
                    for (int l = 0; l <= finest; l++) {
                        GridGeom gl{};
                        frog_grid_info il{};
                        if (make_geometry(c, l, rmn, rmx, gl, il) == FROG_OK) {
                            arena_h += 3 * ((nO * (size_t)gl.n_cp * 5 / 4 + 63) / 64 * 64);
                            // a sparse lattice retires with its bit maps and shared values (retire_current_grid): room for them too
                            if (gl.sparse) arena_h += 3 * (((nO * (size_t)gl.mask_words + 3) / 4 + (size_t)gl.n_cp) * 5 / 4 + 128);
                        }
                    }
                    // first the buffers of the level itself, then -- if that still leaves half of the free memory -- the arena
                    const size_t need_level = (3 * nO * (size_t)gh.n_cp + blocks * E * E * E) * sizeof(float4)
                                              + 3 * nO * (size_t)gh.n_bricks * (size_t)(gh.brick * gh.brick * gh.brick) * sizeof(uint32_t);
                    const size_t need = need_level + arena_h * sizeof(float4);
                    if (getenv("FROG_SETUP_TRACE"))
                        std::fprintf(stderr, "[create] finest level %d: %d control points x %zu images (with head-room): %.2f GB for the level, %.2f GB with the arena, "
                                     "%.2f GB free of %.2f: %s\n", finest, gh.n_cp, nO, need_level / 1e9, need / 1e9, free_b / 1e9, total_b / 1e9,
                                     need < free_b / 2 ? "both reserved now" : need_level < free_b / 2 ? "the level's buffers reserved now" : "NOT reserved");
                    if (need_level < free_b / 2) {
                        if (int rc_ = lattice_alloc(c, gh)) { (void)rc_; (void)hipGetLastError(); }
                        if (need < free_b / 2) arena = std::max(arena, arena_h);
                    }
                }
                (void)hipGetLastError();
            }
            if (c->retired_arena.alloc(arena) != hipSuccess) (void)hipGetLastError();
            c->retired_used = 0;
        }
    }
    create_lap("+ lattices allocated");
    {
        // Resolve the kernels of the deformable stage now: the runtime creates a kernel's function object at its first
        // launch, which cost the first deformable step of a run 0.1-0.15 ms of host time with the GPU idle (rocprofv3
        // kernel trace: a gap in front of the first transform through a lattice).  Asking for the attributes does the same work.
        const void *kernels[] = {
            (const void *)sweep_kernel<SWEEP_DEFORMABLE, true, false, false, false>, (const void *)sweep_kernel<SWEEP_DEFORMABLE, true, false, true, false>,
            (const void *)sweep_kernel<SWEEP_DEFORMABLE, true, false, false, true>, (const void *)sweep_kernel<SWEEP_COUNT, true, false, false, false>,
            (const void *)scatter_kernel, (const void *)lattice_step_kernel<true>, (const void *)lattice_step_kernel<false>,
            (const void *)transform_bspline_tile_kernel<float>, (const void *)transform_bspline_kernel<float>, (const void *)transform_bspline_kernel<float, 2>, (const void *)transform_zero_lattice_kernel,
            (const void *)cp_center_kernel, (const void *)bounds_kernel, (const void *)bounds_final_kernel, (const void *)zero_buffers_kernel,
            (const void *)brick_count_kernel, (const void *)brick_place_kernel, (const void *)cell_order_kernel, (const void *)brick_chunks_kernel,
            (const void *)scan_block_sums_kernel, (const void *)scan_of_sums_kernel, (const void *)scan_apply_kernel,
            (const void *)block_fill_kernel, (const void *)block_len_base_kernel, (const void *)block_sort_kernel,
            (const void *)cull_list_cutoff_kernel, (const void *)cull_allow_kernel, (const void *)cull_validate_kernel, (const void *)cull_disp_kernel,
            (const void *)count_reduce_kernel, (const void *)combine_groups_kernel, (const void *)energy_reduce_kernel,
            (const void *)lattice_step_kernel<true, LS_CPB_SMALL>, (const void *)lattice_step_kernel<false, LS_CPB_SMALL>,
            (const void *)stats_publish_kernel, (const void *)cull_allow_validate_kernel, (const void *)cull_count_kernel,
            (const void *)sweep_kernel<SWEEP_LINEAR, true, false, true, false>, (const void *)sweep_kernel<SWEEP_LINEAR, true, false, false, false>,
        };

        for (const void *k : kernels) { hipFuncAttributes fa; (void)hipFuncGetAttributes(&fa, k); }
        (void)hipGetLastError();
    }
    scratch_warm_kernel<<<1, 64, 0, s>>>(c->stray.p, 0);
    (void)hipGetLastError();
    (void)hipStreamSynchronize(s);
    create_lap("+ kernels resolved, stream idle");
    const auto t_resident = std::chrono::steady_clock::now();
    c->create_s[1] = std::chrono::duration<double>(t_resident - t_layout).count();
    // selections of the first sel_ring - 1 refreshes, ahead of time -- and awaited: a context leaves frog_create with an idle
    // side stream (80 replays are 0.12 s of one-block-per-image work that would otherwise run beside the first iterations)
    for (int k = 0; k + 1 < c->sel_ring; k++)
        if (int rc_ = produce_selection(c)) { frog_destroy(c); return rc_; }
    if (!c->opt.selections_in_background && hipStreamSynchronize(c->side) != hipSuccess) { (void)hipGetLastError(); }
    c->create_s[2] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_resident).count();
    c->create_selections = c->sel_ring > 0 ? c->sel_ring - 1 : 0;
    c->created = true;
    // (The host copy of the layout -- 0.8 GB for the benchmark group -- dies with this scope: 0.07 s of unmapping.  Round 6 gave it
    // to a thread of its own; the first iterations then ran beside the unmapping of memory the runtime had pinned for the uploads,
    // and in one run of two the linear stage of a timed loop took 52 ms instead of 19.  Not kept.)
#undef CREATE_CHECK
    *out = c;
    return FROG_OK;
}

int frog_create_seconds(frog_ctx *ctx, double seconds3[3], int *selections_replayed)
{
    CTX_GUARD(ctx);
    if (!seconds3) return fail(FROG_E_INVALID, "null output");
    for (int k = 0; k < 3; k++) seconds3[k] = ctx->create_s[k];
    if (selections_replayed) *selections_replayed = ctx->create_selections;
    return FROG_OK;
}

int frog_lattice_reallocations(frog_ctx *ctx, int *count)
{
    CTX_GUARD(ctx);
    if (!count) return fail(FROG_E_INVALID, "null output");
    *count = ctx->lattice_reallocs;
    return FROG_OK;
}

int frog_set_stream(frog_ctx *ctx, void *hip_stream)
{
    CTX_GUARD(ctx);
    { const int rc = join_setup(ctx); if (rc) return rc; }
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->setup_stream) FROG_HIP_CHECK(hipStreamSynchronize(ctx->setup_stream));
    if (ctx->own_stream) FROG_HIP_CHECK(hipStreamDestroy(ctx->stream));
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    scratch_warm_kernel<<<1, 64, 0, ctx->stream>>>(ctx->stray.p, 0);
    (void)hipGetLastError();
    if (ctx->helper) return frog_set_stream(ctx->helper, hip_stream);
    return FROG_OK;
}

int frog_get_stream(frog_ctx *ctx, void **hip_stream, int *device)
{
    if (!ctx) return fail(FROG_E_INVALID, "null context");
    if (hip_stream) *hip_stream = (void *)ctx->stream;
    if (device) *device = ctx->device;
    return FROG_OK;
}

int frog_synchronize(frog_ctx *ctx)
{
    CTX_GUARD(ctx);
    { const int rc = join_setup(ctx); if (rc) return rc; }
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return FROG_OK;
}

uint64_t frog_num_points(const frog_ctx *ctx) { return ctx ? ctx->P : 0; }
uint32_t frog_num_images(const frog_ctx *ctx) { return ctx ? ctx->nI : 0; }

// ---- setupLinearTransforms (imageGroup.cxx:806-848) ------------------------------
int frog_linear_init(frog_ctx *ctx, const float anchor_pos[3])
{
    CTX_GUARD(ctx);
    if (!anchor_pos) return fail(FROG_E_INVALID, "null anchor");
    const uint32_t nI = ctx->nI;
    std::vector<float> anchors((size_t)nI * 3);
    float average[3] = { 0, 0, 0 };
    for (uint32_t i = 0; i < nI; i++)
        for (int j = 0; j < 3; j++) {
            const float cpos = anchor_pos[j];
            const double lo = ctx->h_img_bbox[(size_t)i * 6 + j], hi = ctx->h_img_bbox[(size_t)i * 6 + 3 + j];
            const float a = (float)((double)(1 - cpos) * lo + (double)cpos * hi);
            anchors[3 * (size_t)i + j] = a;
            // :823-824: with fixed images the mean runs over the FIRST nI - nf images (upstream's indexing)
            if (i < nI - ctx->nf) average[j] += a / (float)(nI - ctx->nf);
        }
    std::vector<double> hm((size_t)nI * 16, 0.0);
    for (uint32_t i = 0; i < nI; i++) {
        for (int k = 0; k < 4; k++) hm[(size_t)i * 16 + 5 * k] = 1.0;
        for (int j = 0; j < 3; j++) hm[(size_t)i * 16 + 4 * j + 3] = (double)(average[j] - anchors[3 * (size_t)i + j]);
    }
    FROG_HIP_CHECK(hipMemcpyAsync(ctx->mat.p, hm.data(), hm.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // A linear stage may follow anything the C ABI allows before it (a deformable level, a speculative transform of the old
    // matrices): nothing computed for the previous state may be published, and the list -- whatever criterion it was built
    // for -- is rebuilt under the LINEAR stage's cutoffs before the first linear sweep walks it.
    { const int rc = join_setup(ctx); if (rc) return rc; }
    ctx->deformable = false;
    ctx->phase = 0;
    ctx->xyz2_fresh = false; ctx->res_valid = false;
    ctx->disp_current = false; ctx->disp_spec = false; ctx->disp_others = false;
    ctx->cull_need_build = true; ctx->cull_check_due = true;
    stats_publish_kernel<<<dim3(div_up(ctx->nI, 64), 2), 64, 0, ctx->stream>>>(ctx->em.p, ctx->emd.p, ctx->emf.p, ctx->nI, ctx->opt.inlier_threshold, ctx->fast_theta(), ctx->cut_now.p, 1);
    FROG_HIP_CHECK(hipGetLastError());
    return FROG_OK;
}

// ---- transformPoints (imageGroup.cxx:910-916, image.cxx:3-13) -----------------------
__global__ void slab_trailer_kernel(const double *energy, double *trailer) { write_slab_trailer(energy, trailer); }

// `trailer` (null in every call but frog_transform_points_slab's): where the kernel's first thread leaves the step's four
// scalars as they stand on this rank (k_grid.hip.h write_slab_trailer)
static int launch_transform(frog_ctx *ctx, P3 *out, int apply, bool after_step = false, double scalar_seq = 0.0, double *trailer = nullptr)
{
    double *host_scalars = scalar_seq != 0.0 ? ctx->h_energy_dev : nullptr;
    const uint32_t n = ctx->own_pt_end - ctx->own_pt_begin;
    if (!n) return FROG_OK;                     // a context whose images are all empty: nothing to launch
    Span span(ctx, FROG_K_TRANSFORM);
    if (!ctx->deformable) {
        // a context that owns every moving point measures the displacement against the linear stage's list in the same pass
        const bool with_disp = cull_active_linear(ctx) && ctx->cull_lin_builds > 0 && !ctx->cull_need_build && ctx->pos2_snap.p;
        transform_linear_kernel<<<div_up(n, 256), 256, 0, ctx->stream>>>(ctx->pos.p, out, ctx->mat.p,
                                                                        ctx->own_pt_begin, ctx->own_pt_end, apply,
                                                                        with_disp ? ctx->pos2_snap.p : nullptr, ctx->disp_part.p,
                                                                        ctx->disp_allow.p, ctx->cull_state.p, ctx->energy.p,
                                                                        host_scalars, scalar_seq, trailer);
        trailer = nullptr;
        if (with_disp) ctx->disp_n = ctx->disp_own_n = div_up(n, 256);
        ctx->disp_others = false;
        if (out == ctx->pos2.p) ctx->disp_current = with_disp;
        else { ctx->disp_spec = with_disp; ctx->disp_current = false; }
    } else if (ctx->coeff_zero && !after_step) {
        // a fresh lattice: identity (k_grid.hip.h); the displacement against the culling list's snapshot is measured by
        // the check before the next sweep instead
        ctx->disp_current = false;
        if (out != ctx->pos2.p) ctx->disp_spec = false;
        transform_zero_lattice_kernel<<<div_up(n, 256), 256, 0, ctx->stream>>>(ctx->pos.p, out, ctx->own_pt_begin, ctx->own_pt_end, apply);
    } else if (ctx->ref_order) {
        { const int rc = join_setup(ctx); if (rc) return rc; }
        ctx->disp_current = false;
        if (out != ctx->pos2.p) ctx->disp_spec = false;
        ref_transform_bspline_kernel<<<div_up(n, 256), 256, 0, ctx->stream>>>(ctx->pos.p, out, ctx->coeff.p, ctx->own_pt_begin, ctx->own_pt_end,
                                                                             ctx->ib, to_dev(ctx->geom), apply, after_step ? ctx->grad.p : nullptr,
                                                                             ctx->energy.p, ctx->opt.guarantee_diffeomorphism, host_scalars, scalar_seq);
    } else {
        { const int rc = join_setup_positions(ctx); if (rc) return rc; }
        // a context that owns every moving point measures, in the same pass, how far the points are from the snapshot
        // of the outlier-culling list: the check before the next sweep then has nothing left to compute
        // (a context that owns a sub-range measures its own rows; the other ranks' are measured when they arrive:
        // frog_comm_unpack_slab -- or, when the host gathers some other way, by the check before the sweep)
        const bool with_disp = cull_active(ctx) && ctx->cull_builds > 0 && !ctx->cull_need_build;
        // The tiled form pays the staging of 343 coefficients per block; measured on cfg 3 (points per brick 1 700 / 210 / 80 at
        // levels 0 / 1 / 2): 0.062 / 0.062 / 0.088 ms against 0.100 / 0.100 / 0.109 ms point by point.  Bricks of 8^3 cells are
        // only chosen below 24 points per brick of 4^3 (make_geometry), where staging 1 331 coefficients per block cannot pay.
        // A block is one wavefront walking up to SCATTER_CHUNK points: below about a million points (one rank of eight of cfg 3:
        // 240 000) there are too few of them to fill the chip and the thread-per-point form wins (16 against 29 us).
        // FROG_K11_POINTWISE / FROG_K11_TILED force one form (tests).
        const bool tiled = ctx->n_scatter_blocks > 0 && ctx->geom.brick == 4 && !getenv("FROG_K11_POINTWISE")
                           && (n >= 1000000u || getenv("FROG_K11_TILED"));
        if (tiled) {
            // one wavefront per scatter block, the brick's coefficients in LDS (k_grid.hip.h)
            const GeomDev gd = to_dev(ctx->geom);
            const size_t lds = (size_t)K11_TILE_ENTRIES * sizeof(float4);      // brick == 4 (checked above): the strided 7^3 tile
            auto kernel = ctx->k11_f64 ? transform_bspline_tile_kernel<double> : transform_bspline_tile_kernel<float>;
            // FROG_K11_BY_XCD=1 / 0 forces / forbids the brick-order walk (k_grid.hip.h); default: on lattices with many small blocks
            static const int by_xcd_env = getenv("FROG_K11_BY_XCD") ? atoi(getenv("FROG_K11_BY_XCD")) : -1;
            const bool by_xcd = by_xcd_env >= 0 ? by_xcd_env != 0 : ctx->n_scatter_blocks >= 16384u;
            const uint32_t grid = by_xcd ? ((ctx->n_scatter_blocks + 7u) & ~7u) : ctx->n_scatter_blocks;     // the walk's grid: a multiple of 8
            kernel<<<grid, 64, lds, ctx->stream>>>(
                ctx->pos.p, ctx->pos_b.p, out, ctx->coeff.p, ctx->perm.p,
                reinterpret_cast<const ScatterBlock *>(by_xcd ? ctx->scatter_blocks_tmp.p : ctx->scatter_blocks.p),
                ctx->brick_slot_ptr.p + (size_t)ctx->n_owned() * gd.n_bricks, gd, apply,
                with_disp ? ctx->pos2_snap.p : nullptr, ctx->disp_part.p,
                after_step ? ctx->grad.p : nullptr, ctx->energy.p, ctx->opt.guarantee_diffeomorphism,
                ctx->disp_allow.p, ctx->cull_state.p, host_scalars, scalar_seq, trailer, by_xcd ? ctx->n_scatter_blocks : 0u);
            if (with_disp) ctx->disp_n = ctx->disp_own_n = ctx->n_scatter_blocks;
        } else {
            auto kernel = ctx->k11_f64 ? transform_bspline_kernel<double> : (ctx->geom.brick == 8 ? transform_bspline_kernel<float, 2> : transform_bspline_kernel<float>);
            // FROG_K11_POINT_BY_XCD=1 / 0 forces / forbids the XCD-aware order of the blocks (k_grid.hip.h); default: from 8 192 blocks
            static const int pt_xcd_env = getenv("FROG_K11_POINT_BY_XCD") ? atoi(getenv("FROG_K11_POINT_BY_XCD")) : -1;
            const uint32_t nb = div_up(n, 256);
            const bool pt_xcd = pt_xcd_env >= 0 ? pt_xcd_env != 0 : nb >= 8192u;
            kernel<<<pt_xcd ? ((nb + 7u) & ~7u) : nb, 256, 0, ctx->stream>>>(ctx->pos.p, ctx->pos_b.p, out, ctx->coeff.p,
                                                                             ctx->perm.p, n, ctx->ib, to_dev(ctx->geom), apply,
                                                                             with_disp ? ctx->pos2_snap.p : nullptr, ctx->disp_part.p,
                                                                             after_step ? ctx->grad.p : nullptr, ctx->energy.p,
                                                                             ctx->opt.guarantee_diffeomorphism,
                                                                             ctx->disp_allow.p, ctx->cull_state.p, host_scalars, scalar_seq, trailer,
                                                                             pt_xcd ? nb : 0u);
            if (with_disp) ctx->disp_n = ctx->disp_own_n = div_up(n, 256);
        }
        trailer = nullptr;
        ctx->disp_others = false;
        // the per-block maxima now describe `out`: the current xyz2, or the speculative copy until it is published
        if (out == ctx->pos2.p) ctx->disp_current = with_disp;
        else { ctx->disp_spec = with_disp; ctx->disp_current = false; }
    }
    // the forms that do not write the trailer themselves (fresh lattice, reference-order mode)
    if (trailer) slab_trailer_kernel<<<1, 1, 0, ctx->stream>>>(ctx->energy.p, trailer);
    FROG_HIP_CHECK(hipGetLastError());
    if (apply) { ctx->pos_b_stale = true; ctx->rc_valid = false; }      // pos has new values: pos_b (its copy in perm's order) and the reference-order chains are out of date
    return FROG_OK;
}

int frog_transform_points_local(frog_ctx *ctx, int apply)
{
    CTX_GUARD(ctx);
    const uint32_t n = ctx->own_pt_end - ctx->own_pt_begin;
    if (!n) return FROG_OK;
    if (ctx->xyz2_fresh && !apply) {
        // already computed behind the last deformable step (frog_deformable_phase_c): publish it
        ctx->xyz2_fresh = false; ctx->res_valid = false;
        ctx->disp_current = ctx->disp_spec;
        if (ctx->n_owned() == ctx->nI && !ctx->xyz2_exported && ctx->pos2_spec.n == ctx->pos2.n) {
            std::swap(ctx->pos2.p, ctx->pos2_spec.p);           // whole table recomputed, nobody holds its address
            std::swap(ctx->pos2.cap, ctx->pos2_spec.cap);
            return FROG_OK;
        }
        FROG_HIP_CHECK(hipMemcpyAsync(ctx->pos2.p + ctx->own_pt_begin, ctx->pos2_spec.p + ctx->own_pt_begin,
                                      (size_t)n * sizeof(P3), hipMemcpyDeviceToDevice, ctx->stream));
        return FROG_OK;
    }
    ctx->xyz2_fresh = false; ctx->res_valid = false;
    return launch_transform(ctx, ctx->pos2.p, apply);
}

int frog_transform_points(frog_ctx *ctx, int apply) { return frog_transform_points_local(ctx, apply); }

// ---- updateStats (imageGroup.cxx:569-598) --------------------------------------------
int frog_update_stats_local(frog_ctx *ctx)
{
    CTX_GUARD(ctx);
    const uint32_t nO = ctx->n_owned();
    const uint32_t cap = (uint32_t)ctx->sample_cap;
    hipStream_t s = ctx->stream;
    // consume the oldest selection prepared on the side stream
    const int cur = (int)(ctx->sel_consumed % (uint64_t)ctx->sel_ring);
    FROG_HIP_CHECK(hipStreamWaitEvent(s, ctx->sel_done[cur], 0));
    {
        Span span(ctx, FROG_K_STATS);
        sample_distance_kernel<<<dim3(div_up(cap, 256), nO), 256, 0, s>>>(
            ctx->sample_ends[cur].p, ctx->sample_count[cur].p, cap, ctx->pos2.p, ctx->samples.p);
        FROG_HIP_CHECK(hipGetLastError());
        const bool em_serial = getenv("FROG_EM_SERIAL") != nullptr;            // test hook: the term-by-term form
        if (em_serial)
            em_kernel<<<nO, 256, 0, s>>>(ctx->samples.p, ctx->sample_count[cur].p, cap, ctx->ib, ctx->em.p,
                                         ctx->opt.stats_max_iterations, ctx->opt.stats_epsilon);
        else
            em_scan_kernel<<<nO, EM_THREADS, 0, s>>>(ctx->samples.p, ctx->sample_count[cur].p, cap, ctx->ib, ctx->em.p,
                                                     ctx->opt.stats_max_iterations, ctx->opt.stats_epsilon, ctx->em_guess.p);
        FROG_HIP_CHECK(hipGetLastError());
    }
    FROG_HIP_CHECK(hipEventRecord(ctx->ord_read[cur], s));
    ctx->sel_used = cur;
    ctx->sel_consumed++;
    // More selections for the ring once it runs low (ctx.h SEL_LOW_WATER), never into the buffer this refresh consumed (it
    // stays readable: getters): a short ring (FROG_SELECT_RING, or a group whose buffers would not fit) produces one per
    // refresh as before, a long one never during the schedules it was sized for.
    if (ctx->sel_produced - ctx->sel_consumed < (uint64_t)std::min((int)frog_ctx::SEL_LOW_WATER, ctx->sel_ring - 1))
        while (ctx->sel_produced - ctx->sel_consumed < (uint64_t)(ctx->sel_ring - 1)) {
            const int rc = produce_selection(ctx);
            if (rc) return rc;
            if (ctx->sel_ring > 8) break;           // a long ring refills one replay per refresh, spread over the iterations
        }
    // rows of other ranks' images: zero, so that an all-reduce(sum) completes the table
    if (ctx->ib > 0) FROG_HIP_CHECK(hipMemsetAsync(ctx->em.p, 0, (size_t)ctx->ib * sizeof(float4), s));
    if (ctx->ie < ctx->nI)
        FROG_HIP_CHECK(hipMemsetAsync(ctx->em.p + ctx->ie, 0, (size_t)(ctx->nI - ctx->ie) * sizeof(float4), s));
    return FROG_OK;
}

int frog_stats_publish(frog_ctx *ctx)
{
    CTX_GUARD(ctx);
    // the weight constants of the new mixtures and, with them, the certified outlier cutoffs (k_cull.hip.h); the check before
    // the next sweep compares the cutoffs with the list's
    stats_publish_kernel<<<dim3(div_up(ctx->nI, 64), 2), 64, 0, ctx->stream>>>(ctx->em.p, ctx->emd.p, ctx->emf.p, ctx->nI, ctx->opt.inlier_threshold, ctx->fast_theta(), ctx->cut_now.p,
                                                                       ctx->deformable ? 0 : 1);
    // Linear stage: the mixtures tighten from refresh to refresh as the images come together, and with them the distance
    // from which a weight is exactly zero -- a list built for the old cutoffs stays VALID but holds links it no longer
    // needs to.  A new list per refresh (one sweep in ten writes it as it goes: +0.15 ms).  The deformable stage keeps its
    // list while it is valid: its mixtures hardly move.
    if (!ctx->deformable) ctx->cull_need_build = true;
    ctx->cull_check_due = true;         // the next cull_prepare recomputes the allowed displacement and runs the stand-alone check
    FROG_HIP_CHECK(hipGetLastError());
    return FROG_OK;
}

int frog_update_stats(frog_ctx *ctx)
{
    if (ctx && !ctx->whole_group())
        return fail(FROG_E_STATE, "context owns a sub-range of images: use frog_update_stats_local + all-reduce + frog_stats_publish");
    int rc = frog_update_stats_local(ctx);
    if (rc) return rc;
    if (ctx->helper) {
        // the fixed images' mixtures: their half-links see the moving points where they are now
        frog_ctx *h = ctx->helper;
        const size_t n = ctx->own_pt_end - ctx->own_pt_begin;
        FROG_HIP_CHECK(hipMemcpyAsync(h->pos2.p + ctx->own_pt_begin, ctx->pos2.p + ctx->own_pt_begin, n * sizeof(P3),
                                      hipMemcpyDeviceToDevice, ctx->stream));
        rc = frog_update_stats_local(h);
        if (rc) return rc;
        FROG_HIP_CHECK(hipMemcpyAsync(ctx->em.p, h->em.p, (size_t)ctx->nf * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    }
    return frog_stats_publish(ctx);
}

// ---- RANSAC + RANSACBatch (imageGroup.cxx:629-804) -----------------------------------------
int frog_ransac(frog_ctx *ctx, const frog_model *m, uint32_t image, const frog_ransac_options *o, int64_t *n_inliers)
{
    CTX_GUARD(ctx);
    if (!m || !o) return fail(FROG_E_INVALID, "null argument");
    if (image < ctx->ib || image >= ctx->ie) return fail(FROG_E_INVALID, "image not owned by this context");
    if (m->n_images != ctx->nI || m->point_offset[ctx->nI] != ctx->P) return fail(FROG_E_INVALID, "model does not match the context");
    if (ctx->deformable) return fail(FROG_E_STATE, "RANSAC after deformable set-up");
    if (o->iterations < 0 || o->batches < 1) return fail(FROG_E_INVALID, "bad RANSAC options");
    hipStream_t s = ctx->stream;
    const uint32_t pb = ctx->poff[image], nPoints = ctx->poff[image + 1] - pb;
    const int perBatch = o->iterations / o->batches;                         // :636
    const float maxDistance2 = (float)std::pow((double)o->inlier_distance, 2);   // :677,:730
    auto xyz_of = [&](uint16_t img, uint32_t pt) { return m->xyz + 3 * ((size_t)m->point_offset[img] + pt); };

    // candidates: 4 random correspondences each (:743-760).  A draw that lands on a point without
    // links is repeated, so an image without any link would never return: refuse it.
    if (nPoints == 0 || m->row_ptr[pb + nPoints] == m->row_ptr[pb]) return fail(FROG_E_INVALID, "image has no link");
    const uint32_t nCand = (uint32_t)perBatch * (uint32_t)o->batches;
    std::vector<double> cand((size_t)nCand * 12);   // 12 doubles per candidate (rows 0..2)
    std::vector<uint8_t> usable(nCand, 0);        // determinant inside [1/maxScale, maxScale] and a defined rotation
    #pragma omp parallel for schedule(dynamic, 1) num_threads(host_threads())    // batches are independent streams, as upstream's threads (:639-647)
    for (int batch = 0; batch < o->batches; batch++) {
        std::mt19937 rng((uint32_t)(batch * 1000));
        for (int it = 0; it < perBatch; it++) {
            float src[4][3], tgt[4][3];
            for (int j = 0; j < 4; j++) {
                while (true) {
                    const uint32_t pt = (uint32_t)(rng() % nPoints);
                    const uint64_t r0 = m->row_ptr[pb + pt], size = m->row_ptr[pb + pt + 1] - r0;
                    if (size == 0) continue;
                    const uint64_t l = r0 + rng() % size;
                    std::memcpy(src[j], m->xyz + 3 * ((size_t)pb + pt), sizeof src[j]);
                    std::memcpy(tgt[j], xyz_of(m->link_image[l], m->link_point[l]), sizeof tgt[j]);
                    break;
                }
            }
            double M[16];
            const bool ok = frog::similarity_fit(4, [&](size_t i, double a[3], double b[3]) {
                for (int k = 0; k < 3; k++) { a[k] = src[i][k]; b[k] = tgt[i][k]; }
            }, M);
            const float determinant = (float)std::fabs(frog::det3_of_4x4(M));           // :787-788
            const size_t c = (size_t)batch * perBatch + it;
            usable[c] = ok && !((determinant > o->max_scale) || ((double)determinant < 1.0 / (double)o->max_scale));
            std::memcpy(&cand[c * 12], M, 12 * sizeof(double));
        }
    }
    std::vector<unsigned int> counts(nCand, 0u);
    const uint32_t lpb = pb - ctx->own_pt_begin, lpe = lpb + nPoints;
    const uint64_t nLinks = m->row_ptr[pb + nPoints] - m->row_ptr[pb];
    if (nCand) {
        DevBuf<double> d_cand;
        DevBuf<unsigned int> d_counts;
        FROG_HIP_CHECK(d_cand.upload(cand, s));
        FROG_HIP_CHECK(d_counts.alloc(nCand));
        FROG_HIP_CHECK(hipMemsetAsync(d_counts.p, 0, d_counts.bytes(), s));
        const unsigned blocks = div_up(nLinks, (size_t)256 * RANSAC_LINKS);
        ransac_count_kernel<<<blocks, 256, 0, s>>>(ctx->pos.p, ctx->pos2.p, ctx->ref_rowptr.p, ctx->ref_link.p, ctx->new_of_old.p,
                                                  lpb, lpe, d_cand.p, nCand, maxDistance2, d_counts.p);
        FROG_HIP_CHECK(hipGetLastError());
        FROG_HIP_CHECK(hipMemcpyAsync(counts.data(), d_counts.p, d_counts.bytes(), hipMemcpyDeviceToHost, s));
        FROG_HIP_CHECK(hipStreamSynchronize(s));
    }
    // best of each batch (first strictly larger count, :790-795), then best batch (:651-664; upstream takes the
    // batches in the order their threads finished -- here in batch order, so ties resolve the same way every run)
    long long maxNumberOfInliers = 0;
    int bestCand = -1;
    for (int batch = 0; batch < o->batches; batch++) {
        long long batchMax = 0;
        int batchBest = -1;
        for (int it = 0; it < perBatch; it++) {
            const int c = batch * perBatch + it;
            if (!usable[c]) continue;
            if (batchMax < (long long)counts[c]) { batchMax = counts[c]; batchBest = c; }
        }
        if (batchMax > maxNumberOfInliers) { maxNumberOfInliers = batchMax; bestCand = batchBest; }
    }
    double best[16];
    if (bestCand >= 0) {
        std::memcpy(best, &cand[(size_t)bestCand * 12], 12 * sizeof(double));
        best[12] = best[13] = best[14] = 0.0; best[15] = 1.0;
    } else {
        FROG_HIP_CHECK(hipMemcpyAsync(best, ctx->mat.p + (size_t)image * 16, sizeof best, hipMemcpyDeviceToHost, s));
        FROG_HIP_CHECK(hipStreamSynchronize(s));
    }
    // refit on every half-link the best candidate brings within the distance (:666-700)
    std::vector<float> xyz2(3 * ctx->P);
    int rc = download_points(ctx, ctx->pos2.p, xyz2.data());
    if (rc) return rc;
    std::vector<const float *> srcs, tgts;
    for (uint32_t pt = 0; pt < nPoints; pt++) {
        const float *in = m->xyz + 3 * ((size_t)pb + pt);
        const float t[3] = { (float)(best[0] * in[0] + best[1] * in[1] + best[2] * in[2] + best[3]),
                             (float)(best[4] * in[0] + best[5] * in[1] + best[6] * in[2] + best[7]),
                             (float)(best[8] * in[0] + best[9] * in[1] + best[10] * in[2] + best[11]) };
        for (uint64_t l = m->row_ptr[pb + pt]; l < m->row_ptr[pb + pt + 1]; l++) {
            const float *pB = &xyz2[3 * ((size_t)m->point_offset[m->link_image[l]] + m->link_point[l])];
            const float dx = t[0] - pB[0], dy = t[1] - pB[1], dz = t[2] - pB[2];
            if (dx * dx + dy * dy + dz * dz < maxDistance2) { srcs.push_back(in); tgts.push_back(pB); }
        }
    }
    double fit[16];
    if (!frog::similarity_fit(srcs.size(), [&](size_t i, double a[3], double b[3]) {
            for (int k = 0; k < 3; k++) { a[k] = srcs[i][k]; b[k] = tgts[i][k]; }
        }, fit))
        std::memcpy(fit, best, sizeof fit);      // rotation undetermined (collinear inliers): keep the candidate
    FROG_HIP_CHECK(hipMemcpyAsync(ctx->mat.p + (size_t)image * 16, fit, sizeof fit, hipMemcpyHostToDevice, s));
    FROG_HIP_CHECK(hipStreamSynchronize(s));
    if (n_inliers) *n_inliers = maxNumberOfInliers;
    return FROG_OK;
}

// ---- updateLinearTransforms (imageGroup.cxx:1063-1149) ----------------------------------
int frog_linear_step_local(frog_ctx *ctx)
{
    CTX_GUARD(ctx);
    if (ctx->deformable) return fail(FROG_E_STATE, "linear step after deformable set-up");
    if (ctx->ref_order) { ctx->xyz2_fresh = false; return ref_linear_step(ctx); }
    hipStream_t s = ctx->stream;
    // half-links whose weight is exactly zero are left to a list, as the deformable stage's certain outliers are (k_cull.hip.h)
    bool culled = cull_active_linear(ctx);
    ctx->build_in_sweep = false;
    if (culled) {
        Span span(ctx, FROG_K_CULL);
        const int rc = cull_prepare(ctx);
        if (rc) return rc;
        culled = cull_active_linear(ctx);   // no room for the list's buffers: the context goes on without one
    }
    {
        Span span(ctx, ctx->build_in_sweep ? FROG_K_SWEEP_LINEAR_BUILD : FROG_K_SWEEP_LINEAR, ctx->n_sub == 1);
        for (uint32_t sub = 0; sub < ctx->n_sub; sub++)
            launch_sweep<SWEEP_LINEAR>(ctx, sub, s, span.attached ? span.a : nullptr, span.attached ? span.b : nullptr, culled,
                                       ctx->build_in_sweep);
    }
    FROG_HIP_CHECK(hipGetLastError());
    // the per-image update; its blocks also leave the image's energy terms, which the last of them adds up in image order
    // (energy_reduce_kernel, a launch of its own over the tile partials, until round 3)
    linear_update_kernel<<<ctx->n_owned(), 256, 0, s>>>(ctx->tile_partial.p, ctx->img_tile_ptr.p, ctx->n_groups, ctx->ib, ctx->mat.p,
                                                      ctx->opt.linear_alpha, ctx->opt.use_scale, ctx->img_energy.p, ctx->energy_ticket.p + 1,
                                                      ctx->energy.p, culled ? ctx->cull_state.p : nullptr);
    FROG_HIP_CHECK(hipGetLastError());
    ctx->xyz2_fresh = false;
    return FROG_OK;
}

int frog_energy_read(frog_ctx *ctx, double *E, double *n_oversize)
{
    CTX_GUARD(ctx);
    FROG_HIP_CHECK(hipMemcpyAsync(ctx->h_energy, ctx->energy.p, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->h_energy[3] > 0) ctx->cull_need_build = true;       // the sweep found its list out of date (it walked every record)
    if (E) *E = std::sqrt(ctx->h_energy[0] / ctx->h_energy[1]);
    if (n_oversize) *n_oversize = ctx->h_energy[2];
    return FROG_OK;
}

int frog_linear_step(frog_ctx *ctx, double *E)
{
    int rc = frog_linear_step_local(ctx);
    if (rc) return rc;
    // A context that owns the whole group: the transformPoints() that run() calls next (imageGroup.cxx:63) is queued here,
    // into the shadow buffer, and its first thread hands the step's scalars to the host through pinned memory -- the host used
    // to wait for a copy of the scalars, return, and only then queue the transform: 20 us of idle GPU per linear iteration.
    const uint32_t n = ctx->own_pt_end - ctx->own_pt_begin;
    if (!(ctx->whole_group() && ctx->h_energy_dev && n && !ctx->helper && !getenv("FROG_LINEAR_NO_SPECULATION")))
        return frog_energy_read(ctx, E, nullptr);
    if (ctx->pos2_spec.n != ctx->P) FROG_HIP_CHECK(ctx->pos2_spec.alloc(ctx->P));
    const double seq = (double)(++ctx->scalar_seq);
    rc = launch_transform(ctx, ctx->pos2_spec.p, 0, false, seq);
    if (rc) return rc;
    ctx->xyz2_fresh = true;
    rc = wait_step_scalars(ctx, seq);
    if (rc) return rc;
    if (ctx->h_energy[3] > 0) ctx->cull_need_build = true;       // the sweep found its list out of date (it walked every record)
    if (E) *E = std::sqrt(ctx->h_energy[0] / ctx->h_energy[1]);
    return FROG_OK;
}

// ---- setupDeformableTransforms (imageGroup.cxx:159-218) ----------------------------------
int frog_bounds_local(frog_ctx *ctx, double mins[3], double maxs[3])
{
    CTX_GUARD(ctx);
    hipStream_t s = ctx->stream;
    bounds_kernel<<<BOUNDS_BLOCKS, 256, 0, s>>>(ctx->pos.p, ctx->own_pt_begin, ctx->own_pt_end, ctx->bounds_scratch.p);
    bounds_final_kernel<<<1, 256, 0, s>>>(ctx->bounds_scratch.p, BOUNDS_BLOCKS, ctx->bounds_scratch.p + (size_t)BOUNDS_BLOCKS * 6);
    FROG_HIP_CHECK(hipGetLastError());
    float *h = reinterpret_cast<float *>(ctx->h_energy + 4);       // pinned
    FROG_HIP_CHECK(hipMemcpyAsync(h, ctx->bounds_scratch.p + (size_t)BOUNDS_BLOCKS * 6, 6 * sizeof(float), hipMemcpyDeviceToHost, s));
    FROG_HIP_CHECK(hipStreamSynchronize(s));
    // an empty range leaves (+FLT_MAX, -FLT_MAX): neutral in the callers' min / max reduction over ranks
    for (int k = 0; k < 3; k++) { mins[k] = (double)h[k]; maxs[k] = (double)h[3 + k]; }
    return FROG_OK;
}


// box.ScaleAboutCenter(1 + 2*margin) then the lattice (imageGroup.cxx:161-179), and the brick partition of its cells
static int make_geometry(const frog_ctx *ctx, int level, const double mins[3], const double maxs[3], GridGeom &g, frog_grid_info &info)
{
    const double size = (double)ctx->opt.initial_grid_size / std::pow(2, level);
    const float sc = 1 + 2 * ctx->opt.bounding_box_margin;
    for (int k = 0; k < 3; k++) {
        const double cen = 0.5 * (mins[k] + maxs[k]);
        const double lo = cen + (double)sc * (mins[k] - cen);
        const double hi = cen + (double)sc * (maxs[k] - cen);
        const double length = hi - lo;
        int d = (int)std::round(length / size);
        if (d < 1) d = 1;
        g.spacing[k] = length / d;
        g.origin[k] = lo - g.spacing[k];
        g.cells[k] = d;
        g.dims[k] = d + 3;
        info.bbox[2 * k] = lo; info.bbox[2 * k + 1] = hi;
        info.dims[k] = g.dims[k]; info.origin[k] = g.origin[k]; info.spacing[k] = g.spacing[k];
    }
    const size_t G = (size_t)g.dims[0] * g.dims[1] * g.dims[2];
    if (G > 0x7FFFFFFFull) return fail(FROG_E_INVALID, "lattice too large");
    g.n_cp = (int)G;
    const uint32_t nO = ctx->n_owned();
    const uint32_t nPts = ctx->own_pt_end - ctx->own_pt_begin;
    auto bricks_for = [&](int B) { size_t n = 1; for (int k = 0; k < 3; k++) n *= (size_t)((g.cells[k] + B - 1) / B); return n; };
    // brick edge: 4 cells (5.5 KB tile, many resident wavefronts) while bricks keep enough points to
    // amortise their flush, else 8
    g.brick = ((double)nPts / ((double)nO * (double)bricks_for(4)) >= 24.0) ? 4 : 8;
    if (const char *e = getenv("FROG_BRICK")) { const int b = atoi(e); if (b == 4 || b == 8) g.brick = b; }      // test hook
    for (int k = 0; k < 3; k++) g.nbricks[k] = (g.cells[k] + g.brick - 1) / g.brick;
    const size_t nb = bricks_for(g.brick);
    if (nb * nO >= 0x7FFFFFFFull) return fail(FROG_E_INVALID, "too many bricks");
    g.n_bricks = (int)nb;
    if ((size_t)nO * nb * (size_t)(g.brick * g.brick * g.brick) >= 0x7FFFFFFFull) return fail(FROG_E_INVALID, "too many lattice cells");
    // Lattice layout and storage of the finest lattices of many images (ctx.h GridGeom::blocked / sparse, k_grid.hip.h lat()):
    //   blocked: blocks of 16 nodes across the owned images instead of image-major (the lattice step works through ALL images of
    //            16 nodes at a time);
    //   sparse:  entries only for the (image, node) pairs some point of the image reaches -- the others of a node share one value.
    // Measured on cfg 5's level 4 (500 images x 9e5 nodes, `scripts/cfg5_level4_traffic.sh`): the
    // lattice step 8.09 -> 6.7 ms with 41 -> 28 GB moved; the B-spline transform + 3 %; the set of active pairs costs 1.9 ms per
    // lattice to find; bench.py --config 5: 168.5 -> 175 it/s.  Lattices of >= 2^27 pairs; smaller ones (cfg 5's level 3: 7e7 pairs;
    // everything of cfg 3) keep the plain form, where the step is bound by latency, not bytes, and both forms cost a few per cent.
    // FROG_LATTICE_BLOCKED / FROG_LATTICE_SPARSE = 0 / 1 force a form (A/B, tests); reference-order mode keeps the plain one.
    g.lat_images = nO;
    {
        static const char *e = getenv("FROG_LATTICE_BLOCKED");
        static const char *es = getenv("FROG_LATTICE_SPARSE");
        const bool fine = nO >= 32u && (size_t)nO * G >= ((size_t)1 << 27);
        g.blocked = e ? atoi(e) != 0 : fine;
        g.sparse = es ? atoi(es) != 0 : fine;
        if (ctx->ref_order) g.blocked = g.sparse = false;
        g.mask_words = (uint32_t)((G + 31) / 32) + 1u;          // per image: a bit per node (+ 1: a tile row may be written through the word behind its last bit)
    }
    return FROG_OK;
}

// Every buffer whose size follows the lattice.  Allocated with head-room for two more levels (8x the control points
// each) whenever a new block is needed, up to a cap: hipFree / hipMalloc synchronise the device and cost a millisecond
// per level.  frog_create calls this with the lattice the model's own bounding box would give at level 0, so that a
// run's first set-up finds its buffers (and with them those of levels 1 and 2) already there.
static int lattice_alloc(frog_ctx *ctx, const GridGeom &g)
{
    const size_t G = (size_t)g.n_cp;
    const uint32_t nO = ctx->n_owned();
    const uint32_t nPts = ctx->own_pt_end - ctx->own_pt_begin;
    const size_t reserve = ((size_t)nO * G * 64 * sizeof(float4) <= ((size_t)2 << 30)) ? 64 : (((size_t)nO * G * 8 * sizeof(float4) <= ((size_t)2 << 30)) ? 8 : 1);
    const size_t n_keys = (size_t)nO * g.n_bricks * (size_t)(g.brick * g.brick * g.brick);
    const size_t n_bricks_total = (size_t)nO * g.n_bricks;
    const size_t max_blocks = std::max<size_t>(1, scatter_max_blocks((uint32_t)n_bricks_total, nPts, ctx->scatter_chunk));
    const size_t E = (size_t)g.brick + 3;
    const size_t LG = std::max(g.lat_entries(), (size_t)nO * G);       // entries of one lattice in its layout (blocked: nodes padded to 16)
    if (getenv("FROG_SETUP_TRACE"))
        std::fprintf(stderr, "[lattice_alloc] %zu entries per lattice (capacity now %zu), %zu keys (capacity %zu), %zu blocks, mask %zu words (capacity %zu)\n",
                     LG, ctx->coeff.cap, n_keys, ctx->key_counts.cap, max_blocks, g.sparse ? (size_t)nO * g.mask_words : (size_t)0, ctx->lat_mask.cap);
    if (ctx->created && (LG > ctx->coeff.cap || LG > ctx->grad.cap || LG > ctx->gradf.cap || n_keys > ctx->key_counts.cap)) ctx->lattice_reallocs++;
    FROG_HIP_CHECK(ctx->coeff.alloc(LG, LG * reserve));
    FROG_HIP_CHECK(ctx->grad.alloc(LG, LG * reserve));
    FROG_HIP_CHECK(ctx->gradf.alloc(LG, LG * reserve));
    // a third lattice for contexts whose host queues the next step before this one's decision is known (frog_step_speculate)
    // (every context that owns a sub-range, whichever flow its host will choose: frog_create sizes the head-room before any host
    // can call frog_comm_mode -- ADVICE r5 -- and one lattice more per rank is cheap: cfg 5, rank of eight, 0.95 GB)
    if (!ctx->whole_group()) FROG_HIP_CHECK(ctx->grad_spare.alloc(LG, LG * reserve));
    if (g.sparse) {
        FROG_HIP_CHECK(ctx->lat_mask.alloc((size_t)nO * g.mask_words, (size_t)nO * g.mask_words * reserve));
        FROG_HIP_CHECK(ctx->lat_inactive.alloc(G, G * reserve));
        FROG_HIP_CHECK(ctx->ucoeff.alloc(G, G * reserve)); FROG_HIP_CHECK(ctx->ugrad.alloc(G, G * reserve));
        FROG_HIP_CHECK(ctx->ugrad_spare.alloc(G, G * reserve));
    }
    FROG_HIP_CHECK(ctx->gridsum.alloc(3 * G + 4, (3 * G + 4) * reserve));      // + 4: the energy sums' seat on the all-reduce (frog_comm_mode)
    FROG_HIP_CHECK(ctx->key_counts.alloc(n_keys, n_keys * reserve));
    FROG_HIP_CHECK(ctx->brick_ptr_scratch.alloc(n_bricks_total + 1, (n_bricks_total + 1) * reserve));
    FROG_HIP_CHECK(ctx->key_ptr.alloc(n_keys + 1, (n_keys + 1) * reserve));
    FROG_HIP_CHECK(ctx->key_cursor.alloc(n_keys + 1, (n_keys + 1) * reserve));
    FROG_HIP_CHECK(ctx->scan_sums.alloc(div_up(n_keys, SCAN_BLOCK_ITEMS) + 2, (div_up(n_keys, SCAN_BLOCK_ITEMS) + 2) * reserve));
    if (ctx->perm.n != nPts) FROG_HIP_CHECK(ctx->perm.alloc(nPts));
    if (ctx->perm_tmp.n != nPts) FROG_HIP_CHECK(ctx->perm_tmp.alloc(nPts));
    if (ctx->pos_b.n != nPts) FROG_HIP_CHECK(ctx->pos_b.alloc(nPts));
    if (ctx->perm_key.n != nPts) FROG_HIP_CHECK(ctx->perm_key.alloc(nPts));
    if (ctx->pos2_spec.n != ctx->P) FROG_HIP_CHECK(ctx->pos2_spec.alloc(ctx->P));
    FROG_HIP_CHECK(ctx->brick_slot_ptr.alloc(n_bricks_total + 1, (n_bricks_total + 1) * reserve));
    FROG_HIP_CHECK(ctx->scatter_blocks.alloc(max_blocks * sizeof(ScatterBlock), max_blocks * sizeof(ScatterBlock) * reserve));
    FROG_HIP_CHECK(ctx->scatter_blocks_tmp.alloc(max_blocks * sizeof(ScatterBlock), max_blocks * sizeof(ScatterBlock) * reserve));
    FROG_HIP_CHECK(ctx->len_hist.alloc(2 * (SCATTER_CHUNK + 1)));
    // tile storage: a finer level needs about as many blocks (bricks hold fewer points) but brick edge 8 instead of 4 has
    // 2.4x the tile
    FROG_HIP_CHECK(ctx->scatter_stage.alloc(max_blocks * E * E * E, max_blocks * E * E * E * std::min<size_t>(reserve, 8)));
    return FROG_OK;
}

static int retire_current_grid(frog_ctx *ctx)
{
    if (ctx->grids.empty() || ctx->grids.back().retired) return FROG_OK;
    GridRecord &gr = ctx->grids.back();
    const size_t n = ctx->geom.lat_entries();           // in the layout it stands in (frog_get_grid extracts an image from it)
    gr.blocked = ctx->geom.blocked; gr.lat_images = ctx->geom.lat_images;
    // the finished lattice stays on the device (a copy on the stream: no host round trip inside the
    // regrid path; 43 MB per lattice at level 2 of the 100-image group); frog_get_grid reads it back on demand
    gr.kept = std::make_shared<DevBuf<float4>>();
    const size_t room = (n + 63) / 64 * 64;
    if (n && ctx->retired_arena.p && ctx->retired_used + room <= ctx->retired_arena.cap) {
        gr.kept->borrow(ctx->retired_arena.p + ctx->retired_used, n);
        ctx->retired_used += room;
    } else {
        FROG_HIP_CHECK(gr.kept->alloc_async(std::max<size_t>(1, n), ctx->stream));
    }
    if (n) FROG_HIP_CHECK(hipMemcpyAsync(gr.kept->p, ctx->coeff.p, n * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    gr.sparse = ctx->geom.sparse; gr.mask_words = ctx->geom.mask_words;
    if (gr.sparse) {                            // which pairs `kept` holds, and what the others are: in the arena behind it when there is room
        const size_t G = (size_t)ctx->geom.n_cp;
        const size_t mask_words = (size_t)ctx->n_owned() * gr.mask_words;
        const size_t mask_room = ((mask_words + 3) / 4 + 63) / 64 * 64, u_room = (G + 63) / 64 * 64;      // in float4 entries
        gr.kept_mask = std::make_shared<DevBuf<uint32_t>>();
        gr.kept_u = std::make_shared<DevBuf<float4>>();
        if (ctx->retired_arena.p && ctx->retired_used + mask_room + u_room <= ctx->retired_arena.cap) {
            gr.kept_mask->borrow(reinterpret_cast<uint32_t *>(ctx->retired_arena.p + ctx->retired_used), mask_words);
            gr.kept_u->borrow(ctx->retired_arena.p + ctx->retired_used + mask_room, G);
            ctx->retired_used += mask_room + u_room;
        } else {
            FROG_HIP_CHECK(gr.kept_mask->alloc_async(std::max<size_t>(1, mask_words), ctx->stream));
            FROG_HIP_CHECK(gr.kept_u->alloc_async(std::max<size_t>(1, G), ctx->stream));
        }
        FROG_HIP_CHECK(hipMemcpyAsync(gr.kept_mask->p, ctx->lat_mask.p, mask_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
        FROG_HIP_CHECK(hipMemcpyAsync(gr.kept_u->p, ctx->ucoeff.p, G * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    }
    gr.retired = true;
    return FROG_OK;
}

int frog_deformable_setup_bounds(frog_ctx *ctx, int level, const double mins[3], const double maxs[3], frog_grid_info *out)
{
    CTX_GUARD(ctx);
    if (level < 0 || level > 30) return fail(FROG_E_INVALID, "bad level");
    ctx->xyz2_fresh = false; ctx->res_valid = false; ctx->rc_valid = false;
    // FROG_SETUP_TRACE=1: where the host's time in this call goes (a level's first lattice has been seen to take 0.7 s of it on
    // some boxes and 0.016 s on others)
    static const bool setup_trace = getenv("FROG_SETUP_TRACE") != nullptr;
    const auto t_setup = std::chrono::steady_clock::now();
    auto setup_lap = [&](const char *what) {
        if (setup_trace) std::fprintf(stderr, "[setup level %d] %-28s %9.3f ms\n", level, what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_setup).count());
    };
    int rc = join_setup(ctx);                   // a set-up right behind a set-up
    if (rc) return rc;
    setup_lap("joined the last set-up");
    rc = retire_current_grid(ctx);
    if (rc) return rc;
    setup_lap("+ finished lattice retired");

    GridGeom g{};
    frog_grid_info info{};
    rc = make_geometry(ctx, level, mins, maxs, g, info);
    if (rc) return rc;
    const uint32_t nO = ctx->n_owned();
    const uint32_t nPts = ctx->own_pt_end - ctx->own_pt_begin;
    const size_t nb = (size_t)g.n_bricks;
    ctx->geom = g;
    info.n_grid = (int)ctx->grids.size();

    rc = lattice_alloc(ctx, g);                 // within the head-room reserved earlier, as a rule: no hipMalloc here
    if (rc) return rc;
    setup_lap("+ lattice buffers");
    // sizes the kernels below rely on, checked here where an error can still be returned to the caller
    const uint32_t keys_per_brick = (uint32_t)(g.brick * g.brick * g.brick);
    const size_t n_keys64 = (size_t)nO * nb * keys_per_brick;
    if (n_keys64 >= 0x7FFFFFFFull) return fail(FROG_E_INVALID, "too many lattice cells");
    ctx->n_scatter_blocks = scatter_max_blocks(nO * (uint32_t)nb, nPts, ctx->scatter_chunk);
    ctx->coeff_zero = true;
    // The device work of the set-up -- zeroing, the sort of the points, the block table -- is queued on the set-up stream,
    // behind what `stream` holds NOW (the re-based coordinates, the copy of the finished lattice, the last readers of the old
    // sort: the fork event), but not yet: the host queues it when the first consumer asks (join_setup: as a rule the step's
    // scatter, right after its half-link sweep has been launched), so that the transform, the statistics refresh and the
    // sweep that open the level are in the queue first and the set-up runs beside them instead of in front of them.
    if (ctx->setup_async && ctx->setup_stream) {
        FROG_HIP_CHECK(hipEventRecord(ctx->setup_fork, ctx->stream));
        ctx->setup_deferred = true;
    } else {
        rc = queue_setup_kernels(ctx, ctx->stream);
        if (rc) return rc;
    }

    if (!ctx->deformable) {
        // end of the linear stage: its list (if any) is counted for frog_cull_stats_linear, and the deformable stage starts
        // with cutoffs of its own criterion and a list of its own
        if (ctx->cull_lin_builds && ctx->act_cnt.p) {
            FROG_HIP_CHECK(hipMemsetAsync(ctx->lin_listed.p, 0, sizeof(unsigned long long), ctx->stream));
            cull_count_kernel<<<CULL_COUNT_BLOCKS, 256, 0, ctx->stream>>>(ctx->act_cnt.p, (uint32_t)ctx->act_cnt.n, ctx->lin_listed.p);
        }
        stats_publish_kernel<<<dim3(div_up(ctx->nI, 64), 2), 64, 0, ctx->stream>>>(ctx->em.p, ctx->emd.p, ctx->emf.p, ctx->nI, ctx->opt.inlier_threshold, ctx->fast_theta(), ctx->cut_now.p, 0);
        FROG_HIP_CHECK(hipGetLastError());
        ctx->cull_need_build = true;
        ctx->cull_check_due = true;
    }
    setup_lap("+ kernels queued or deferred");
    GridRecord rec;
    rec.info = info;
    ctx->grids.push_back(std::move(rec));
    ctx->deformable = true;
    ctx->phase = 0;
    if (out) *out = info;
    return FROG_OK;
}

int frog_deformable_setup(frog_ctx *ctx, int level, frog_grid_info *out)
{
    if (ctx && !ctx->whole_group())
        return fail(FROG_E_STATE, "context owns a sub-range of images: use frog_bounds_local + all-reduce + frog_deformable_setup_bounds");
    double mn[3], mx[3];
    int rc = frog_bounds_local(ctx, mn, mx);
    if (rc) return rc;
    return frog_deformable_setup_bounds(ctx, level, mn, mx, out);
}

// buffers of the culling list (4 bytes per half-link for the listed records); at frog_create, so that the first deformable
// step does not pay for them
static int cull_allocate_buffers(frog_ctx *ctx);

// The list is an optimisation: when its buffers do not fit (4-8 bytes per half-link, a second coordinate table) the context
// runs without one -- every sweep walks all records -- instead of failing.
static int cull_allocate(frog_ctx *ctx)
{
    if (ctx->act_cnt.p || !ctx->cull_enabled) return FROG_OK;
    const int rc = cull_allocate_buffers(ctx);
    if (rc == FROG_OK) return rc;
    (void)hipGetLastError();
    ctx->act_recs32.release(); ctx->act_recs.release(); ctx->act_cnt.release(); ctx->pos2_snap.release();
    ctx->cut_list.release(); ctx->disp_part.release(); ctx->cull_state.release(); ctx->disp_allow.release();
    ctx->cull_enabled = false;
    ctx->fused_sweep = ctx->fused_sweep && ctx->fused_forced;
    return FROG_OK;
}

static int cull_allocate_buffers(frog_ctx *ctx)
{
    hipStream_t s = ctx->stream;
    if (ctx->rec_format.narrow) {
        FROG_HIP_CHECK(ctx->act_recs32.alloc(ctx->L_recs));
        FROG_HIP_CHECK(hipMemsetAsync(ctx->act_recs32.p, 0, ctx->act_recs32.bytes(), s));    // null records, as the padding of the full array
    } else {
        FROG_HIP_CHECK(ctx->act_recs.alloc(ctx->L_recs));
        FROG_HIP_CHECK(hipMemsetAsync(ctx->act_recs.p, 0, ctx->act_recs.bytes(), s));
    }
    FROG_HIP_CHECK(ctx->act_cnt.alloc(std::max<size_t>(1, (size_t)ctx->n_tiles * ctx->n_groups)));
    FROG_HIP_CHECK(ctx->pos2_snap.alloc(ctx->P));
    FROG_HIP_CHECK(ctx->cut_list.alloc(ctx->nI));
    {
        uint32_t blocks = 0;
        for (uint32_t i = 0; i < ctx->nI; i++) blocks += div_up(ctx->poff[i + 1] - ctx->poff[i], CULL_BLOCK_POINTS);
        // one slot per producer block: cull_disp_kernel's, or the B-spline transform's (at most one block per point + one per
        // SCATTER_CHUNK points in its tiled form)
        FROG_HIP_CHECK(ctx->disp_part.alloc(std::max<size_t>(blocks, (size_t)ctx->P + ctx->P / ctx->scatter_chunk + 16)));
    }
    FROG_HIP_CHECK(ctx->cull_state.alloc(2));
    FROG_HIP_CHECK(hipMemsetAsync(ctx->cull_state.p, 0, ctx->cull_state.bytes(), s));
    FROG_HIP_CHECK(ctx->disp_allow.alloc(1));
    FROG_HIP_CHECK(hipMemsetAsync(ctx->disp_allow.p, 0xFF, sizeof(float), s));        // NaN until the first list exists
    ctx->cull_need_build = true;
    return FROG_OK;
}

// Certified outlier culling (k_cull.hip.h): (re)build the list when the host knows it is due, then check it against
// the coordinates and mixtures the sweep is about to read; the sweep takes the result from cull_state on the device.
static int cull_prepare(frog_ctx *ctx)
{
    hipStream_t s = ctx->stream;
    const uint32_t nI = ctx->nI;
    int rc = cull_allocate(ctx);
    if (rc) return rc;
    if (!ctx->cull_enabled) return FROG_OK;     // the buffers did not fit
    if (ctx->cull_need_build) {
        cull_list_cutoff_kernel<<<div_up(nI, 64), 64, 0, s>>>(ctx->cut_now.p, nI, ctx->deformable ? ctx->cull_scale : ctx->cull_lin_scale,
                                                              ctx->deformable ? ctx->cull_pad : ctx->cull_lin_pad, ctx->cut_list.p);
        FROG_HIP_CHECK(hipMemcpyAsync(ctx->pos2_snap.p, ctx->pos2.p, ctx->P * sizeof(P3), hipMemcpyDeviceToDevice, s));
        // the list itself: written by the sweep this call prepares (it walks every record anyway: k_links.hip.h BUILD), or,
        // for the record formats that sweep does not cover, by a pass of its own
        ctx->build_in_sweep = sweep_builds_list(ctx) && !getenv("FROG_CULL_BUILD_PASS");
        if (!ctx->build_in_sweep) {
            const SweepArgs args = sweep_args(ctx, 0);
            const dim3 grid(sweep_blocks(ctx), ctx->n_sub);
            if (ctx->rec_format.narrow)
                cull_build_kernel<false><<<grid, 256, 0, s>>>(args, ctx->cut_list.p, ctx->act_recs32.p, ctx->act_cnt.p);
            else
                cull_build_kernel<true><<<grid, 256, 0, s>>>(args, ctx->cut_list.p, ctx->act_recs.p, ctx->act_cnt.p);
            FROG_HIP_CHECK(hipGetLastError());
        }
        ctx->cull_need_build = false;
        if (ctx->deformable) ctx->cull_builds++; else ctx->cull_lin_builds++;
        ctx->disp_n = 0;                    // the points are where the snapshot has them: no displacement to look at
        ctx->disp_own_n = 0;
        ctx->disp_current = true;
        ctx->disp_others = true;            // every row, whoever owns it
        ctx->cull_check_due = true;
    }
    // displacement since the build: already in disp_part when the transform that produced the current xyz2 measured it
    // (launch_transform; whole-group contexts whose xyz2 nobody else writes), else one pass over all points
    const bool measured = ctx->disp_current && ((ctx->whole_group() && !ctx->xyz2_exported) || ctx->disp_others);
    if (!measured) {
        uint32_t max_pts = 1;
        for (uint32_t i = 0; i < nI; i++) max_pts = std::max(max_pts, ctx->poff[i + 1] - ctx->poff[i]);
        const dim3 grid(div_up(max_pts, CULL_BLOCK_POINTS), nI);
        cull_disp_kernel<<<grid, 256, 0, s>>>(ctx->pos2.p, ctx->pos2_snap.p, ctx->d_poff.p, ctx->disp_part.p);
        ctx->disp_n = grid.x * grid.y;
    }
    // The transform that measured the displacement has already compared it with disp_allow and raised cull_state if it
    // had to (k_grid.hip.h); the stand-alone check (which also LOWERS the flag again) runs when the cutoffs or the list
    // have changed since, or when the displacement was measured here.  While the flag is up the host rebuilds anyway.
    // (a sub-range context: its transform compared only its own rows with the allowance, so the stand-alone check runs
    // every step there -- one block over a few thousand maxima instead of a pass over every coordinate)
    if (ctx->cull_check_due || !measured || !ctx->whole_group()) {
        if (ctx->cull_check_due)
            cull_allow_validate_kernel<<<1, 256, 0, s>>>(ctx->cut_now.p, ctx->cut_list.p, ctx->disp_part.p, ctx->disp_n, nI,
                                                         ctx->disp_allow.p, ctx->cull_state.p);
        else
            cull_validate_kernel<<<1, 256, 0, s>>>(ctx->cut_now.p, ctx->cut_list.p, ctx->disp_part.p, ctx->disp_n, nI, ctx->cull_state.p);
        ctx->cull_check_due = false;
    }
    FROG_HIP_CHECK(hipGetLastError());
    return FROG_OK;
}

// ---- updateDeformableTransforms (imageGroup.cxx:234-472) -----------------------------------
int frog_deformable_phase_a(frog_ctx *ctx, float alpha)
{
    CTX_GUARD(ctx);
    if (!ctx->deformable) return fail(FROG_E_STATE, "deformable step before frog_deformable_setup");
    ctx->xyz2_fresh = false; ctx->res_valid = false;
    if (ctx->ref_order) return ref_deformable_phase_a(ctx, alpha);
    hipStream_t s = ctx->stream;
    const GeomDev gd = to_dev(ctx->geom);
    const uint32_t nO = ctx->n_owned();
    // the gradient lattice proper lives in the staged tiles of the scatter; `gradf` only receives stray points (Fill(0), :249)
    bool culled = cull_active(ctx);
    bool fused_energy = false;
    ctx->build_in_sweep = false;
    if (culled) {
        Span span(ctx, FROG_K_CULL);
        int rc = cull_prepare(ctx);
        if (rc) return rc;
        culled = cull_active(ctx);          // no room for the list's buffers: the context goes on without one
    }
    {
        // a launch that also writes the culling list is timed as a group of its own: it is not the steady-state kernel
        Span span(ctx, ctx->build_in_sweep ? FROG_K_SWEEP_BUILD : FROG_K_SWEEP_DEFORMABLE, ctx->n_sub == 1);
        for (uint32_t sub = 0; sub < ctx->n_sub; sub++)
            launch_sweep<SWEEP_DEFORMABLE>(ctx, sub, s, span.attached ? span.a : nullptr, span.attached ? span.b : nullptr, culled,
                                           ctx->build_in_sweep);
    }
    FROG_HIP_CHECK(hipGetLastError());
    { const int rc = join_setup_positions(ctx); if (rc) return rc; }     // the scatter is the first kernel that needs the new lattice's sort
    {
        Span span(ctx, FROG_K_COMBINE);
        // the per-point sums are only materialised when something other than the scatter reads them (landmark
        // constraints here, frog_get_point_sums later): the scatter adds the N_XCD partial sums itself
        static const bool always_combine = getenv("FROG_COMBINE") != nullptr;      // test hook
        const bool fused = sweep_fused_now(ctx, ctx->build_in_sweep && sweep_builds_list(ctx));
        ctx->point_sums_stale = !(ctx->n_hard || always_combine || fused);     // the fused sweep writes the points' sums itself
        if (!ctx->point_sums_stale && !fused)
            combine_groups_kernel<<<div_up(ctx->own_pt_end - ctx->own_pt_begin, 256), 256, 0, s>>>(
                ctx->group_sums.p, ctx->own_pt_end - ctx->own_pt_begin, ctx->own_pt_begin, ctx->point_sums.p);
        // the energy sums: by the first blocks of the scatter's launch (k_grid.hip.h), unless something between here and the
        // scatter needs them (landmark constraints add to them) or there is no scatter to ride on
        ctx->stray_parity ^= 1u;
        fused_energy = ctx->n_scatter_blocks && !ctx->n_hard && !getenv("FROG_ENERGY_PASS");
        if (!fused_energy)
            energy_reduce_kernel<<<ENERGY_BLOCKS, 256, 0, s>>>(ctx->tile_partial.p, ctx->n_tiles * ctx->n_groups, 2, 0, ctx->energy_blocks.p,
                                                               ctx->energy_ticket.p, ctx->energy.p, culled ? ctx->cull_state.p : nullptr,
                                                               ctx->stray.p);
        if (ctx->n_hard) {                                      // landmark constraints, imageGroup.cxx:280-295
            hard_links_kernel<<<div_up(ctx->n_hard, 64), 64, 0, s>>>(ctx->pos2.p, ctx->point_sums.p, ctx->hl_point.p, ctx->hl_ptr.p,
                                                                    ctx->hl_partner.p, ctx->n_hard, ctx->hard_weight2, ctx->hl_partial.p);
            hard_energy_kernel<<<1, 1, 0, s>>>(ctx->hl_partial.p, ctx->n_hard, ctx->energy.p);
        }
    }
    FROG_HIP_CHECK(hipGetLastError());
    if (ctx->n_scatter_blocks) {
        Span span(ctx, FROG_K_SCATTER);
        const size_t tile_bytes = (size_t)(gd.brick + 3) * (gd.brick + 3) * (gd.brick + 3) * sizeof(float4);
        ScatterEnergy en{};
        en.stray_total = ctx->stray.p + 2;
        if (fused_energy) {
            en.partial = ctx->tile_partial.p; en.n = ctx->n_tiles * ctx->n_groups;
            en.block_sums = ctx->energy_blocks.p; en.ticket = ctx->energy_ticket.p; en.energy = ctx->energy.p;
            en.list_invalid = culled ? ctx->cull_state.p : nullptr;
            en.stray_next = ctx->stray.p + (ctx->stray_parity ^ 1u);
        }
        scatter_kernel<<<ctx->n_scatter_blocks + (fused_energy ? ENERGY_BLOCKS : 0), 64, tile_bytes, s>>>(
            ctx->pos_b.p, ctx->point_sums.p, ctx->point_sums_stale ? ctx->group_sums.p : nullptr,
            ctx->own_pt_end - ctx->own_pt_begin, ctx->own_pt_begin, ctx->perm.p,
            reinterpret_cast<const ScatterBlock *>(ctx->scatter_blocks.p), ctx->brick_slot_ptr.p + (size_t)nO * gd.n_bricks,
            ctx->gradf.p, ctx->scatter_stage.p, ctx->stray.p + ctx->stray_parity, gd, en);
        FROG_HIP_CHECK(hipGetLastError());
    }
    {
        // flush of the staged tiles + control-point step + sum over the owned images; for a context that owns the whole
        // group also the mean removal and the oversize count (phase B is then empty): one launch
        Span span(ctx, FROG_K_LATTICE);
        LatticeStepArgs la{};
        la.stage = ctx->scatter_stage.p; la.brick_slot_ptr = ctx->brick_slot_ptr.p;
        la.gradf = ctx->gradf.p; la.stray = ctx->stray.p + ctx->stray_parity;
        la.coeff = ctx->coeff.p; la.grad = ctx->grad.p; la.gridsum = ctx->gridsum.p;
        la.mask = ctx->lat_mask.p; la.n_inactive = ctx->lat_inactive.p; la.ucoeff = ctx->ucoeff.p; la.ugrad = ctx->ugrad.p;
        la.energy_tail = ctx->two_collectives && !ctx->whole_group() ? ctx->gridsum.p + 3 * (size_t)gd.n_cp : nullptr;
        la.n_owned = nO; la.n_images = ctx->nf ? 0u : ctx->nI; la.alpha = alpha;          // :398: no mean removal with fixed images
        for (int k = 0; k < 3; k++) la.lim[k] = (double)ctx->opt.max_displacement_ratio * ctx->geom.spacing[k];
        la.energy = ctx->energy.p;
        ctx->centered_in_a = ctx->whole_group();
        // a coarse lattice takes the narrow block shape: four times the blocks (k_grid.hip.h LS_CPB_SMALL)
        static const int force_shape = [] { const char *e = getenv("FROG_LS_SHAPE"); return e ? atoi(e) : 0; }();      // 16 / 4: A/B
        const bool narrow = force_shape ? force_shape == LS_CPB_SMALL : gd.n_cp < LS_SMALL_NODES;
        if (narrow) {
            const dim3 lgrid(div_up(gd.n_cp, LS_CPB_SMALL));
            if (ctx->centered_in_a) lattice_step_kernel<true, LS_CPB_SMALL><<<lgrid, LS_CPB_SMALL * LS_IC, 0, s>>>(la, gd);
            else lattice_step_kernel<false, LS_CPB_SMALL><<<lgrid, LS_CPB_SMALL * LS_IC, 0, s>>>(la, gd);
        } else {
            if (ctx->centered_in_a) lattice_step_kernel<true><<<div_up(gd.n_cp, LS_CPB), LS_THREADS, 0, s>>>(la, gd);
            else lattice_step_kernel<false><<<div_up(gd.n_cp, LS_CPB), LS_THREADS, 0, s>>>(la, gd);
        }
    }
    FROG_HIP_CHECK(hipGetLastError());
    ctx->pending_alpha = alpha;
    ctx->phase = 1;
    return FROG_OK;
}

int frog_deformable_phase_b(frog_ctx *ctx)
{
    CTX_GUARD(ctx);
    if (ctx->phase != 1) return fail(FROG_E_STATE, "phase_b without phase_a");
    hipStream_t s = ctx->stream;
    const GridGeom &g = ctx->geom;
    const float maxD = ctx->opt.max_displacement_ratio;
    if (ctx->centered_in_a) { ctx->phase = 2; return FROG_OK; }       // phase A already removed the group's own mean
    Span span(ctx, FROG_K_LATTICE);
    // :398: the group mean is removed only when no image is fixed
    cp_center_kernel<<<div_up(g.n_cp, 256), 256, 0, s>>>(ctx->grad.p, ctx->n_owned(), to_dev(g), ctx->nf ? 0u : ctx->nI, ctx->gridsum.p,
                                                        (double)maxD * g.spacing[0], (double)maxD * g.spacing[1],
                                                        (double)maxD * g.spacing[2], ctx->energy.p,
                                                        ctx->two_collectives ? ctx->gridsum.p + 3 * (size_t)g.n_cp : nullptr,
                                                        ctx->lat_mask.p, ctx->lat_inactive.p, ctx->ucoeff.p, ctx->ugrad.p);
    FROG_HIP_CHECK(hipGetLastError());
    ctx->phase = 2;
    return FROG_OK;
}

int frog_deformable_phase_c(frog_ctx *ctx, double *E)
{
    CTX_GUARD(ctx);
    if (ctx->phase != 2) return fail(FROG_E_STATE, "phase_c without phase_b");
    // The accept / reject decision (imageGroup.cxx:434-439) is on the device: the (all-reduced) oversize count.
    // Queued here, before the host knows it: the transformPoints() that run() calls next in either case (accepted: :118
    // with the proposal lattice; rejected: with the standing coefficients), into a shadow buffer -- the GPU keeps working
    // while the host waits for the four scalars, which are copied out first; the wait below is for that copy alone.
    // The commit itself (:441-468) costs nothing: once the host has the decision it exchanges the two buffers.
    // The scalars reach the host without a copy of their own when a transform through the lattice follows (it does unless the
    // context owns no point): its first thread writes them, then the step's sequence number, into pinned memory the host
    // spins on (k_grid.hip.h publish_step_scalars).
    const bool direct = ctx->h_energy_dev && ctx->own_pt_end > ctx->own_pt_begin;
    double seq = 0.0;
    if (direct) {
        seq = (double)(++ctx->scalar_seq);
    } else {
        FROG_HIP_CHECK(hipMemcpyAsync(ctx->h_energy, ctx->energy.p, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        FROG_HIP_CHECK(hipEventRecord(ctx->energy_copied, ctx->stream));
    }
    // the transformPoints() that run() calls next, from the lattice the guard's decision selects (on the device)
    int rc = launch_transform(ctx, ctx->pos2_spec.p, 0, true, seq);  // xyz2 itself changes only when the caller asks
    if (rc) return rc;
    ctx->xyz2_fresh = true;
    if (direct) {
        rc = wait_step_scalars(ctx, seq);
        if (rc) return rc;
    } else {
        FROG_HIP_CHECK(hipEventSynchronize(ctx->energy_copied));
    }
    const double e = std::sqrt(ctx->h_energy[0] / ctx->h_energy[1]), nbig = ctx->h_energy[2];
    if (ctx->h_energy[3] > 0) ctx->cull_need_build = true;       // this step's sweep found the culling list out of date
    if (!(ctx->opt.guarantee_diffeomorphism && nbig > 0)) {
        // accepted (:441-468): the proposal lattice becomes the coefficients -- the two buffers change roles
        std::swap(ctx->coeff.p, ctx->grad.p);
        std::swap(ctx->coeff.cap, ctx->grad.cap);
        std::swap(ctx->coeff.n, ctx->grad.n);
        std::swap(ctx->ucoeff.p, ctx->ugrad.p); std::swap(ctx->ucoeff.cap, ctx->ugrad.cap); std::swap(ctx->ucoeff.n, ctx->ugrad.n);   // (sparse lattices: their companions)
        ctx->coeff_zero = false;
    }
    ctx->phase = 0;
    if (E) *E = (ctx->opt.guarantee_diffeomorphism && nbig > 0) ? -1.0 : e;      // :434-439
    return FROG_OK;
}

int frog_deformable_step(frog_ctx *ctx, float alpha, double *E)
{
    if (ctx && !ctx->whole_group())
        return fail(FROG_E_STATE, "context owns a sub-range of images: use the phase_a/b/c entry points");
    int rc = frog_deformable_phase_a(ctx, alpha);
    if (rc) return rc;
    rc = frog_deformable_phase_b(ctx);
    if (rc) return rc;
    return frog_deformable_phase_c(ctx, E);
}

// ---- countInliers (imageGroup.cxx:988-1060) --------------------------------------------------
int frog_count_inliers(frog_ctx *ctx, frog_counts *per_image)
{
    CTX_GUARD(ctx);
    if (!per_image) return fail(FROG_E_INVALID, "null output");
    hipStream_t s = ctx->stream;
    const uint32_t nO = ctx->n_owned();
    for (uint32_t sub = 0; sub < ctx->n_sub; sub++)
        launch_sweep<SWEEP_COUNT>(ctx, sub, s);
    FROG_HIP_CHECK(hipGetLastError());
    count_reduce_kernel<<<div_up(nO, 64), 64, 0, s>>>(ctx->tile_counts.p, ctx->img_tile_ptr.p, ctx->n_groups, ctx->ib, nO, ctx->img_counts.p);
    FROG_HIP_CHECK(hipGetLastError());
    std::vector<long long> h((size_t)nO * 2);
    std::vector<float4> hem(ctx->nI);
    FROG_HIP_CHECK(hipMemcpyAsync(h.data(), ctx->img_counts.p, h.size() * sizeof(long long), hipMemcpyDeviceToHost, s));
    FROG_HIP_CHECK(hipMemcpyAsync(hem.data(), ctx->em.p, hem.size() * sizeof(float4), hipMemcpyDeviceToHost, s));
    FROG_HIP_CHECK(hipStreamSynchronize(s));
    for (uint32_t i = 0; i < nO; i++) {
        frog_counts &c = per_image[ctx->ib + i];
        c.points = ctx->poff[ctx->ib + i + 1] - ctx->poff[ctx->ib + i];
        c.inliers = h[(size_t)i * 2]; c.outliers = h[(size_t)i * 2 + 1];
        c.pairs = c.inliers + c.outliers;
        c.c1 = hem[ctx->ib + i].x; c.c2 = hem[ctx->ib + i].y; c.ratio = hem[ctx->ib + i].z; c.pad_ = 0;
    }
    return FROG_OK;
}

// ---- read-back ---------------------------------------------------------------------------------
int frog_get_points(frog_ctx *ctx, float *xyz, float *xyz2)
{
    CTX_GUARD(ctx);
    int rc = FROG_OK;
    if (xyz) rc = download_points(ctx, ctx->pos.p, xyz);
    if (!rc && xyz2) rc = download_points(ctx, ctx->pos2.p, xyz2);
    return rc;
}

int frog_set_points2(frog_ctx *ctx, const float *xyz2)
{
    CTX_GUARD(ctx);
    if (!xyz2) return fail(FROG_E_INVALID, "null input");
    std::vector<P3> h(ctx->P);
    for (size_t p = 0; p < ctx->P; p++)
        h[ctx->h_new_of_old[p]] = P3{ xyz2[3 * p], xyz2[3 * p + 1], xyz2[3 * p + 2] };
    ctx->xyz2_fresh = false; ctx->res_valid = false; ctx->disp_current = false;
    FROG_HIP_CHECK(hipMemcpyAsync(ctx->pos2.p, h.data(), h.size() * sizeof(P3), hipMemcpyHostToDevice, ctx->stream));
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return FROG_OK;
}

__global__ void gather_points_kernel(const P3 *pos2, const uint32_t *idx, uint32_t n, float *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const P3 p = pos2[idx[i]];
    out[3 * i] = p.x; out[3 * i + 1] = p.y; out[3 * i + 2] = p.z;
}

int frog_get_points2_subset(frog_ctx *ctx, const uint64_t *points, size_t n, float *xyz2)
{
    CTX_GUARD(ctx);
    if (n && (!points || !xyz2)) return fail(FROG_E_INVALID, "null argument");
    if (!n) return FROG_OK;
    std::vector<uint32_t> idx(n);
    for (size_t k = 0; k < n; k++) {
        if (points[k] >= ctx->P) return fail(FROG_E_INVALID, "point index out of range");
        idx[k] = ctx->h_new_of_old[points[k]];
    }
    FROG_HIP_CHECK(ctx->subset_idx.alloc(n));
    FROG_HIP_CHECK(ctx->subset_out.alloc(3 * n));
    hipStream_t s = ctx->stream;
    FROG_HIP_CHECK(hipMemcpyAsync(ctx->subset_idx.p, idx.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    gather_points_kernel<<<div_up(n, 256), 256, 0, s>>>(ctx->pos2.p, ctx->subset_idx.p, (uint32_t)n, ctx->subset_out.p);
    FROG_HIP_CHECK(hipGetLastError());
    FROG_HIP_CHECK(hipMemcpyAsync(xyz2, ctx->subset_out.p, 3 * n * sizeof(float), hipMemcpyDeviceToHost, s));
    FROG_HIP_CHECK(hipStreamSynchronize(s));
    return FROG_OK;
}

int frog_get_linear(frog_ctx *ctx, uint32_t image, double m16[16])
{
    CTX_GUARD(ctx);
    if (image >= ctx->nI || !m16) return fail(FROG_E_INVALID, "bad image");
    FROG_HIP_CHECK(hipMemcpyAsync(m16, ctx->mat.p + (size_t)image * 16, 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return FROG_OK;
}

int frog_get_em(frog_ctx *ctx, uint32_t image, float out[3])
{
    CTX_GUARD(ctx);
    if (image >= ctx->nI || !out) return fail(FROG_E_INVALID, "bad image");
    float4 v;
    FROG_HIP_CHECK(hipMemcpyAsync(&v, ctx->em.p + image, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    out[0] = v.x; out[1] = v.y; out[2] = v.z;
    return FROG_OK;
}

int frog_set_em(frog_ctx *ctx, uint32_t image, const float in[3])
{
    CTX_GUARD(ctx);
    if (image >= ctx->nI || !in) return fail(FROG_E_INVALID, "bad image");
    float4 v = make_float4(in[0], in[1], in[2], 0.f);
    FROG_HIP_CHECK(hipMemcpyAsync(ctx->em.p + image, &v, sizeof v, hipMemcpyHostToDevice, ctx->stream));
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return frog_stats_publish(ctx);
}

int frog_set_em_rows(frog_ctx *ctx, const float *table4, uint32_t image_begin, uint32_t image_end)
{
    CTX_GUARD(ctx);
    if (!table4 || image_begin > image_end || image_end > ctx->nI) return fail(FROG_E_INVALID, "bad rows");
    if (image_end > image_begin)
        FROG_HIP_CHECK(hipMemcpyAsync(ctx->em.p + image_begin, table4 + 4 * (size_t)image_begin, (size_t)(image_end - image_begin) * sizeof(float4),
                                      hipMemcpyHostToDevice, ctx->stream));
    return FROG_OK;
}

int frog_get_samples(frog_ctx *ctx, uint32_t image, float *samples, uint32_t *ordinals, int cap, int *n)
{
    CTX_GUARD(ctx);
    if (ctx->helper && image < ctx->nf) return frog_get_samples(ctx->helper, image, samples, ordinals, cap, n);
    if (image < ctx->ib || image >= ctx->ie) return fail(FROG_E_INVALID, "image not owned by this context");
    const uint32_t li = image - ctx->ib;
    uint32_t cnt = 0;
    const int ub = ctx->sel_used;
    FROG_HIP_CHECK(hipMemcpyAsync(&cnt, ctx->sample_count[ub].p + li, sizeof cnt, hipMemcpyDeviceToHost, ctx->stream));
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (n) *n = (int)cnt;
    const size_t m = std::min<size_t>(cnt, cap > 0 ? (size_t)cap : 0);
    if (m && samples)
        FROG_HIP_CHECK(hipMemcpyAsync(samples, ctx->samples.p + (size_t)li * ctx->sample_cap, m * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    if (m && ordinals)
        FROG_HIP_CHECK(hipMemcpyAsync(ordinals, ctx->sample_ord[ub].p + (size_t)li * ctx->sample_cap, m * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return FROG_OK;
}

// Stats::getHistogram (stats.cxx:121-131) over the retained samples; integer
// binning of values computed on the device, done on the host at output time.
int frog_get_histogram(frog_ctx *ctx, uint32_t image, float *bins, int cap, int *n)
{
    if (!ctx) return fail(FROG_E_INVALID, "null context");
    int cnt = 0;
    int rc = frog_get_samples(ctx, image, nullptr, nullptr, 0, &cnt);
    if (rc) return rc;
    std::vector<float> smp((size_t)std::max(cnt, 1));
    rc = frog_get_samples(ctx, image, smp.data(), nullptr, cnt, &cnt);
    if (rc) return rc;
    if (cnt == 0) { if (n) *n = 0; return FROG_OK; }
    float mx = smp[0];
    for (int i = 1; i < cnt; i++) mx = std::max(mx, smp[i]);
    const int size = (int)std::round(mx / 1.0f) + 1;
    std::vector<float> h((size_t)size, 0.f);
    for (int i = 0; i < cnt; i++) h[(size_t)std::round(smp[i] / 1.0f)]++;
    if (n) *n = size;
    if (bins) std::memcpy(bins, h.data(), (size_t)std::min(size, cap) * sizeof(float));
    return FROG_OK;
}

int frog_num_grids(const frog_ctx *ctx) { return ctx ? (int)ctx->grids.size() : 0; }

int frog_get_grid(frog_ctx *ctx, uint32_t image, int k, frog_grid_info *info, float *coeffs, size_t cap)
{
    CTX_GUARD(ctx);
    { const int rc_ = join_setup(ctx); if (rc_) return rc_; }
    if (k < 0 || k >= (int)ctx->grids.size()) return fail(FROG_E_INVALID, "bad lattice index");
    GridRecord &gr = ctx->grids[k];
    if (info) *info = gr.info;
    if (!coeffs) return FROG_OK;
    if (image < ctx->ib || image >= ctx->ie) return fail(FROG_E_INVALID, "image not owned by this context");
    const size_t G = (size_t)gr.info.dims[0] * gr.info.dims[1] * gr.info.dims[2];
    const size_t li = image - ctx->ib;
    const size_t nfl = std::min(cap, 3 * G);
    const float4 *src = gr.retired ? gr.kept->p : ctx->coeff.p;
    std::vector<float4> h(G);
    {
        // the image's nodes out of the lattice's layout (image-major or blocked: k_grid.hip.h lat()) into a contiguous run
        GridGeom lg{};
        lg.n_cp = (int)G;
        lg.blocked = gr.retired ? gr.blocked : ctx->geom.blocked;
        lg.lat_images = gr.retired ? gr.lat_images : ctx->geom.lat_images;
        lg.sparse = gr.retired ? gr.sparse : ctx->geom.sparse;
        lg.mask_words = gr.retired ? gr.mask_words : ctx->geom.mask_words;
        const uint32_t *mask = !lg.sparse ? nullptr : (gr.retired ? gr.kept_mask->p : ctx->lat_mask.p);
        const float4 *u = !lg.sparse ? nullptr : (gr.retired ? gr.kept_u->p : ctx->ucoeff.p);
        FROG_HIP_CHECK(ctx->extract_tmp.alloc(G));
        lattice_extract_kernel<<<div_up(G, 256), 256, 0, ctx->stream>>>(src, to_dev(lg), (uint32_t)li, mask, u, ctx->extract_tmp.p);
        FROG_HIP_CHECK(hipGetLastError());
        FROG_HIP_CHECK(hipMemcpyAsync(h.data(), ctx->extract_tmp.p, G * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    }
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < nfl; i++) { const float4 &v = h[i / 3]; coeffs[i] = (i % 3 == 0) ? v.x : (i % 3 == 1) ? v.y : v.z; }
    return FROG_OK;
}

int frog_set_hard_links(frog_ctx *ctx, const uint64_t *point, const uint64_t *partner, size_t n, float weight2)
{
    CTX_GUARD(ctx);
    ctx->rc_valid = false;            // the reference-order chains look the sums up by row only without landmark constraints
    if (n && (!point || !partner)) return fail(FROG_E_INVALID, "null argument");
    // keep the links of owned points, grouped by point in first-appearance order, link order preserved
    std::vector<uint32_t> pts, ptr(1, 0), prt;
    std::vector<std::vector<uint32_t>> per;
    std::vector<int64_t> slot(ctx->P, -1);
    for (size_t k = 0; k < n; k++) {
        if (point[k] >= ctx->P || partner[k] >= ctx->P) return fail(FROG_E_INVALID, "hard link point out of range");
        if (point[k] < ctx->own_pt_begin || point[k] >= ctx->own_pt_end) continue;
        if (slot[point[k]] < 0) { slot[point[k]] = (int64_t)pts.size(); pts.push_back(ctx->h_new_of_old[point[k]]); per.emplace_back(); }
        per[(size_t)slot[point[k]]].push_back(ctx->h_new_of_old[partner[k]]);
    }
    for (const auto &v : per) { prt.insert(prt.end(), v.begin(), v.end()); ptr.push_back((uint32_t)prt.size()); }
    hipStream_t s = ctx->stream;
    FROG_HIP_CHECK(hipStreamSynchronize(s));
    ctx->n_hard = (uint32_t)pts.size();
    ctx->hard_weight2 = weight2;
    ctx->res_valid = false;
    if (!ctx->n_hard) return FROG_OK;
    FROG_HIP_CHECK(ctx->hl_point.upload(pts, s));
    FROG_HIP_CHECK(ctx->hl_ptr.upload(ptr, s));
    FROG_HIP_CHECK(ctx->hl_partner.upload(prt, s));
    FROG_HIP_CHECK(ctx->hl_partial.alloc((size_t)2 * ctx->n_hard));
    FROG_HIP_CHECK(hipStreamSynchronize(s));
    return FROG_OK;
}

// saveErrorMaps, imageGroup.cxx:475-567
int frog_residual_sums(frog_ctx *ctx)
{
    CTX_GUARD(ctx);
    { const int rc_ = join_setup(ctx); if (rc_) return rc_; }
    if (ctx->phase != 0) return fail(FROG_E_STATE, "residual sums inside a deformable step");
    hipStream_t s = ctx->stream;
    const uint32_t n = ctx->own_pt_end - ctx->own_pt_begin;
    if (ctx->ref_order) {
        const int rc = ref_point_sums(ctx, false);
        if (rc) return rc;
    } else {
        for (uint32_t sub = 0; sub < ctx->n_sub; sub++) launch_sweep<SWEEP_DEFORMABLE>(ctx, sub, s);
        if (n && !sweep_fused_now(ctx, false)) combine_groups_kernel<<<div_up(n, 256), 256, 0, s>>>(ctx->group_sums.p, n, ctx->own_pt_begin, ctx->point_sums.p);
        ctx->point_sums_stale = false;
        if (ctx->n_hard)                                            // :520-533
            hard_links_kernel<<<div_up(ctx->n_hard, 64), 64, 0, s>>>(ctx->pos2.p, ctx->point_sums.p, ctx->hl_point.p, ctx->hl_ptr.p,
                                                                    ctx->hl_partner.p, ctx->n_hard, ctx->hard_weight2, nullptr);
    }
    FROG_HIP_CHECK(hipGetLastError());
    // host copies of the owned rows (internal numbering): sums and rebased coordinates
    ctx->h_res_sums.resize(n);
    ctx->h_res_pos.resize(n);
    if (n) {
        FROG_HIP_CHECK(hipMemcpyAsync(ctx->h_res_sums.data(), ctx->point_sums.p + ctx->own_pt_begin, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, s));
        FROG_HIP_CHECK(hipMemcpyAsync(ctx->h_res_pos.data(), ctx->pos.p + ctx->own_pt_begin, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, s));
    }
    FROG_HIP_CHECK(hipStreamSynchronize(s));
    ctx->res_valid = true;
    return FROG_OK;
}

int frog_get_error_map(frog_ctx *ctx, uint32_t image, frog_grid_info *info, float *out, size_t cap)
{
    CTX_GUARD(ctx);
    { const int rc_ = join_setup(ctx); if (rc_) return rc_; }
    if (!ctx->deformable || ctx->grids.empty()) return fail(FROG_E_STATE, "no lattice");
    if (!ctx->res_valid) return fail(FROG_E_STATE, "frog_get_error_map without frog_residual_sums");
    if (image < ctx->ib || image >= ctx->ie) return fail(FROG_E_INVALID, "image not owned by this context");
    const frog_grid_info &gi = ctx->grids.back().info;
    if (info) *info = gi;
    const size_t G = (size_t)gi.dims[0] * gi.dims[1] * gi.dims[2];
    if (!out) return FROG_OK;
    if (cap < 4 * G) return fail(FROG_E_INVALID, "error map buffer too small");
    std::fill(out, out + 4 * G, 0.f);                                           // :489
    const long long inc[3] = { 4, 4LL * gi.dims[0], 4LL * gi.dims[0] * gi.dims[1] };
    for (uint32_t p = ctx->poff[image]; p < ctx->poff[image + 1]; p++) {        // reference point order
        const uint32_t q = ctx->h_new_of_old[p] - ctx->own_pt_begin;
        const float4 sm = ctx->h_res_sums[q];
        if (sm.w == 0.f) continue;                                              // :535
        const float4 ps = ctx->h_res_pos[q];
        const float pos[3] = { ps.x, ps.y, ps.z };
        long long id = 0;
        for (int k = 0; k < 3; k++) {                                           // :539-544
            const float coord = (float)(((double)pos[k] - gi.origin[k]) / gi.spacing[k]);
            id += (long long)std::floor(coord) * inc[k];
        }
        if (id < 0 || (size_t)id + 3 >= 4 * G) continue;
        out[id + 3] += sm.w;
        out[id] += sm.x; out[id + 1] += sm.y; out[id + 2] += sm.z;
    }
    for (size_t i = 0; i < G; i++)                                              // :551-556
        if (out[4 * i + 3] > 0)
            for (int k = 0; k < 3; k++) out[4 * i + k] /= out[4 * i + 3];
    return FROG_OK;
}

int frog_get_point_sums(frog_ctx *ctx, float *out)
{
    CTX_GUARD(ctx);
    if (ctx->point_sums_stale && ctx->own_pt_end > ctx->own_pt_begin) {
        combine_groups_kernel<<<div_up(ctx->own_pt_end - ctx->own_pt_begin, 256), 256, 0, ctx->stream>>>(
            ctx->group_sums.p, ctx->own_pt_end - ctx->own_pt_begin, ctx->own_pt_begin, ctx->point_sums.p);
        FROG_HIP_CHECK(hipGetLastError());
        ctx->point_sums_stale = false;
    }
    std::vector<float4> h(ctx->P);
    FROG_HIP_CHECK(hipMemcpyAsync(h.data(), ctx->point_sums.p, h.size() * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (size_t p = 0; p < ctx->P; p++) std::memcpy(out + 4 * p, &h[ctx->h_new_of_old[p]], sizeof(float4));
    return FROG_OK;
}

int frog_get_gradient(frog_ctx *ctx, uint32_t image, float *out, size_t cap)
{
    CTX_GUARD(ctx);
    { const int rc_ = join_setup(ctx); if (rc_) return rc_; }
    if (!ctx->deformable) return fail(FROG_E_STATE, "no lattice");
    if (image < ctx->ib || image >= ctx->ie) return fail(FROG_E_INVALID, "image not owned by this context");
    const size_t G = (size_t)ctx->geom.n_cp;
    const size_t n = std::min(cap, 4 * G);
    FROG_HIP_CHECK(ctx->extract_tmp.alloc(G));
    {
        GridGeom lg = ctx->geom;
        lg.sparse = false;                      // the gradient lattice has an entry for every pair (only stray points write it)
        lattice_extract_kernel<<<div_up(G, 256), 256, 0, ctx->stream>>>(ctx->gradf.p, to_dev(lg), image - ctx->ib, nullptr, nullptr, ctx->extract_tmp.p);
    }
    FROG_HIP_CHECK(hipGetLastError());
    FROG_HIP_CHECK(hipMemcpyAsync(out, ctx->extract_tmp.p, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return FROG_OK;
}

int frog_cull_stats_linear(frog_ctx *ctx, uint64_t *lists_built, uint64_t *listed, uint64_t *owned)
{
    CTX_GUARD(ctx);
    if (lists_built) *lists_built = ctx->cull_lin_builds;
    if (owned) *owned = ctx->L_own;
    if (listed) {
        *listed = 0;
        if (ctx->cull_lin_builds && ctx->act_cnt.n) {
            if (!ctx->deformable) {                 // still in the linear stage: count the current list
                FROG_HIP_CHECK(hipMemsetAsync(ctx->lin_listed.p, 0, sizeof(unsigned long long), ctx->stream));
                cull_count_kernel<<<CULL_COUNT_BLOCKS, 256, 0, ctx->stream>>>(ctx->act_cnt.p, (uint32_t)ctx->act_cnt.n, ctx->lin_listed.p);
            }
            unsigned long long v = 0;
            FROG_HIP_CHECK(hipMemcpyAsync(&v, ctx->lin_listed.p, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
            FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            *listed = v;
        }
    }
    return FROG_OK;
}

int frog_cull_stats(frog_ctx *ctx, uint64_t *lists_built, uint64_t *listed, uint64_t *owned)
{
    CTX_GUARD(ctx);
    if (lists_built) *lists_built = ctx->cull_builds;
    if (owned) *owned = ctx->L_own;
    if (listed) {
        *listed = 0;
        if (ctx->cull_builds && ctx->act_cnt.n) {
            std::vector<uint32_t> h(ctx->act_cnt.n);
            FROG_HIP_CHECK(hipMemcpyAsync(h.data(), ctx->act_cnt.p, ctx->act_cnt.bytes(), hipMemcpyDeviceToHost, ctx->stream));
            FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
            for (uint32_t v : h) *listed += v & ~CULL_DUP_BIT;
        }
    }
    return FROG_OK;
}

__global__ void test_probability_kernel(const float4 em, const float *d2, size_t n, float *fast, float *exact)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fast[i] = inlier_probability(d2[i], em_derived_of(em));             // exactly what a sweep step does with its d2
    exact[i] = inlier_probability_exact(sqrt_rn(d2[i]), em);
}

int frog_test_inlier_probability(int device, const float em3[3], const float *d2, size_t n, float *fast, float *exact)
{
    if (!em3 || (n && (!d2 || !fast || !exact))) return fail(FROG_E_INVALID, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(FROG_E_NODEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return fail(FROG_E_INVALID, "device index out of range");
    FROG_HIP_CHECK(hipSetDevice(device));
    if (!n) return FROG_OK;
    DevBuf<float> dd, df, de;
    FROG_HIP_CHECK(dd.alloc(n)); FROG_HIP_CHECK(df.alloc(n)); FROG_HIP_CHECK(de.alloc(n));
    FROG_HIP_CHECK(hipMemcpy(dd.p, d2, n * sizeof(float), hipMemcpyHostToDevice));
    test_probability_kernel<<<div_up(n, 256), 256>>>(make_float4(em3[0], em3[1], em3[2], 0.f), dd.p, n, df.p, de.p);
    FROG_HIP_CHECK(hipGetLastError());
    FROG_HIP_CHECK(hipMemcpy(fast, df.p, n * sizeof(float), hipMemcpyDeviceToHost));
    FROG_HIP_CHECK(hipMemcpy(exact, de.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return FROG_OK;
}

// what a deformable sweep step does with its d2 before the threshold band decides (k_links.hip.h): form 0 = the one-exponential
// form inside the pair's range, 1 = the general form, 2 = the one-exponential form's value below threshold - band (an outlier
// whatever its range)
__global__ void test_weight_pair_kernel(const float4 emA, const float4 emB, float threshold, const float *d2, size_t n, float *weight,
                                        unsigned char *form)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float theta = threshold - THRESHOLD_BAND;
    const EmFast fA = em_fast_of(emA, theta);
    EmFast fB = em_fast_of(emB, theta);
    fB.lo = fmaxf(fB.lo, fA.lo); fB.hi = fminf(fB.hi, fA.hi);
    bool in_range;
    float w = inlier_weight_pair(d2[i], fA, fB, in_range);
    unsigned char f = 0;
    if (w < theta) f = 2;
    else if (!in_range) { w = fminf(inlier_probability(d2[i], em_derived_of(emA)), inlier_probability(d2[i], em_derived_of(emB))); f = 1; }
    weight[i] = w; form[i] = f;
}

int frog_test_inlier_weight_pair(int device, const float emA3[3], const float emB3[3], float threshold, const float *d2, size_t n,
                                 float *weight, unsigned char *form)
{
    if (!emA3 || !emB3 || (n && (!d2 || !weight || !form))) return fail(FROG_E_INVALID, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(FROG_E_NODEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return fail(FROG_E_INVALID, "device index out of range");
    FROG_HIP_CHECK(hipSetDevice(device));
    if (!n) return FROG_OK;
    DevBuf<float> dd, dw; DevBuf<unsigned char> df;
    FROG_HIP_CHECK(dd.alloc(n)); FROG_HIP_CHECK(dw.alloc(n)); FROG_HIP_CHECK(df.alloc(n));
    FROG_HIP_CHECK(hipMemcpy(dd.p, d2, n * sizeof(float), hipMemcpyHostToDevice));
    test_weight_pair_kernel<<<div_up(n, 256), 256>>>(make_float4(emA3[0], emA3[1], emA3[2], 0.f), make_float4(emB3[0], emB3[1], emB3[2], 0.f),
                                                      threshold, dd.p, n, dw.p, df.p);
    FROG_HIP_CHECK(hipGetLastError());
    FROG_HIP_CHECK(hipMemcpy(weight, dw.p, n * sizeof(float), hipMemcpyDeviceToHost));
    FROG_HIP_CHECK(hipMemcpy(form, df.p, n, hipMemcpyDeviceToHost));
    return FROG_OK;
}

__global__ void test_weights_kernel(const double *f, size_t n, double *out)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double F[4];
    bspline_weights<double>(F, f[i]);
    for (int k = 0; k < 4; k++) out[4 * i + k] = F[k];
}

int frog_test_bspline_weights(int device, const double *f, size_t n, double *out4n)
{
    if (n && (!f || !out4n)) return fail(FROG_E_INVALID, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(FROG_E_NODEVICE, "no HIP device");
    if (device < 0 || device >= ndev) return fail(FROG_E_INVALID, "device index out of range");
    FROG_HIP_CHECK(hipSetDevice(device));
    if (!n) return FROG_OK;
    DevBuf<double> df, dw;
    FROG_HIP_CHECK(df.alloc(n)); FROG_HIP_CHECK(dw.alloc(4 * n));
    FROG_HIP_CHECK(hipMemcpy(df.p, f, n * sizeof(double), hipMemcpyHostToDevice));
    test_weights_kernel<<<div_up(n, 256), 256>>>(df.p, n, dw.p);
    FROG_HIP_CHECK(hipGetLastError());
    FROG_HIP_CHECK(hipMemcpy(out4n, dw.p, 4 * n * sizeof(double), hipMemcpyDeviceToHost));
    return FROG_OK;
}

int frog_test_cull_ranges(frog_ctx *ctx, uint64_t *ranges, uint64_t *with_election)
{
    CTX_GUARD(ctx);
    if (!ranges || !with_election) return fail(FROG_E_INVALID, "null output");
    *ranges = 0; *with_election = 0;
    if (ctx->cull_builds && ctx->act_cnt.n) {
        std::vector<uint32_t> h(ctx->act_cnt.n);
        FROG_HIP_CHECK(hipMemcpyAsync(h.data(), ctx->act_cnt.p, ctx->act_cnt.bytes(), hipMemcpyDeviceToHost, ctx->stream));
        FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (uint32_t v : h) {
            if (v & ~CULL_DUP_BIT) (*ranges)++;
            if (v & CULL_DUP_BIT) (*with_election)++;
        }
    }
    return FROG_OK;
}

int frog_test_stray_points(frog_ctx *ctx, uint64_t *n)
{
    CTX_GUARD(ctx);
    { const int rc_ = join_setup(ctx); if (rc_) return rc_; }
    if (!n) return fail(FROG_E_INVALID, "null output");
    unsigned int v = 0;
    FROG_HIP_CHECK(hipMemcpyAsync(&v, ctx->stray.p + 2, sizeof v, hipMemcpyDeviceToHost, ctx->stream));
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    *n = v;
    return FROG_OK;
}

int frog_test_em_refit(frog_ctx *ctx, int term_by_term)
{
    CTX_GUARD(ctx);
    const uint32_t nO = ctx->n_owned(), cap = (uint32_t)ctx->sample_cap;
    const int cur = ctx->sel_used;
    hipStream_t s = ctx->stream;
    if (term_by_term)
        em_kernel<<<nO, 256, 0, s>>>(ctx->samples.p, ctx->sample_count[cur].p, cap, ctx->ib, ctx->em.p,
                                     ctx->opt.stats_max_iterations, ctx->opt.stats_epsilon);
    else
        em_scan_kernel<<<nO, EM_THREADS, 0, s>>>(ctx->samples.p, ctx->sample_count[cur].p, cap, ctx->ib, ctx->em.p,
                                                 ctx->opt.stats_max_iterations, ctx->opt.stats_epsilon, nullptr);     // test hook: from cold guesses
    FROG_HIP_CHECK(hipGetLastError());
    return frog_stats_publish(ctx);
}

int frog_profile_enable(frog_ctx *ctx, int on)
{
    CTX_GUARD(ctx);
    ctx->profiling = (on == 2 || on == 3) ? 2 : (on != 0);
    ctx->profile_stride = on == 3 ? 4 : 1;
    return FROG_OK;
}

int frog_profile_read(frog_ctx *ctx, frog_kernel_time *out, int reset)
{
    CTX_GUARD(ctx);
    FROG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (auto &sp : ctx->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            ctx->timed_ms[sp.slot] += (double)ms;
            ctx->timed_n[sp.slot] += 1;
        }
        ctx->free_events.emplace_back(sp.a, sp.b);
    }
    ctx->spans.clear();
    // a group's line: all its launches, at the mean of the timed ones (the same thing when every launch is timed)
    for (int k = 0; k < FROG_K_COUNT_; k++) {
        const uint64_t launches = ctx->timed_n[k] ? std::max(ctx->span_seen[k], ctx->timed_n[k]) : 0;
        ctx->ktime[k].launches = launches;
        ctx->ktime[k].ms_total = ctx->timed_n[k] ? ctx->timed_ms[k] / (double)ctx->timed_n[k] * (double)launches : 0.0;
    }
    if (out) std::memcpy(out, ctx->ktime, sizeof ctx->ktime);
    if (reset) {
        std::memset(ctx->ktime, 0, sizeof ctx->ktime);
        std::memset(ctx->span_seen, 0, sizeof ctx->span_seen);
        std::memset(ctx->timed_ms, 0, sizeof ctx->timed_ms);
        std::memset(ctx->timed_n, 0, sizeof ctx->timed_n);
    }
    return FROG_OK;
}

// rows [row_begin[r], row_begin[r + 1]) of every rank r != self: slab slot r -> the coordinate table
constexpr int UNPACK_MAX_RANKS = 64;
struct UnpackArgs {
    uint64_t row_begin[UNPACK_MAX_RANKS + 1];
    uint64_t slot_bytes;            // distance between two ranks' slots
    uint32_t world, self;           // self == world: the own rows are copied too (frog_comm_unpack_slab_step)
    // frog_comm_unpack_slab_step: the slots' trailers (4 doubles at slot + trailer_off) are added up over the ranks for the scalars in
    // sum_mask and the step's four scalars handed to the host (k_grid.hip.h publish_step_scalars)
    uint64_t trailer_off;
    uint32_t sum_mask;
    double *energy, *host_scalars;
    double seq;
};
// `snap` (null: no list to check): the block also leaves the largest distance of the rows it copies from the culling list's
// snapshot in disp_part[blockIdx.y * gridDim.x + blockIdx.x] (k_cull.hip.h: what cull_disp_kernel computes in a pass of its own)
__global__ __launch_bounds__(256) void unpack_slab_kernel(const P3 *slab, P3 *pos2, const UnpackArgs a, const P3 *snap, uint32_t *disp_part)
{
    __shared__ uint32_t sh[4];
    const uint32_t r = blockIdx.y;
    if (a.sum_mask && blockIdx.x == 0 && r == 0 && threadIdx.x == 0) {
        // the ranks' trailers in rank order: integers (oversize counts, flags) add exactly; the energy sums of a linear step
        // in the same order on every rank, so every rank prints the same E
        for (int k = 0; k < 4; k++) {
            if (!(a.sum_mask >> k & 1u)) continue;
            double sum = 0.0;
            for (uint32_t q = 0; q < a.world; q++)
                sum += reinterpret_cast<const double *>(reinterpret_cast<const unsigned char *>(slab) + q * a.slot_bytes + a.trailer_off)[k];
            a.energy[k] = sum;
        }
        __threadfence();
        if (a.host_scalars) {               // null: the host gets them by copy + event (frog_comm_unpack_slab_step)
            #pragma unroll
            for (int k = 0; k < 4; k++) __hip_atomic_store(&a.host_scalars[k], a.energy[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __threadfence_system();
            __hip_atomic_store(&a.host_scalars[7], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    uint32_t m = 0;
    if (r != a.self) {
        const uint64_t n = a.row_begin[r + 1] - a.row_begin[r];
        const P3 *slot = reinterpret_cast<const P3 *>(reinterpret_cast<const unsigned char *>(slab) + r * a.slot_bytes);
        for (uint64_t k = (uint64_t)blockIdx.x * 256u + threadIdx.x; k < n; k += (uint64_t)gridDim.x * 256u) {
            const P3 v = slot[k];
            pos2[a.row_begin[r] + k] = v;
            if (snap) {
                const P3 q = snap[a.row_begin[r] + k];
                const float dx = v.x - q.x, dy = v.y - q.y, dz = v.z - q.z;
                m = max(m, __float_as_uint(__builtin_sqrtf(dx * dx + dy * dy + dz * dz)) & 0x7FFFFFFFu);
            }
        }
    }
    if (!snap) return;
    #pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, off, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) disp_part[blockIdx.y * gridDim.x + blockIdx.x] = max(max(sh[0], sh[1]), max(sh[2], sh[3]));
}

int frog_comm_unpack_slab(frog_ctx *ctx, const void *slab, uint64_t slot_rows, uint32_t world_size, const uint64_t *row_begin, uint32_t self)
{
    CTX_GUARD(ctx);
    if (!slab || !row_begin || world_size < 1 || world_size > (uint32_t)UNPACK_MAX_RANKS || self >= world_size)
        return fail(FROG_E_INVALID, "bad slab arguments");
    UnpackArgs a{};
    uint64_t longest = 0;
    for (uint32_t r = 0; r <= world_size; r++) a.row_begin[r] = row_begin[r];
    for (uint32_t r = 0; r < world_size; r++) {
        if (row_begin[r + 1] < row_begin[r] || row_begin[r + 1] > ctx->P) return fail(FROG_E_INVALID, "rows must be ascending and inside the table");
        longest = std::max(longest, row_begin[r + 1] - row_begin[r]);
    }
    if (longest > slot_rows) return fail(FROG_E_INVALID, "slot shorter than the longest shard");
    a.slot_bytes = slot_rows * sizeof(P3); a.world = world_size; a.self = self;
    if (longest == 0) return FROG_OK;
    const dim3 grid((unsigned)std::min<uint64_t>(div_up(longest, 256), 1024), world_size);
    // The own rows' displacement against the culling list's snapshot was measured by the transform that produced them
    // (disp_part[0 .. disp_own_n)); the rows arriving here are measured as they are copied, behind those entries -- the check
    // before the next sweep then has every row covered without a pass of its own over the whole table.
    const bool list = ctx->deformable ? (cull_active(ctx) && ctx->cull_builds > 0) : (cull_active_linear(ctx) && ctx->cull_lin_builds > 0);
    const bool measure = ctx->disp_current && list && !ctx->cull_need_build && ctx->pos2_snap.p && ctx->disp_part.p
                         && (size_t)ctx->disp_own_n + (size_t)grid.x * grid.y <= ctx->disp_part.n;
    unpack_slab_kernel<<<grid, 256, 0, ctx->stream>>>(reinterpret_cast<const P3 *>(slab), ctx->pos2.p, a,
                                                      measure ? ctx->pos2_snap.p : nullptr, measure ? ctx->disp_part.p + ctx->disp_own_n : nullptr);
    FROG_HIP_CHECK(hipGetLastError());
    if (measure) { ctx->disp_n = ctx->disp_own_n + grid.x * grid.y; ctx->disp_others = true; }
    else { ctx->disp_current = false; ctx->disp_others = false; }      // measured by the check before the next sweep
    return FROG_OK;
}

// ---- sharded contexts: two collectives per deformable iteration, one per linear iteration (include/frog_hip.h) -----------------
int frog_comm_mode(frog_ctx *ctx, int two_collectives)
{
    CTX_GUARD(ctx);
    if (ctx->phase != 0) return fail(FROG_E_STATE, "frog_comm_mode inside a deformable step");
    ctx->two_collectives = two_collectives != 0;
    // the third lattice of the speculative flow is part of lattice_alloc's head-room only when the mode is known there; a mode
    // switched on with a lattice already standing takes it here, outside any iteration (it used to be allocated inside
    // frog_step_speculate, between two steps of the timed loop)
    if (ctx->two_collectives && !ctx->whole_group() && ctx->grad.p && ctx->grad_spare.n != ctx->grad.n)
        FROG_HIP_CHECK(ctx->grad_spare.alloc(ctx->grad.n, ctx->grad.cap));
    return FROG_OK;
}

int frog_transform_points_slab(frog_ctx *ctx, int apply, int after_step, void *slab, uint64_t slot_rows, uint32_t self)
{
    CTX_GUARD(ctx);
    if (!slab) return fail(FROG_E_INVALID, "null slab");
    const uint32_t n = ctx->own_pt_end - ctx->own_pt_begin;
    if (n > slot_rows) return fail(FROG_E_INVALID, "slot shorter than this context's shard");
    if (after_step && ctx->phase != 2) return fail(FROG_E_STATE, "frog_transform_points_slab(after_step) without phase_b");
    if (after_step && apply) return fail(FROG_E_INVALID, "the transform behind a step does not re-base");
    unsigned char *slot = static_cast<unsigned char *>(slab) + (size_t)self * FROG_SLAB_SLOT_BYTES(slot_rows);
    double *trailer = reinterpret_cast<double *>(slot + FROG_SLAB_SLOT_BYTES(slot_rows) - FROG_SLAB_TRAILER_BYTES);
    ctx->xyz2_fresh = false; ctx->res_valid = false;
    if (!n) {           // a context whose images are all empty still owes its trailer
        slab_trailer_kernel<<<1, 1, 0, ctx->stream>>>(ctx->energy.p, trailer);
        FROG_HIP_CHECK(hipGetLastError());
        return FROG_OK;
    }
    // the kernels address their output by point index: row own_pt_begin is the slot's first
    P3 *out = reinterpret_cast<P3 *>(slot) - ctx->own_pt_begin;
    return launch_transform(ctx, out, apply, after_step != 0, 0.0, trailer);
}

int frog_comm_unpack_slab_step(frog_ctx *ctx, const void *slab, uint64_t slot_rows, uint32_t world_size, const uint64_t *row_begin,
                               uint32_t self, uint32_t sum_mask)
{
    CTX_GUARD(ctx);
    if (!slab || !row_begin || world_size < 1 || world_size > (uint32_t)UNPACK_MAX_RANKS || self >= world_size || sum_mask > 15u)
        return fail(FROG_E_INVALID, "bad slab arguments");
    UnpackArgs a{};
    uint64_t longest = 0;
    for (uint32_t r = 0; r <= world_size; r++) a.row_begin[r] = row_begin[r];
    for (uint32_t r = 0; r < world_size; r++) {
        if (row_begin[r + 1] < row_begin[r] || row_begin[r + 1] > ctx->P) return fail(FROG_E_INVALID, "rows must be ascending and inside the table");
        longest = std::max(longest, row_begin[r + 1] - row_begin[r]);
    }
    if (longest > slot_rows) return fail(FROG_E_INVALID, "slot shorter than the longest shard");
    if (row_begin[self] != ctx->own_pt_begin || row_begin[self + 1] != ctx->own_pt_end) return fail(FROG_E_INVALID, "row_begin[self] is not this context's shard");
    a.slot_bytes = FROG_SLAB_SLOT_BYTES(slot_rows); a.world = world_size; a.self = world_size;      // every slot, the own one included
    a.trailer_off = a.slot_bytes - FROG_SLAB_TRAILER_BYTES;
    a.sum_mask = sum_mask;
    if (sum_mask) {
        a.energy = ctx->energy.p; a.host_scalars = ctx->h_energy_dev;
        a.seq = (double)(++ctx->scalar_seq);
        ctx->pending_seq = a.seq;
        ctx->finish_deformable = ctx->phase == 2;
    }
    const dim3 grid((unsigned)std::min<uint64_t>(std::max<uint64_t>(div_up(longest, 256), 1), 1024), world_size);
    // the own rows' displacement against the culling list's snapshot: measured by the transform that wrote them into the slab
    // (disp_part[0 .. disp_own_n), recorded as "speculative" because its output was not the table); the other ranks' as they are copied
    ctx->disp_current = ctx->disp_spec; ctx->disp_spec = false;
    const bool list = ctx->deformable ? (cull_active(ctx) && ctx->cull_builds > 0) : (cull_active_linear(ctx) && ctx->cull_lin_builds > 0);
    const bool measure = ctx->disp_current && list && !ctx->cull_need_build && ctx->pos2_snap.p && ctx->disp_part.p
                         && (size_t)ctx->disp_own_n + (size_t)grid.x * grid.y <= ctx->disp_part.n;
    unpack_slab_kernel<<<grid, 256, 0, ctx->stream>>>(reinterpret_cast<const P3 *>(slab), ctx->pos2.p, a,
                                                      measure ? ctx->pos2_snap.p : nullptr, measure ? ctx->disp_part.p + ctx->disp_own_n : nullptr);
    FROG_HIP_CHECK(hipGetLastError());
    // (the own slot's rows are measured twice when `measure`: once by the transform, once here -- a maximum does not mind)
    if (measure) { ctx->disp_n = ctx->disp_own_n + grid.x * grid.y; ctx->disp_others = true; }
    else { ctx->disp_current = false; ctx->disp_others = false; }      // measured by the check before the next sweep
    // No device-visible address of the pinned scalar block (hipHostGetDevicePointer failed, or FROG_SCALARS_COPY=1): the summed
    // scalars reach the host by copy + event, as in frog_deformable_phase_c, and frog_step_finish waits for the event.
    ctx->scalars_by_copy = sum_mask && !ctx->h_energy_dev;
    if (ctx->scalars_by_copy) {
        FROG_HIP_CHECK(hipMemcpyAsync(ctx->h_energy, ctx->energy.p, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        FROG_HIP_CHECK(hipEventRecord(ctx->energy_copied, ctx->stream));
    }
    return FROG_OK;
}

int frog_step_speculate(frog_ctx *ctx)
{
    CTX_GUARD(ctx);
    if (ctx->pending_seq == 0.0 || !ctx->finish_deformable || ctx->phase != 2 || ctx->speculated)
        return fail(FROG_E_STATE, "frog_step_speculate needs a deformable step whose scalars are on their way (frog_comm_unpack_slab_step)");
    if (ctx->grad_spare.n != ctx->grad.n) {
        // (frog_comm_mode was switched on after the lattice was made: the one allocation outside a set-up)
        FROG_HIP_CHECK(ctx->grad_spare.alloc(ctx->grad.n, ctx->grad.cap));
    }
    // as if accepted (imageGroup.cxx:441-468): the proposal lattice becomes the coefficients, the spare one takes the next proposals;
    // what is needed to undo it stays in spec_*
    ctx->spec_coeff_zero = ctx->coeff_zero;
    auto rotate = [](DevBuf<float4> &a, DevBuf<float4> &b) { std::swap(a.p, b.p); std::swap(a.cap, b.cap); std::swap(a.n, b.n); };
    rotate(ctx->coeff, ctx->grad);              // coeff = proposal, grad = old coefficients
    rotate(ctx->grad, ctx->grad_spare);         // grad = spare, spare = old coefficients
    rotate(ctx->ucoeff, ctx->ugrad); rotate(ctx->ugrad, ctx->ugrad_spare);      // (sparse lattices: their companions)
    ctx->coeff_zero = false;
    ctx->speculated = true;
    ctx->phase = 0;
    return FROG_OK;
}

int frog_step_finish(frog_ctx *ctx, double *E)
{
    CTX_GUARD(ctx);
    if (ctx->pending_seq == 0.0) return fail(FROG_E_STATE, "frog_step_finish without frog_comm_unpack_slab_step(sum_mask != 0)");
    int rc = FROG_OK;
    if (ctx->scalars_by_copy) { ctx->scalars_by_copy = false; FROG_HIP_CHECK(hipEventSynchronize(ctx->energy_copied)); }
    else rc = wait_step_scalars(ctx, ctx->pending_seq);
    ctx->pending_seq = 0.0;
    if (rc) return rc;
    if (ctx->h_energy[3] > 0) ctx->cull_need_build = true;       // a rank's sweep found its culling list out of date
    const double e = std::sqrt(ctx->h_energy[0] / ctx->h_energy[1]), nbig = ctx->h_energy[2];
    if (ctx->finish_deformable) {
        // updateDeformableTransforms' decision (imageGroup.cxx:434-439) and commit (:441-468), as frog_deformable_phase_c
        ctx->finish_deformable = false;
        const bool rejected = ctx->opt.guarantee_diffeomorphism && nbig > 0;
        auto rotate = [](DevBuf<float4> &a, DevBuf<float4> &b) { std::swap(a.p, b.p); std::swap(a.cap, b.cap); std::swap(a.n, b.n); };
        if (ctx->speculated) {
            ctx->speculated = false;
            if (rejected) {
                // undo frog_step_speculate: the lattices take their old roles; the phase A queued meanwhile worked on coordinates of
                // a step that did not happen -- per-point sums, staged tiles, proposals in the spare lattice: nothing of it is read again
                rotate(ctx->grad, ctx->grad_spare);
                rotate(ctx->coeff, ctx->grad);
                rotate(ctx->ugrad, ctx->ugrad_spare); rotate(ctx->ucoeff, ctx->ugrad);
                ctx->coeff_zero = ctx->spec_coeff_zero;
                ctx->phase = 0;
                ctx->xyz2_fresh = false; ctx->res_valid = false;
            }
        } else {
            if (ctx->phase != 2) return fail(FROG_E_STATE, "frog_step_finish: no deformable step pending");
            if (!rejected) { rotate(ctx->coeff, ctx->grad); rotate(ctx->ucoeff, ctx->ugrad); ctx->coeff_zero = false; }
            ctx->phase = 0;
        }
        if (E) *E = rejected ? -1.0 : e;
        return FROG_OK;
    }
    if (E) *E = e;
    return FROG_OK;
}

int frog_comm_buffer(frog_ctx *ctx, int which, void **ptr, size_t *bytes, size_t *row_begin, size_t *row_end)
{
    CTX_GUARD(ctx);
    size_t b = 0, rb = 0, re = 0;
    void *p = nullptr;
    switch (which) {
    case FROG_BUF_XYZ2: ctx->xyz2_exported = true; ctx->disp_current = false; p = ctx->pos2.p; b = ctx->pos2.bytes(); rb = ctx->own_pt_begin; re = ctx->own_pt_end; break;
    case FROG_BUF_EM: p = ctx->em.p; b = ctx->em.bytes(); rb = ctx->ib; re = ctx->ie; break;
    case FROG_BUF_ENERGY: p = ctx->energy.p; b = ctx->energy.bytes(); rb = 0; re = 4; break;
    case FROG_BUF_GRIDSUM:
        if (!ctx->deformable) return fail(FROG_E_STATE, "no lattice");
        // 3 G proposal sums; with two collectives per iteration (frog_comm_mode) the step's energy sums and list flag behind them
        p = ctx->gridsum.p; re = 3 * (size_t)ctx->geom.n_cp + (ctx->two_collectives ? 4 : 0); b = re * sizeof(double); rb = 0; break;
    default: return fail(FROG_E_INVALID, "unknown buffer");
    }
    if (ptr) *ptr = p;
    if (bytes) *bytes = b;
    if (row_begin) *row_begin = rb;
    if (row_end) *row_end = re;
    return FROG_OK;
}

} // extern "C"
