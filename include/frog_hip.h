/*
 * frog_hip.h -- C ABI of the MI355X (gfx950) device library libfrog_hip.so.
 *
 * The drop-in seam of valette/FROG's groupwise solver: the reference has no
 * plugin/FFI interface (frog is one statically linked TU set,
 * registration/CMakeLists.txt:17), so the boundary is the set of ImageGroup
 * methods that ImageGroup::run (registration/imageGroup.cxx:31-157) invokes in
 * its loops.  Each entry point below replaces one of them; a host that keeps
 * the reference's control flow (this repo's libfrog_host.so, or the reference's
 * own run() with the stub shown in INTEGRATION.md) calls them in the same
 * order.  Plain pointers and sizes only; the library owns all device memory.
 *
 * Conventions carried over from the reference: single caller thread, calls
 * strictly sequential (the reference's phases are barriers); every function
 * returns an int status (FROG_OK = 0; the reference's only error channels are
 * exit(1) and the -1 energy, which is kept: *E = -1 when a deformable step is
 * rejected by the diffeomorphism guard).  There is NO CPU fallback: without a
 * HIP device frog_create fails with FROG_E_NODEVICE.
 *
 * Multi-GPU: one process per GPU, each context owns a contiguous range of
 * images [image_begin, image_end) (the unit of every omp-for in the reference)
 * and keeps read-only replicas of what the link loops read from other images
 * (transformed coordinates, EM parameters).  The *_local / phase_* entry points
 * stop where the reference reads another image's state; the host runs the
 * collective (RCCL all-gather / all-reduce) on the buffers exposed by
 * frog_comm_buffer and calls the next phase.  With one rank the plain entry
 * points run all phases back to back.
 */
#ifndef FROG_HIP_H
#define FROG_HIP_H

#include "frog_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct frog_ctx frog_ctx;

/* ---- life cycle ----------------------------------------------------------- */

int frog_device_count(void);
/* Initialises the HIP runtime on `device` (the first HIP call of a process: device discovery, code objects) and returns.  A host
 * may call it from a thread of its own while it reads its input (bin/frog does, beside readPairs); frog_create works without it. */
int frog_device_warm(int device);
/* Message for the last non-OK status returned on this thread. */
const char *frog_last_error(void);

/* ImageGroup ctor + readPairs' in-memory result + setupStats
 * (imageGroup.h:52-82, imageGroup.cxx:1353-1417, :1151-1159).
 * Copies the model to `device`, builds the device-side link layout and one
 * reservoir per image.  image_begin/image_end select the images this context
 * updates (0, n_images for a single GPU). */
int frog_create(const frog_model *model, const frog_options *options, int device,
                uint32_t image_begin, uint32_t image_end, frog_ctx **out);
void frog_destroy(frog_ctx *ctx);

/* Run all work of this context on an existing HIP stream (hipStream_t), e.g.
 * the stream a host framework issues its collectives against.  Default: a
 * stream created by frog_create. */
int frog_set_stream(frog_ctx *ctx, void *hip_stream);
/* Where frog_create's time went, in seconds: [0] the host-side layout build (Morton numbering, partner-major 4-byte records,
 * tiles: all host cores), [1] device allocations, uploads and the first kernels until the model is resident, [2] the reservoir
 * selections replayed ahead of time on the side stream (selections_replayed of them: Stats::addSample's mt19937 acceptance tests
 * depend on nothing but the call count, stats.h:58-76, so a whole run's are produced here instead of beside the sweeps). */
int frog_create_seconds(frog_ctx *ctx, double seconds3[3], int *selections_replayed);
/* How many times a lattice set-up (frog_deformable_setup*) had to allocate lattice buffers after frog_create -- 0 when
 * frog_options::max_levels_hint told frog_create how fine the lattices would get and its estimate of the registered group's box
 * held: no multi-gigabyte hipMalloc inside the caller's loops (5 to 1 500 ms each on the test boxes). */
int frog_lattice_reallocations(frog_ctx *ctx, int *count);

/* The stream the context's work is enqueued on and its device: what a caller needs to order its own collectives
 * (RCCL: include/frog_comm.h) against the library's kernels.  Either pointer may be NULL. */
int frog_get_stream(frog_ctx *ctx, void **hip_stream, int *device);
int frog_synchronize(frog_ctx *ctx);

/* ---- the six methods run() calls ------------------------------------------- */

/* setupLinearTransforms (imageGroup.cxx:806-848). */
int frog_linear_init(frog_ctx *ctx, const float anchor[3]);
/* transformPoints(apply) (imageGroup.cxx:910-916 -> image.cxx:3-13). */
int frog_transform_points(frog_ctx *ctx, int apply);
/* updateStats (imageGroup.cxx:569-598): reservoir refresh + EM fit per image. */
int frog_update_stats(frog_ctx *ctx);
/* updateLinearTransforms (imageGroup.cxx:1063-1149); *E = sqrt(sum w2 d2 / sum w2). */
int frog_linear_step(frog_ctx *ctx, double *E);

/* ImageGroup::RANSAC + RANSACBatch (imageGroup.cxx:629-804), the stage that replaces the linear
 * iterations when fixed images are present (run(), :40-49, `-r 1`, the default with -fi).
 * `iterations / batches` candidates per batch, batch b drawn from std::mt19937(b * 1000): four
 * random (point, link) correspondences between the image's xyz and the partners' xyz, a
 * least-squares similarity transform through them (vtkLandmarkTransform, similarity mode), its
 * inlier count = half-links of the image with |T(xyz) - partner xyz2|^2 < inlier_distance^2.
 * Candidates whose |determinant| is outside [1/max_scale, max_scale] are skipped.  The best
 * candidate is refitted on all its inlier half-links and becomes the image's matrix
 * (frog_get_linear); *n_inliers = the best candidate's count (before the refit, as upstream
 * reports it).  All candidates are counted in one launch.  Upstream's `batches` is
 * omp_get_num_procs(), i.e. a property of the machine: pass the value you want reproduced.
 * `model` must be the model the context was created from (it is not retained by the context;
 * the candidate draws need the link table in reference order). */
typedef struct frog_ransac_options {
    int32_t iterations;         /* -ri   5000   imageGroup.h:70 */
    int32_t batches;            /*       omp_get_num_procs() upstream */
    float   inlier_distance;    /* -rid  50     imageGroup.h:73 */
    float   max_scale;          /* -rs   10     imageGroup.h:74 */
} frog_ransac_options;
int frog_ransac(frog_ctx *ctx, const frog_model *model, uint32_t image, const frog_ransac_options *options,
                int64_t *n_inliers);
/* setupDeformableTransforms(level) (imageGroup.cxx:159-218). */
int frog_deformable_setup(frog_ctx *ctx, int level, frog_grid_info *out);
/* updateDeformableTransforms(alpha) (imageGroup.cxx:234-472); *E = -1 and no
 * state change when guarantee_diffeomorphism is set and a coefficient exceeds
 * max_displacement_ratio * spacing. */
int frog_deformable_step(frog_ctx *ctx, float alpha, double *E);
/* countInliers (imageGroup.cxx:988-1060); per_image has n_images entries, only
 * the owned range is filled. */
int frog_count_inliers(frog_ctx *ctx, frog_counts *per_image);

/* ---- read-back for the writers (saveTransforms, histograms, bbox) ---------- */

uint64_t frog_num_points(const frog_ctx *ctx);
uint32_t frog_num_images(const frog_ctx *ctx);
/* xyz / xyz2 as 3*P floats (either may be NULL). */
int frog_get_points(frog_ctx *ctx, float *xyz, float *xyz2);
/* Overwrite xyz2 (3*P floats) -- test hook to start two implementations from
 * identical coordinates. */
int frog_set_points2(frog_ctx *ctx, const float *xyz2);
/* xyz2 of the listed points (global indices in the model's order), 3 floats each: what the
 * reference reads from a handful of Point::xyz2 per iteration (landmark measures,
 * imageGroup.cxx:1229-1281) without copying the whole table back. */
int frog_get_points2_subset(frog_ctx *ctx, const uint64_t *points, size_t n, float *xyz2_out3n);
int frog_get_linear(frog_ctx *ctx, uint32_t image, double matrix16[16]);
int frog_get_em(frog_ctx *ctx, uint32_t image, float c1_c2_ratio[3]);
int frog_set_em(frog_ctx *ctx, uint32_t image, const float c1_c2_ratio[3]);
/* Rows [image_begin, image_end) of the mixture table from a host table of n_images x 4 floats (c1, c2, ratio, 0), queued on the
 * context's stream; NOT followed by frog_stats_publish.  For a host that stands in for the all-reduce of FROG_BUF_EM
 * (a single-process proxy of one rank of a sharded run: the other ranks' rows come from a table handed in). */
int frog_set_em_rows(frog_ctx *ctx, const float *table4, uint32_t image_begin, uint32_t image_end);
/* Retained samples of the last refresh, and the ordinal (position in the
 * image's half-link traversal) each one was drawn from. */
int frog_get_samples(frog_ctx *ctx, uint32_t image, float *samples, uint32_t *ordinals,
                     int cap, int *n);
/* Stats::getHistogram(1.0) over the retained samples (stats.cxx:121-131). */
int frog_get_histogram(frog_ctx *ctx, uint32_t image, float *bins, int cap, int *n);
int frog_num_grids(const frog_ctx *ctx);
/* Lattice k of the chain: geometry + 3*G coefficients of `image`. */
int frog_get_grid(frog_ctx *ctx, uint32_t image, int k, frog_grid_info *info,
                  float *coeffs, size_t cap_floats);
/* Per-point (sDisp xyz, sWeight) of the last deformable step (4*P floats) and the
 * gradient lattice (4*G floats) as left by the scatter -- diagnostics / tests. */
int frog_get_point_sums(frog_ctx *ctx, float *out4P);
int frog_get_gradient(frog_ctx *ctx, uint32_t image, float *out4G, size_t cap_floats);

/* Landmark constraints (-lc): Point::hardLinks (imageGroup.cxx:1210-1225).  n directed links
 * point <- partner (global point indices in the model's order; the links of one point are used in
 * the order given).  In every deformable step and in the error maps each adds
 * weight2 * (partner.xyz2 - point.xyz2) / weight2 to the point's sums after its regular links,
 * and weight2 * dist2 / weight2 to the energy sums (:280-295, :520-533); weight2 =
 * (nImages * landmarksConstraintsWeight)^2 (:237).  n = 0 removes them. */
int frog_set_hard_links(frog_ctx *ctx, const uint64_t *point, const uint64_t *partner, size_t n, float weight2);

/* saveErrorMaps (imageGroup.cxx:475-567).  frog_residual_sums runs the half-link sweep on
 * the CURRENT xyz2 (per-point sDisp/sWeight over inlier links, :493-533) for the owned
 * images -- no collective, the xyz2 replica is whole after transformPoints.
 * frog_get_error_map then adds one owned image's sums at the lattice node
 * floor((xyz - origin) / spacing) of the LAST lattice, in the reference's point order
 * (host side: 4 f32 adds per point whose order must be the reference's), divides by the
 * weight (:551-556) and returns 4*G floats (what the reference writes to
 * errorMaps/<image>.nii.gz).  Points outside the lattice image are skipped (undefined
 * behaviour upstream).  Any later step invalidates the sums. */
int frog_residual_sums(frog_ctx *ctx);
int frog_get_error_map(frog_ctx *ctx, uint32_t image, frog_grid_info *info, float *out4G, size_t cap_floats);

/* ---- split phases for one-process-per-GPU runs ----------------------------- */

enum {
    FROG_BUF_XYZ2    = 0,  /* float[P][3] xyz2 in the library's internal point order (a permutation
                            * inside each image); owned rows written by transform          */
    FROG_BUF_EM      = 1,  /* float4[n_images]  c1,c2,ratio,0; owned rows written by stats  */
    FROG_BUF_ENERGY  = 2,  /* double[4]   sum w2 d2, sum w2, #oversize coefficients, 0      */
    FROG_BUF_GRIDSUM = 3   /* double[3*G] sum over owned images of the proposed coefficients (+ 4 doubles with frog_comm_mode(ctx, 1)) */
};
/* Device pointer + size of a collective buffer; `row_begin`/`row_end` (in
 * elements of the buffer's row type) delimit what this context owns (NULL ok). */
int frog_comm_buffer(frog_ctx *ctx, int which, void **device_ptr, size_t *bytes,
                     size_t *row_begin, size_t *row_end);

/* Ragged shards gathered with ONE equal-size all-gather: every rank contributes its owned xyz2 rows at the start of a slot of
 * `slot_rows` rows (>= the longest shard); `slab` holds the world_size slots in rank order.  This call copies the other
 * ranks' rows (row_begin[r] .. row_begin[r + 1], r != self) from the slab into FROG_BUF_XYZ2, in one launch on the context's
 * stream (a host that unpacks with one copy per rank pays world_size - 1 launches per iteration). */
int frog_comm_unpack_slab(frog_ctx *ctx, const void *slab, uint64_t slot_rows, uint32_t world_size,
                          const uint64_t *row_begin, uint32_t self);

/* ---- Two collectives per deformable iteration, one per linear iteration (round 5) ------------------------------------------
 * The split phases below cost a deformable iteration three collectives: all-reduce of the proposal sums, all-reduce of
 * (energy sums, oversize count, list flag), all-gather of the coordinates -- each latency-bound, against ~0.15 ms of kernels
 * per rank at eight ranks.  With frog_comm_mode(ctx, 1) the middle one disappears:
 *   - the energy sums and the list flag ride on the all-reduce of the proposal sums: FROG_BUF_GRIDSUM grows by four doubles,
 *     phase A leaves the rank's sums there, phase B takes the group's back;
 *   - the oversize count (known only after phase B) rides on the NEXT coordinate gather: the transform that follows the step
 *     is queued at once, from the proposal lattice unless the rank's OWN count forbids it, straight into the rank's slot of
 *     the slab the gather moves (frog_transform_points_slab, after_step = 1), with the rank's four scalars as the slot's
 *     trailer; frog_comm_unpack_slab_step copies every slot's rows into FROG_BUF_XYZ2, adds the trailers up and hands the
 *     step's scalars to the host; frog_step_finish waits for them, decides (imageGroup.cxx:434-439) and commits.  A rejected
 *     step leaves speculative coordinates in the table, which is harmless: run()'s reject path re-bases from the standing
 *     coefficients and gathers again before anything reads them (imageGroup.cxx:97-115).
 * A linear iteration's two sums travel the same way (sum_mask 0b1011): one collective instead of two.
 *
 * Slab layout: world_size slots of FROG_SLAB_SLOT_BYTES(slot_rows) bytes, slot r = rank r's rows (12 bytes each, slot_rows >=
 * the longest shard) padded to a multiple of 64 bytes, then a trailer of FROG_SLAB_TRAILER_BYTES whose first four doubles are
 * the rank's (sum w2 d2, sum w2, oversize count, list flag).  ONE equal-size all-gather of slot bytes moves it
 * (ncclAllGather in place: include/frog_comm.h frog_comm_all_gather_slab). */
#define FROG_SLAB_TRAILER_BYTES 64
#define FROG_SLAB_SLOT_BYTES(slot_rows) (((size_t)(slot_rows) * 12 + 63) / 64 * 64 + FROG_SLAB_TRAILER_BYTES)
int frog_comm_mode(frog_ctx *ctx, int two_collectives);
/* transformPoints of the owned images into slot `self` of `slab` (not into FROG_BUF_XYZ2: frog_comm_unpack_slab_step puts every
 * slot there after the gather).  after_step = 1: the transform that follows frog_deformable_phase_b, see above. */
int frog_transform_points_slab(frog_ctx *ctx, int apply, int after_step, void *slab, uint64_t slot_rows, uint32_t self);
/* Every slot's rows -> FROG_BUF_XYZ2 (one launch).  sum_mask: bit k set = scalar k of FROG_BUF_ENERGY becomes the sum of the
 * ranks' trailers (0b0100 after a deformable step: the oversize count; 0b1011 after a linear step: energy sums and list flag);
 * sum_mask != 0 also hands the four scalars to the host: call frog_step_finish next.  0: coordinates only. */
int frog_comm_unpack_slab_step(frog_ctx *ctx, const void *slab, uint64_t slot_rows, uint32_t world_size,
                               const uint64_t *row_begin, uint32_t self, uint32_t sum_mask);
/* Waits for the scalars.  After a deformable step: *E as frog_deformable_phase_c (-1 = rejected, nothing committed),
 * the proposal committed otherwise; after a linear step: *E. */
int frog_step_finish(frog_ctx *ctx, double *E);
/* The host's own latency out of the loop: between frog_comm_unpack_slab_step(sum_mask 0b0100) and frog_step_finish a host may
 * call frog_step_speculate and then frog_deformable_phase_a for the NEXT iteration -- the step is taken as accepted (the proposal
 * lattice becomes the coefficients, a third lattice takes the next proposals), so that the GPU has the next sweep in its queue
 * while the scalars travel to the host.  frog_step_finish then confirms, or -- the step was rejected, 4 of 650 on the benchmark
 * group -- gives the lattices their old roles back; the phase A queued meanwhile is void (*E = -1 as usual; continue with
 * run()'s reject path).  Not across a statistics refresh or the end of a level: those need the decision first. */
int frog_step_speculate(frog_ctx *ctx);

/* updateStats, owned images only; afterwards all-reduce(sum) FROG_BUF_EM
 * (non-owned rows are zero) and call frog_stats_publish. */
int frog_update_stats_local(frog_ctx *ctx);
int frog_stats_publish(frog_ctx *ctx);
/* transformPoints, owned images; afterwards all-gather FROG_BUF_XYZ2. */
int frog_transform_points_local(frog_ctx *ctx, int apply);
/* updateLinearTransforms, owned images; afterwards all-reduce(sum) the first
 * two doubles of FROG_BUF_ENERGY and read E with frog_energy_read. */
int frog_linear_step_local(frog_ctx *ctx);
int frog_energy_read(frog_ctx *ctx, double *E, double *n_oversize);
/* Bounding box of the owned images' xyz; all-reduce min/max on the host side,
 * then frog_deformable_setup_bounds with the group-wide box. */
int frog_bounds_local(frog_ctx *ctx, double mins[3], double maxs[3]);
int frog_deformable_setup_bounds(frog_ctx *ctx, int level, const double mins[3],
                                 const double maxs[3], frog_grid_info *out);
/* phase A: link pass, scatter, control-point step, sum of proposals over owned
 * images -> all-reduce(sum) FROG_BUF_GRIDSUM and FROG_BUF_ENERGY[0..1];
 * phase B: subtract the group mean, count oversize coefficients ->
 * all-reduce(sum) FROG_BUF_ENERGY[2]; phase C: commit or reject, *E as above. */
int frog_deformable_phase_a(frog_ctx *ctx, float alpha);
int frog_deformable_phase_b(frog_ctx *ctx);
int frog_deformable_phase_c(frog_ctx *ctx, double *E);

/* ---- certified outlier culling of the deformable sweep ------------------------------
 * In updateDeformableTransforms a half-link with weight < inlierThreshold adds nothing
 * (imageGroup.cxx:268-278).  The library keeps a list of the half-links that are not PROVABLY below
 * the threshold (distance bound per image from its mixture + the points' displacement since the list
 * was built; frog_amd/csrc/device/k_cull.hip.h) and sweeps only those; per-point sums, lattices,
 * coordinates and the census are bit-identical to sweeping every half-link (a skipped link would have added +0.0 to f32
 * chains whose order is kept), the two f64 energy sums equal up to their re-association.  On by default; the environment variable FROG_CULL=0 at frog_create turns
 * it off.  frog_cull_stats reports what it did: lists built so far, half-links in the last list and
 * half-links owned (0 listed = no list yet). */
int frog_cull_stats(frog_ctx *ctx, uint64_t *lists_built, uint64_t *listed_half_links, uint64_t *owned_half_links);
/* The same for the LINEAR stage (updateLinearTransforms has no threshold, imageGroup.cxx:1100-1117, but a half-link whose
 * weight is exactly zero adds nothing to its 18 sums either: the list leaves out the half-links whose distance puts the
 * sweep's weight at exactly zero; FROG_CULL_LINEAR=0 turns it off; the 18 sums are per-lane f64 accumulators, so results equal
 * the full sweep's up to f64 re-association, 1e-16 relative): lists built during the linear stage, half-links in the
 * last of them, half-links owned. */
int frog_cull_stats_linear(frog_ctx *ctx, uint64_t *lists_built, uint64_t *listed_half_links, uint64_t *owned_half_links);

/* ---- test hook: the sweep's inlier weight -------------------------------------------
 * Evaluates, on `device`, Stats::getInlierProbability (stats.h:84-92) for n SQUARED distances d2 (what a sweep step
 * has in hand: the f32 sum of three squared differences) with the mixture (c1, c2, ratio), twice: `fast` = the f32
 * form the half-link sweeps use for every link, `exact` = the form with the reference's own promotions (f64 exp,
 * evaluated at the correctly rounded f32 square root of d2, the reference's `dist`) that decides weights within 1e-4
 * of the inlier threshold.  tests/test_gpu_round2.py compares both with the reference build of stats.cxx. */
int frog_test_inlier_probability(int device, const float c1_c2_ratio[3], const float *d2, size_t n,
                                 float *fast, float *exact);

/* Test hook: the DEFORMABLE sweeps' weight of a half-link between an image with mixture A and one with mixture B, for n squared
 * distances, as a sweep step forms it before the threshold band decides: weight[i], and form[i] = 0 (one exponential for
 * min(pA, pB), d2 inside the pair's range), 1 (the general form: `fast` of frog_test_inlier_probability for both images),
 * 2 (the one-exponential value, below inlierThreshold - 1e-4: the sweep drops the link as an outlier without asking for its
 * range).  tests/test_gpu_round2.py compares with the reference build of stats.cxx. */
int frog_test_inlier_weight_pair(int device, const float c1_c2_ratio_a[3], const float c1_c2_ratio_b[3], float inlier_threshold,
                                 const float *d2, size_t n, float *weight, unsigned char *form);

/* Test hook: vtkBSplineTransformWeights (imageGroup.cxx:221-232) as the device's scatter and reference-order kernels
 * evaluate it: the four f64 weights of every fraction f[i] into out4n[4 i .. 4 i + 3].  tests/test_gpu_round5.py compares
 * them bit for bit with the reference's own function (oracle/_ref/libfrog_refweights.so). */
int frog_test_bspline_weights(int device, const double *f, size_t n, double *out4n);

/* Test hook: how many points the B-spline scatter has found outside the brick they were sorted into since the context
 * was created (they are handled, through global atomics: slowly and in no fixed order).  Inside the lattice's box --
 * always, for the group's own points -- this must stay 0; a non-zero count means the (image, brick, cell) sort is broken. */
int frog_test_stray_points(frog_ctx *ctx, uint64_t *n);
/* test hook: non-empty (tile, partner group) ranges of the current culling list, and how many of them hold a step in
 * which two lanes carry the same point (those are swept with the lane election, the others without; k_cull.hip.h) */
int frog_test_cull_ranges(frog_ctx *ctx, uint64_t *ranges, uint64_t *with_election);

/* Test hook: Stats::estimateDistribution (stats.cxx:14-70) again, on the samples the last refresh retained and from
 * the CURRENT (c1, c2, ratio) of every owned image (set them with frog_set_em first), with the term-by-term form of
 * the accumulators (term_by_term != 0) or the prefix-sum form frog_update_stats uses.  Both must give the same bits. */
int frog_test_em_refit(frog_ctx *ctx, int term_by_term);

/* ---- live kernel timing (HIP events on the context's stream) ------------------ */
enum {
    FROG_K_SWEEP_LINEAR = 0,   /* half-link sweep of updateLinearTransforms     */
    FROG_K_SWEEP_DEFORMABLE,   /* half-link sweep of updateDeformableTransforms */
    FROG_K_SCATTER,            /* B-spline gradient scatter                     */
    FROG_K_LATTICE,            /* control-point step, mean removal, commit      */
    FROG_K_TRANSFORM,          /* transformPoints                               */
    FROG_K_STATS,              /* reservoir + distances + EM fit                */
    FROG_K_COMBINE,            /* per-point sums of the partner groups added up, energy reduction */
    FROG_K_CULL,               /* outlier-culling list: validity check before a deformable sweep, rebuilds */
    FROG_K_SWEEP_BUILD,        /* the deformable sweep that walks EVERY half-link and writes the next culling list (once per list) */
    FROG_K_SWEEP_LINEAR_BUILD, /* the linear sweep that does the same for the linear stage's list (once per statistics refresh) */
    FROG_K_COUNT_
};
typedef struct frog_kernel_time {
    double   ms_total;      /* launches x the mean hipEventElapsedTime of the timed launches (= their sum when all are timed) */
    uint64_t launches;      /* all launches of the group, timed or not */
} frog_kernel_time;
/* on = 1: every launch of the kernels above is bracketed by a pair of hipEvents recorded on the
 * context's stream.  The markers keep consecutive kernels from overlapping their tails and
 * ramp-ups: measured 6 % of the iteration rate with all seven groups bracketed.  on = 2 puts events on
 * the half-link sweeps only (the kernels a roofline is quoted for), on their own dispatch packets: 1.6 % -- such a launch starts
 * ~6 us late and holds its successor back ~5.  on = 3: as 2 with events on every list-writing sweep and on ONE steady sweep
 * in four (the steady launches of a level are alike to 1 %): 0.4 %.  on = 0: off. */
int frog_profile_enable(frog_ctx *ctx, int on);
/* Waits for the stream, adds up the recorded pairs into out[FROG_K_COUNT_];
 * reset != 0 clears the accumulators afterwards. */
int frog_profile_read(frog_ctx *ctx, frog_kernel_time *out, int reset);

#ifdef __cplusplus
}
#endif
#endif /* FROG_HIP_H */
