/*
 * frog_host.h -- C ABI of the host-side library (libfrog_host.so).
 *
 * File formats and host control flow of the FROG groupwise solver, i.e. the
 * parts of registration/imageGroup.cxx + registration/frog.cxx that stay on the
 * CPU: parsing pairs.bin, driving the iteration schedule of ImageGroup::run and
 * writing the result files.  All numeric work is delegated to libfrog_hip.so
 * (frog_hip.h); this library contains no solver arithmetic and no CPU fallback.
 *
 * Reference interfaces replaced (paths relative to /root/reference):
 *   ImageGroup::readPairs            registration/imageGroup.cxx:1353-1417
 *   pairs.bin writer                 match/match.cpp:675-744
 *   ImageGroup::run                  registration/imageGroup.cxx:31-157
 *   main / flag parsing              registration/frog.cxx:8-221
 *   writeFrogJSON (-j form)          tools/transformIO.h:163-258
 *   saveDistanceHistograms           registration/imageGroup.cxx:850-885
 *   saveMeasures                     registration/imageGroup.cxx:1475-1491
 */
#ifndef FROG_HOST_H
#define FROG_HOST_H

#include "frog_types.h"
#include "frog_hip.h"
#include "frog_match.h"
#include "frog_chain.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- pairs.bin ------------------------------------------------------------ */

/* An in-memory pairs.bin: per-image point tables, the pair blocks in file
 * order, and the half-link CSR in reference order (built on load). */
typedef struct frog_pairs frog_pairs;

/* readPairs (imageGroup.cxx:1353-1417).  pointIdType is u32 (INT_PTIDS=ON, the
 * reference default, CMakeLists.txt:8-12).  Returns NULL on I/O error; a block
 * with size 0 is the reference's "Error : number of pairs is 0" (exit(1)):
 * reported as *status = FROG_E_INVALID. */
frog_pairs *frog_pairs_read(const char *path, int *status);
/* Writes the format of match.cpp:684-742. */
int  frog_pairs_write(const frog_pairs *p, const char *path);
void frog_pairs_free(frog_pairs *p);

/* SoA/CSR view (pointers stay valid until frog_pairs_free). */
void frog_pairs_model(const frog_pairs *p, frog_model *out);
uint64_t frog_pairs_num_pairs(const frog_pairs *p);
uint64_t frog_pairs_num_points(const frog_pairs *p);
uint32_t frog_pairs_num_images(const frog_pairs *p);
uint32_t frog_pairs_num_blocks(const frog_pairs *p);
/* block b: images, size, and pointers to its (p1,p2) index arrays */
int frog_pairs_block(const frog_pairs *p, uint32_t b, uint16_t *image1, uint16_t *image2,
                     uint32_t *size, const uint32_t **p1, const uint32_t **p2);

/* Builds a frog_pairs from caller arrays (copies).  blocks are given as
 * block_image1/2[nb], block_ptr[nb+1] into p1/p2. */
frog_pairs *frog_pairs_from_arrays(uint32_t n_images, const uint32_t *point_offset,
                                   const float *xyz, const float *other /* may be NULL */,
                                   uint32_t n_blocks, const uint16_t *block_image1,
                                   const uint16_t *block_image2, const uint64_t *block_ptr,
                                   const uint32_t *p1, const uint32_t *p2);

/* Host threads this library's parallel loops start (readPairs' CSR, error maps, transforms/): what OpenMP offers
 * (OMP_NUM_THREADS, -nt), capped by the CPUs the process may use -- affinity mask and cgroup CPU quota -- and by 64.  The
 * reference asks omp_get_num_procs() (frog.cxx -nt default); on a host that shows 256 hardware threads to a container
 * with a 16-CPU quota that starts 256 threads which are throttled together. */
int frog_host_threads(void);

/* ---- synthetic groups (the reference ships no data) -- */
typedef struct frog_synth_params {
    uint32_t n_images;
    uint32_t points_per_image;      /* exact                                    */
    uint32_t n_landmarks;           /* common-space landmarks, 0 => points_per_image */
    double   pairs_per_block;       /* mean pairs per linked image pair         */
    uint32_t partners_per_image;    /* 0 => all image pairs; else ~k random partners */
    float    outlier_fraction;      /* fraction of false (random) pairs, 0.3    */
    float    noise_sigma;           /* keypoint localisation noise, mm (2.0)    */
    float    bump_amplitude;        /* smooth deformation amplitude, mm (15)    */
    float    scale_min, scale_max;  /* per-axis scale range (0.8, 1.25)         */
    float    translation_range;     /* +- mm (100)                              */
    uint64_t seed;
} frog_synth_params;

void frog_synth_defaults(frog_synth_params *p);
frog_pairs *frog_synth_generate(const frog_synth_params *p);

/* Appends n points without links to `image` (after its existing points; later images' points
 * shift).  This is how the reference stores landmarks: extra entries of Image::points
 * (imageGroup.cxx:1185-1201), moved by transformPoints like every keypoint and counted in
 * the bounding boxes. */
int frog_pairs_append_points(frog_pairs *p, uint32_t image, const float *xyz, uint32_t n);

/* Overwrites the coordinates of all points of `image` (3 floats per point).  Used for fixed
 * images, whose keypoints are moved to their registered position before the solve
 * (readAndApplyFixedImagesTransforms, imageGroup.cxx:1419-1456: `xyz := T(xyz)`). */
int frog_pairs_set_points(frog_pairs *p, uint32_t image, const float *xyz);

/* ---- one rank's share of a timed schedule --------------------------------------------------------------------------
 * The loops of ImageGroup::run (imageGroup.cxx:54-66, :78-128) for ONE context, timed: what `bench.py` measures, without
 * an interpreter between two iterations.  `comm` = NULL: the context owns the whole group (or is a stand-alone proxy of one
 * rank: see proxy_*), no collective is issued; otherwise the context's communicator of include/frog_comm.h (one process per
 * GPU: frog_comm_create_rank over RCCL, or frog_comm_create_shm), bound to `ctx`, and the places where the reference's loops
 * read another image's state are its collectives, exactly as in `bin/frog -ng N` (image_group.cpp runSharded).
 *
 * Sequence: setupLinearTransforms, transformPoints, warmup_linear untimed linear iterations; then, between two barriers
 * (frog_comm_barrier + frog_synchronize), `linear` linear iterations, transformPoints(apply), and for every level with
 * per_level[l] > 0 a lattice set-up and per_level[l] ACCEPTED deformable iterations with the reference's regrid /
 * alpha-halving state machine and re-basing; updateStats whenever the iteration counter of its loop is a multiple of
 * stat_interval (the linear counter runs on from the warm-up, as in run()). */
#define FROG_SCHEDULE_MAX_LEVELS 16
#define FROG_SCHEDULE_MAX_LATTICES 512
typedef struct frog_schedule_plan {
    uint32_t plan_bytes, result_bytes;      /* sizeof(frog_schedule_plan), sizeof(frog_schedule_result) as the CALLER sees them:
                                             * a binding whose layout differs is refused instead of being misread     */
    int32_t warmup_linear;
    int32_t linear;
    int32_t n_levels;
    int32_t per_level[FROG_SCHEDULE_MAX_LEVELS];
    int32_t stat_interval;          /* statIntervalUpdate, 10                                         */
    float   deformable_alpha;       /* 0.02                                                           */
    float   anchor[3];              /* linearInitializationAnchor                                     */
    int32_t profile;                /* frog_profile_enable mode for the timed region: 0, 1 (every kernel group, read per
                                     * phase behind a device drain) or 2 (the half-link sweeps only)   */
    int32_t time_comm;              /* frog_comm_timing for the timed region                          */
    /* stand-alone proxy of one rank (comm = NULL, context owns a sub-range): what the other ranks would send -- their
     * xyz2 (model order, 3 floats per point of the whole group; own rows ignored) installed once after the first
     * transformPoints, and their rows of the mixture table (n_images x 4 floats) after every statistics refresh */
    const float *proxy_xyz2;
    const float *proxy_em;
} frog_schedule_plan;

typedef struct frog_schedule_lattice {
    int32_t level, dims[3], iterations;     /* accepted iterations taken on this lattice */
    double  setup_host_s;                   /* host time of its set-up call               */
} frog_schedule_lattice;

typedef struct frog_schedule_result {
    double   elapsed_s;                                 /* the timed region on this rank's clock, barrier to barrier */
    double   phase_s[1 + FROG_SCHEDULE_MAX_LEVELS];     /* linear, level 0, level 1, ...                             */
    int32_t  iterations;                                /* timed iterations done                                     */
    int32_t  grids_per_level[FROG_SCHEDULE_MAX_LEVELS];
    int32_t  n_lattices;
    frog_schedule_lattice lattices[FROG_SCHEDULE_MAX_LATTICES];
    double   final_E;                                   /* as run() records it: rounded to float                     */
    frog_kernel_time kernels[FROG_K_COUNT_];            /* frog_profile_read over the timed region                   */
    frog_kernel_time kernels_by_phase[1 + FROG_SCHEDULE_MAX_LEVELS][FROG_K_COUNT_];     /* profile == 1 only         */
    double   comm_ms[4];                                /* frog_comm_timing_read                                     */
    uint64_t comm_calls[4], comm_sampled[4];
    uint64_t replica_hash;      /* FNV-1a over the bits of this rank's replica of every xyz2 and of the mixture table after a
                                 * final transformPoints(apply): equal on every rank, whatever carried the collectives       */
} frog_schedule_result;

int frog_run_schedule(frog_ctx *ctx, struct frog_comm *comm, const frog_schedule_plan *plan, frog_schedule_result *out);

/* ---- NIfTI-1 writer for lattice images ------------------------------------------
 * Replaces vtkNIFTIImageWriter at tools/transformIO.h:196-208 (B-spline coefficient
 * sidecars `<i>.json.<n>.nii.gz`, 3 components) and registration/imageGroup.cxx:559-563
 * (`errorMaps/<i>.nii.gz`, 4 components).  `interleaved` holds dims[0]*dims[1]*dims[2]
 * voxels, x fastest, n_components floats each (the vtkImageData layout); the file stores
 * them plane by plane (dim[5] = n_components), FLOAT32, spacing in pixdim, origin in the
 * qform/sform offsets (what the reference's readers use, transformIO.h:439-453).
 * A path ending in ".gz" is gzip-compressed. */
int frog_nifti_write(const char *path, const uint32_t dims[3], const double spacing[3],
                     const double origin[3], uint32_t n_components, const float *interleaved);

/* ---- surf3d keypoint files (the inputs of `match`) ----------------------------------
 * Rows x, y, z, scale, laplacianSign, response, descriptor... as match.cpp reads them:
 * ".csv" (match.cpp:117-146), ".csv.gz" (:48-83), ".bin" (24 + 48 floats per row,
 * :149-179 -- including upstream's extra row at end of file, see keypoints_io.cpp).
 * frog_keypoints_view fills a frog_keypoints (frog_match.h) pointing into the file
 * object; frog_keypoints_select keeps the listed rows (the pruning of match.cpp:548-590);
 * frog_keypoints_write writes the same formats (writeCSV, :85-114). */
typedef struct frog_keypoint_file frog_keypoint_file;
frog_keypoint_file *frog_keypoints_read(const char *path, int *status);
void frog_keypoints_free(frog_keypoint_file *f);
uint32_t frog_keypoints_count(const frog_keypoint_file *f);
void frog_keypoints_view(const frog_keypoint_file *f, frog_keypoints *out);
int frog_keypoints_select(frog_keypoint_file *f, const uint32_t *keep, uint32_t n_keep);
int frog_keypoints_write(const char *path, const frog_keypoints *k);

/* ---- transform files (what frog writes; tools/transformIO.h:375-460 reads) -----------
 * frog_transform_read parses transforms/<i>.json in either form (coefficients inline, or in
 * the .nii.gz sidecars named by "file", looked up next to the JSON) into the link list of
 * include/frog_chain.h; the object owns the coefficient arrays the links point to.
 * frog_volume_geometry returns the voxel grid of a NIfTI-1 (.nii/.nii.gz) or MetaImage (.mhd)
 * volume: the sampling grid of CheckDiffeomorphism. */
typedef struct frog_transform_file frog_transform_file;
frog_transform_file *frog_transform_read(const char *json_path, int *status);
void frog_transform_free(frog_transform_file *f);
uint32_t frog_transform_num_links(const frog_transform_file *f);
const frog_chain_link *frog_transform_links(const frog_transform_file *f);
int frog_volume_geometry(const char *path, uint32_t dims[3], double spacing[3], double origin[3]);

/* ---- scalar volumes (tools/VolumeTransform.cxx:86-101 reader, :146-202 writers) ----------------
 * NIfTI-1 single files (.nii, .nii.gz) and MetaImage (.mhd + .raw/.zraw, .mha); little-endian,
 * one component.  frog_volume (frog_chain.h) views the file's own buffer until frog_volume_free.
 * frog_volume_write picks the format from the suffix: .mhd (header + zlib-compressed .zraw beside
 * it, vtkMetaImageWriter's default) or .nii / .nii.gz. */
typedef struct frog_volume_file frog_volume_file;
frog_volume_file *frog_volume_read(const char *path, int *status);
void frog_volume_free(frog_volume_file *f);
void frog_volume_view(const frog_volume_file *f, frog_volume *out);
int frog_volume_range(const frog_volume *v, double *lo, double *hi);
int frog_volume_write(const char *path, const frog_volume *v);

#ifdef __cplusplus
}
#endif
#endif
