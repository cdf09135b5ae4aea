/*
 * frog_types.h -- plain-data types shared by the device library (frog_hip.h),
 * the host library (frog_host.h) and the CPU oracle (oracle/frog_oracle.h).
 *
 * Everything here is POD with fixed-width members so that the same layout can
 * be described from ctypes / cgo / JNI without a C++ compiler.
 *
 * Reference data model this flattens (all paths relative to /root/reference):
 *   Link  {u16 image, u32 point}            registration/point.h:11-16
 *   Point {other[3], xyz[3], xyz2[3], links} registration/point.h:19-32
 *   Image {points, stats, transform, ...}   registration/image.h:10-28
 *   option fields + defaults                registration/imageGroup.h:14-82
 *   Stats statics                           registration/stats.cxx:10-12
 */
#ifndef FROG_TYPES_H
#define FROG_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Keypoint group as parsed from pairs.bin (imageGroup.cxx:1353-1417), flattened
 * to SoA + CSR.  Points of image i are the global indices
 * [point_offset[i], point_offset[i+1]).  Half-links of global point p are the
 * CSR entries [row_ptr[p], row_ptr[p+1]) IN REFERENCE ORDER, i.e. the order in
 * which readPairs push_back()s them (imageGroup.cxx:1405-1406): that order
 * fixes the f32 summation order of the deformable step and the order in which
 * updateStats feeds samples to the reservoir.  The caller keeps ownership. */
typedef struct frog_model {
    uint32_t        n_images;
    const uint32_t *point_offset;   /* [n_images + 1]                         */
    const float    *xyz;            /* [3 * P]   Point::xyz                    */
    const uint64_t *row_ptr;        /* [P + 1]                                 */
    const uint16_t *link_image;     /* [L]       Link::image                   */
    const uint32_t *link_point;     /* [L]       Link::point (index in image)  */
} frog_model;

/* Solver options that reach the hot path.  Defaults: frog_options_default(). */
typedef struct frog_options {
    float   linear_alpha;               /* -la   0.5    imageGroup.h:68 */
    int32_t use_scale;                  /* -s    1      imageGroup.h:79 */
    float   initial_grid_size;          /* -g    100    imageGroup.h:66 */
    float   bounding_box_margin;        /*       0.1    imageGroup.h:59 */
    float   inlier_threshold;           /* -t    0.5    imageGroup.h:67 */
    int32_t guarantee_diffeomorphism;   /* -gd   1      imageGroup.h:63 */
    float   max_displacement_ratio;     /* -gm   0.4    imageGroup.h:73 */
    int32_t stats_max_size;             /* -ss   10000  stats.cxx:10    */
    int32_t stats_max_iterations;       /* -emi  10000  stats.cxx:11    */
    float   stats_epsilon;              /* -se   1e-6   stats.cxx:12    */
    int32_t n_fixed_images;             /* -fi   0      imageGroup.h:69: the first n images are already
                                         * registered (their xyz is final) and never move            */
    int32_t max_levels_hint;            /* -dl          0 = unknown.  A HINT, not a limit: how many deformable levels the caller
                                         * is going to run (imageGroup.h:62 deformableLevels).  frog_create then sizes the
                                         * lattice buffers for the finest of them at once, so that no multi-gigabyte device
                                         * allocation happens between two iterations                                  */
    int32_t reference_order;            /* -exact 0     0 = the product kernels (sums re-associated: per-point sums per partner
                                         * group, tree-reduced f64 sums, tiled scatter, fast f32 weight, f32 B-spline
                                         * evaluation; results within the bars of DESIGN.md 2a of the mode below).
                                         * 1 = every solver loop in the reference's own order and arithmetic (one f32
                                         * chain per point in readPairs order, getInlierProbability with its f64 exp, the
                                         * scatter image by image and point by point, f64 B-spline evaluation): the mode
                                         * that meets "transform parameters within 1e-4 relative" on RAW coefficients -- it
                                         * has the CPU restatement's bits -- at about 1/10 of the speed (285 against 2 500
                                         * iterations/s on the 100-image benchmark group)                 */
    int32_t selections_in_background;   /*       0      frog_create draws the reservoir selections of the first refreshes ahead of
                                         * time on a side stream (replayUpdateStats' rand() sequence, imageGroup.cxx:887-933).
                                         * 0 = and waits for them: the context starts with an idle side stream and a timed
                                         * loop sees none of it.  1 = returns at once: the draws run beside the first
                                         * iterations (a refresh waits for its own selection only) -- what a whole run wants
                                         * (bin/frog: 0.1 s of 2)                                                       */
    int32_t reserved[2];                /* must be 0                      */
} frog_options;

/* Geometry of one B-spline control-point lattice
 * (setupDeformableTransforms, imageGroup.cxx:159-218). */
typedef struct frog_grid_info {
    int32_t dims[3];        /* control points per axis (= cells + 3) */
    int32_t n_grid;         /* ordinal of this lattice in the chain (0 = first) */
    double  origin[3];
    double  spacing[3];
    double  bbox[6];        /* scaled bounding box xmin,xmax,ymin,ymax,zmin,zmax */
} frog_grid_info;

/* Per-image link census (countInliers, imageGroup.cxx:988-1060). */
typedef struct frog_counts {
    int64_t points;
    int64_t pairs;      /* half-links of the image */
    int64_t inliers;
    int64_t outliers;
    float   c1, c2, ratio;
    float   pad_;
} frog_counts;

enum {
    FROG_OK            = 0,
    FROG_E_INVALID     = 1,   /* bad argument / unsupported option            */
    FROG_E_NODEVICE    = 2,   /* no usable HIP device: there is no CPU fallback */
    FROG_E_HIP         = 3,   /* a HIP runtime call failed                    */
    FROG_E_STATE       = 4,   /* call sequence violated                       */
    FROG_E_NOMEM       = 5,
    FROG_E_IO          = 6    /* a file could not be written                  */
};

/* Fills *o with the reference defaults listed above. */
static inline void frog_options_default(frog_options *o)
{
    size_t i;
    o->linear_alpha = 0.5f;
    o->use_scale = 1;
    o->initial_grid_size = 100.0f;
    o->bounding_box_margin = 0.1f;
    o->inlier_threshold = 0.5f;
    o->guarantee_diffeomorphism = 1;
    o->max_displacement_ratio = 0.4f;
    o->stats_max_size = 10000;
    o->stats_max_iterations = 10000;
    o->stats_epsilon = 1e-6f;
    o->n_fixed_images = 0;
    o->max_levels_hint = 0;
    o->reference_order = 0;
    o->selections_in_background = 0;
    for (i = 0; i < sizeof(o->reserved) / sizeof(o->reserved[0]); i++) o->reserved[i] = 0;
}

#ifdef __cplusplus
}
#endif
#endif /* FROG_TYPES_H */
