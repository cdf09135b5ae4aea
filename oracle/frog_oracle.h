/*
 * frog_oracle.h -- C ABI of the CPU oracle.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker
 * (or as the timed CPU baseline).  Nothing under frog_amd/ links, imports or
 * dlopen()s it; the product path fails when the HIP library is missing.
 *
 * What it is: a VTK-free C++/OpenMP restatement of the groupwise solver of
 * valette/FROG (registration/imageGroup.cxx, stats.{h,cxx}, image.cxx) over the
 * flat SoA/CSR model of include/frog_types.h, with the reference's threading
 * (omp parallel-for over images, serial inside an image), its float/double
 * promotion rules and its link traversal order.  Every function cites the
 * reference lines it follows.
 *
 * Parity status:
 *   - EM statistics (Stats::*) are PINNED: oracle/_ref builds the reference's own
 *     registration/stats.cxx unmodified and tests/test_oracle_stats.py compares
 *     the restatement against it bit-for-bit, and against tests/golden/ (stats_golden.json)
 *     generated from it (tests/golden/make_stats_golden.py).
 *   - The solver loops (imageGroup.cxx) and the VTK transform arithmetic are
 *     "PARITY UNPINNED": imageGroup.cxx needs VTK/Boost/picojson, which this
 *     image lacks, the reference ships no tests, golden vectors or sample
 *     inputs, and VTK is an un-vendored, version-unpinned dependency.  The
 *     restatement follows the source line by line (and VTK's published
 *     vtkLinearTransform / vtkBSplineTransform / vtkBoundingBox algorithms); it
 *     is self-checked by analytic properties in tests/test_oracle_solver.py.
 */
#ifndef FROG_ORACLE_H
#define FROG_ORACLE_H

#include "../include/frog_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct frogo_group frogo_group;
typedef struct frogo_stats frogo_stats;

/* ---- group (ImageGroup restated) ---------------------------------------- */
frogo_group *frogo_create(const frog_model *m, const frog_options *o);
void frogo_destroy(frogo_group *g);
void frogo_set_threads(int n);                       /* frog.cxx:143-145 (-nt)      */
int  frogo_get_max_threads(void);

void frogo_setup_stats(frogo_group *g);                           /* :1151 */
void frogo_linear_init(frogo_group *g, const float anchor[3]);    /* :806  */
void frogo_transform_points(frogo_group *g, int apply);           /* :910  */
void frogo_update_stats(frogo_group *g);                          /* :569  */
double frogo_linear_step(frogo_group *g);                         /* :1063 */
void frogo_deformable_setup(frogo_group *g, int level, frog_grid_info *out); /* :159 */
double frogo_deformable_step(frogo_group *g, float alpha);        /* :234  */
void frogo_count_inliers(frogo_group *g, frog_counts *per_image); /* :988  */
int  frogo_set_hard_links(frogo_group *g, const uint64_t *point, const uint64_t *partner, size_t n, float weight2); /* :1210 */
int  frogo_error_map(frogo_group *g, uint32_t image, float *out4G, size_t cap_floats); /* :475 */

/* Split phases (several instances owning disjoint image ranges, combined by the caller:
 * the CPU stand-in for one-process-per-GPU runs in tests/test_distributed_gloo.py). */
void frogo_set_range(frogo_group *g, uint32_t image_begin, uint32_t image_end);
void frogo_linear_step_local(frogo_group *g, double out2[2]);
void frogo_bounds_local(frogo_group *g, double mins[3], double maxs[3]);
void frogo_deformable_setup_bounds(frogo_group *g, int level, const double mins[3], const double maxs[3], frog_grid_info *out);
void frogo_deformable_phase_a(frogo_group *g, float alpha, double *gridsum3G, double out2[2]);
long frogo_deformable_phase_b(frogo_group *g, const double *gridsum_all3G);
void frogo_deformable_phase_c(frogo_group *g);

/* Whole driver (run(), imageGroup.cxx:31-157) without file output.  E_out
 * receives one value per ACCEPTED iteration (the `measures` vector).
 * Returns the number of values written (<= cap); n_grids_out[level] gets the
 * number of lattices the level used. */
int frogo_run(frogo_group *g, int linear_iterations, int deformable_levels,
              int deformable_iterations, float deformable_alpha,
              int stat_interval, const float anchor[3],
              double *E_out, int cap, int *n_grids_out);

/* RANSAC + RANSACBatch (:629-804), ransac_oracle.cpp; returns the best candidate's inlier count */
long frogo_ransac(frogo_group *g, uint32_t image, int iterations, int batches, float inlier_distance, float max_scale);
const float *frogo_xyz_ptr(const frogo_group *g);
const float *frogo_xyz2_ptr(const frogo_group *g);
const uint32_t *frogo_point_offset_ptr(const frogo_group *g);
const uint64_t *frogo_row_ptr(const frogo_group *g);
const uint16_t *frogo_link_image_ptr(const frogo_group *g);
const uint32_t *frogo_link_point_ptr(const frogo_group *g);
void frogo_set_matrix(frogo_group *g, uint32_t image, const double in16[16]);

/* state read-back */
uint64_t frogo_num_points(const frogo_group *g);
void frogo_get_xyz(const frogo_group *g, float *out3P);
void frogo_get_xyz2(const frogo_group *g, float *out3P);
void frogo_set_xyz2(frogo_group *g, const float *in3P);
void frogo_get_matrix(const frogo_group *g, uint32_t image, double out16[16]);
void frogo_get_em(const frogo_group *g, uint32_t image, float out3[3]);
void frogo_set_em(frogo_group *g, uint32_t image, const float in3[3]);
int  frogo_get_samples(const frogo_group *g, uint32_t image, float *out, int cap);
/* ordinal (0-based position in the image's half-link traversal) of each sample
 * retained by the LAST updateStats */
int  frogo_get_sample_ordinals(const frogo_group *g, uint32_t image, uint32_t *out, int cap);
int  frogo_get_histogram(frogo_group *g, uint32_t image, float *out, int cap);
int  frogo_num_grids(const frogo_group *g);
int  frogo_get_grid(const frogo_group *g, uint32_t image, int k, frog_grid_info *info,
                    float *coeffs3G, size_t cap_floats);
/* per-point accumulators of the last deformable step (sDisp xyz, sWeight) */
void frogo_get_point_sums(const frogo_group *g, float *out4P);
/* proposed coefficients after the control-point step and mean removal are not
 * kept; the gradient image before the step is: */
int  frogo_get_gradient(const frogo_group *g, uint32_t image, float *out4G, size_t cap_floats);
/* oracle-only: keep a copy of the gradient image as the scatter leaves it (imageGroup.cxx:301-338), before the
 * control-point step overwrites its first three components with the proposals; -1 when not kept */
void frogo_keep_raw_gradient(frogo_group *g, int on);
int  frogo_get_gradient_raw(const frogo_group *g, uint32_t image, float *out4G, size_t cap_floats);

/* ---- Stats restated (stats.h / stats.cxx), usable stand-alone ------------- */
frogo_stats *frogo_stats_new(int max_size, int max_iterations, float epsilon);
void  frogo_stats_free(frogo_stats *s);
void  frogo_stats_add_slots(frogo_stats *s, int n);               /* stats.h:36 */
void  frogo_stats_reset(frogo_stats *s);                          /* stats.h:52 */
void  frogo_stats_add_samples(frogo_stats *s, const float *v, int n); /* stats.h:58 */
void  frogo_stats_estimate(frogo_stats *s);                       /* stats.cxx:14 */
float frogo_stats_inlier_probability(const frogo_stats *s, float d);  /* stats.h:84 */
void  frogo_stats_get_params(const frogo_stats *s, float out3[3]);
void  frogo_stats_set_params(frogo_stats *s, const float in3[3]);
int   frogo_stats_size(const frogo_stats *s);
int   frogo_stats_get_samples(const frogo_stats *s, float *out, int cap);
int   frogo_stats_histogram(frogo_stats *s, float bin, float *out, int cap); /* stats.cxx:121 */
float frogo_chipdf(float x);                                      /* stats.h:10 */
void  frogo_bspline_weights_n(const double *f, int n, double *out4n); /* imageGroup.cxx:221-232, pinned by _ref/libfrog_refweights.so */

#ifdef __cplusplus
}
#endif
#endif
