/* frog_match.h -- C ABI of the keypoint matcher (the producer of pairs.bin).
 *
 * Replaces the pairing stage of the reference's `match` tool (match/match.cpp):
 *   ComputeMatches(points2, points1, threshold, dist2second, matchAll, anatVal, sym)
 *                                                            match.cpp:255-336
 * called once per image pair (first < second) from main (match.cpp:640-660) as
 *   ComputeMatches(*allPoints[first], *allPoints[second], dist, dist2second, ...)
 * i.e. every keypoint of image `second` (the query) scans all keypoints of image `first`
 * (the candidates) that pass the Laplacian-sign, scale-ratio and optional anatomical
 * tests, keeps the nearest and the second nearest descriptor (squared L2, f32, summed
 * in dimension order as the scalar `norm`, match.cpp:242-251) and emits the pair
 * (candidate, query) when sqrt(d1/d2) < dist2second (or there is no second) and
 * sqrt(d1) < threshold.  Pairs come out in query order, as upstream.
 *
 * Arithmetic contract: every comparison the reference makes is made on the same f32
 * values (no FMA contraction, IEEE division and square root), so the pair lists are
 * identical to the scalar build of the reference -- index work, bit-exact.  (Upstream's
 * USE_SSE_FOR_MATCHING build sums the dimensions in another order; it is a different
 * reference.)
 *
 * The anatomical test (`-anat`) compares frog_keypoints.xyz: pass the positions it is meant for
 * (bin/match -transformPrefix hands over the transformed ones, match.cpp:517-558).
 * matchAll (`-all`, match.cpp:297-302): every candidate within the threshold emits a pair at once -- and, as upstream, the
 * index pushed is the `match` variable (the nearest candidate so far that was NOT within the threshold, carried over from
 * earlier queries when there is none yet), not the candidate itself; the second-nearest test does not apply.  Reproduced as
 * it is, in the caller's candidate order, by a sequential pair of kernels (device/match.hip match_all_kernel).
 */
#ifndef FROG_MATCH_H
#define FROG_MATCH_H

#include <stddef.h>
#include <stdint.h>

#include "frog_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* One image's keypoints, host arrays (the rows of a surf3d .csv/.csv.gz/.bin file,
 * match.cpp:48-83: x, y, z, scale, laplacianSign, response, descriptor...). */
typedef struct frog_keypoints {
    uint32_t n;                 /* keypoints                                   */
    uint32_t dim;               /* descriptor length (48 for surf3d)           */
    const float *xyz;           /* [n][3]                                      */
    const float *scale;         /* [n], > 0                                    */
    const float *laplacian;     /* [n]                                         */
    const float *response;      /* [n] (not used by the pairing)               */
    const float *desc;          /* [n][dim]                                    */
} frog_keypoints;

typedef struct frog_match_options {
    float threshold;            /* -d   (match.cpp:357: 0.22)                  */
    float dist2second;          /* -d2  (1)                                    */
    float anat;                 /* -anat (0 = off)                             */
    int sym;                    /* -sym: also match first against second       */
    int all;                    /* -all: matchAll (match.cpp:297-302)          */
    int reserved[3];
} frog_match_options;

void frog_match_options_default(frog_match_options *o);

typedef struct frog_matcher frog_matcher;

/* Copies every image's keypoints to `device` (they stay resident for all pairs). */
int frog_matcher_create(const frog_keypoints *images, uint32_t n_images, int device, frog_matcher **out);
void frog_matcher_destroy(frog_matcher *m);

/* ComputeMatches for `n_jobs` image pairs (first[k], second[k]); with o->sym the reverse
 * direction is appended as upstream (match.cpp:645-648).  Job k's pairs are returned in
 * p_first[offset[k] .. offset[k+1]) / p_second[...] (indices inside image first[k] /
 * second[k]); `offset` has n_jobs+1 entries.  The two arrays are allocated by the library
 * (free them with frog_match_free).  Jobs are pipelined on the device. */
int frog_matcher_run(frog_matcher *m, const uint16_t *first, const uint16_t *second, size_t n_jobs,
                     const frog_match_options *o, uint64_t *offset, uint32_t **p_first, uint32_t **p_second);
void frog_match_free(void *p);

/* HIP-event time of the pairing kernels of the last frog_matcher_run (ms) and the number of
 * (query, candidate) descriptor distances it evaluated. */
/* Test hook: the matrix-core filter's approximate -|q - c|^2 / 2 for 32 candidates x 32 queries (dim = 48 or 64, row-major
 * descriptors), out[c * 32 + q]: form 0 = the f32 instruction chain, 1 = three products of bf16 (hi, lo) splits -- through the
 * operand builders and instruction sequences the filter kernels use.  *bound = the relative bound (on -2 x product, in units
 * of |q|^2 + |c|^2) the verification stage allows for that form. */
int frog_match_test_products(int device, const float *cand, const float *query, uint32_t dim, int form, float *out, float *bound);

int frog_matcher_last_stats(const frog_matcher *m, double *kernel_ms, double *distances);
/* Passes (ComputeMatches calls: one per job, two with -sym) of the last frog_matcher_run by the form that ran them:
 * [0] the exact vector-ALU kernel alone (FROG_MATCH_VALU, -all, descriptors longer than 64 or non-finite), [1] the f32
 * matrix-core filter (v_mfma_f32_32x32x2_f32) + exact verification, [2] the bf16 matrix-core filter (three products of (hi, lo)
 * splits, v_mfma_f32_32x32x16_bf16) + exact verification.  What bench_match.py's `issued` figures are taken from. */
int frog_matcher_last_forms(const frog_matcher *m, uint64_t passes_by_form[3]);

#ifdef __cplusplus
}
#endif
#endif
