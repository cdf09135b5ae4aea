/* frog_chain.h -- C ABI for applying and checking a FROG transform chain (the step after the
 * registration): what `frog` writes to transforms/<i>.json, evaluated on the GPU.
 *
 * Replaces, for forward evaluation,
 *   tools/PointsTransform.cxx:70-80      vtkGeneralTransform::TransformPoint on a point
 *   tools/CheckDiffeomorphism.cxx:67-85  InternalTransformDerivative on every voxel of a
 *                                        grid, count of negative Jacobian determinants
 * The chain is the PostMultiply concatenation the reference's readers build
 * (tools/transformIO.h:375-460): transforms are applied in listed order, a
 * vtkMatrixToLinearTransform first, then one vtkBSplineTransform per lattice.
 *
 * Arithmetic (f64 throughout, as VTK's double path; VTK itself is absent from the image,
 * so this follows its documented semantics -- parity unpinned):
 *   linear   : y = M [x 1]
 *   B-spline : u = (x - origin) / spacing, cubic uniform basis (the F0..F3 of
 *              registration/imageGroup.cxx:221-232), 4x4x4 taps, taps outside the lattice
 *              contribute zero (BorderModeZero), y = x + d(x)
 *   Jacobian : analytic, I + (basis derivative / spacing) products; the chain's Jacobian is
 *              the product of the links' Jacobians at the successive points.
 * Inverse transforms (-ti) are not built.
 */
#ifndef FROG_CHAIN_H
#define FROG_CHAIN_H

#include <stddef.h>
#include <stdint.h>

#include "frog_types.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { FROG_T_LINEAR = 0, FROG_T_BSPLINE = 1 };

typedef struct frog_chain_link {
    int type;                   /* FROG_T_LINEAR | FROG_T_BSPLINE                        */
    double matrix[16];          /* linear: row-major 4x4                                  */
    uint32_t dims[3];           /* B-spline: control points per axis                      */
    double origin[3], spacing[3];
    const float *coeffs;        /* B-spline: dims[0]*dims[1]*dims[2] x 3 floats, x fastest */
} frog_chain_link;

typedef struct frog_chain frog_chain;

/* Copies the links (and their coefficients) to `device`. */
int frog_chain_create(const frog_chain_link *links, uint32_t n_links, int device, frog_chain **out);
void frog_chain_destroy(frog_chain *c);
uint32_t frog_chain_num_links(const frog_chain *c);

/* out[i] = chain(in[i]), n points of 3 doubles (host arrays). */
int frog_chain_apply(frog_chain *c, const double *in3n, double *out3n, size_t n);

/* Jacobian determinant of the chain at origin + (i,j,k)*spacing for every node of a dims grid
 * (CheckDiffeomorphism.cxx:67-85): number of nodes with a negative determinant and the
 * smallest determinant met. */
int frog_chain_check(frog_chain *c, const double origin[3], const double spacing[3], const uint32_t dims[3],
                     uint64_t *n_negative, double *min_determinant);

#ifdef __cplusplus
}
#endif
#endif
