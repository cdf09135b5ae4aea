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
 * Inverse of a B-spline link (FROG_T_BSPLINE_INVERSE): Newton's method with vtkWarpTransform's
 *              defaults (tolerance 1e-3, 500 iterations, step shortening when the residual grows);
 *              the inverse of a linear link is its inverted matrix (frog_chain_invert_links).
 */
#ifndef FROG_CHAIN_H
#define FROG_CHAIN_H

#include <stddef.h>
#include <stdint.h>

#include "frog_types.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { FROG_T_LINEAR = 0, FROG_T_BSPLINE = 1, FROG_T_BSPLINE_INVERSE = 2 };

typedef struct frog_chain_link {
    int type;                   /* FROG_T_LINEAR | FROG_T_BSPLINE | FROG_T_BSPLINE_INVERSE */
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

/* vtkGeneralTransform::Inverse() of a chain (tools/VolumeTransform.cxx:55-57, PointsTransform's -ti):
 * the links in reverse order, matrices inverted, lattices switched between forward and inverse
 * evaluation.  `out` receives n links (coefficient pointers are shared with `in`).  Returns
 * FROG_E_INVALID for a singular matrix. */
int frog_chain_invert_links(const frog_chain_link *in, uint32_t n, frog_chain_link *out);

/* ---- volume reslicing (tools/VolumeTransform.cxx:119-136 = vtkImageReslice) ----------------------
 * A scalar volume on a regular grid; `data` is x-fastest, one component. */
enum { FROG_V_U8 = 0, FROG_V_I8, FROG_V_U16, FROG_V_I16, FROG_V_U32, FROG_V_I32, FROG_V_F32, FROG_V_F64 };
typedef struct frog_volume {
    uint32_t dims[3];
    double   spacing[3], origin[3];
    int      dtype;             /* FROG_V_*                                                     */
    void    *data;
} frog_volume;
static inline size_t frog_volume_voxel_bytes(int dtype)
{
    switch (dtype) {
    case FROG_V_U8: case FROG_V_I8: return 1;
    case FROG_V_U16: case FROG_V_I16: return 2;
    case FROG_V_U32: case FROG_V_I32: case FROG_V_F32: return 4;
    case FROG_V_F64: return 8;
    default: return 0;
    }
}

/* out(voxel) = source(chain(position of the voxel)): `chain` maps the output grid's space to the
 * source's (for a registration transform T of the source that is T^-1: frog_chain_invert_links).
 * `out` describes the output grid (VolumeTransform takes it from the reference volume); its
 * dtype must be the source's and its data buffer is filled.  interpolation: 0 nearest, otherwise
 * trilinear.  A sample more than half a voxel outside the source's voxel centres gives
 * `background` (VTK's default border); integer outputs are rounded half up and clamped to the
 * type's range, as vtkImageReslice does. */
int frog_chain_reslice(frog_chain *c, const frog_volume *source, frog_volume *out, int interpolation, double background);

#ifdef __cplusplus
}
#endif
#endif
