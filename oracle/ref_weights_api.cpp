// ref_weights_api.cpp -- C ABI around the REFERENCE's own vtkBSplineTransformWeights (imageGroup.cxx:221-232).
//
// TEST INFRASTRUCTURE.  This file contains no reference code.  imageGroup.cxx as a whole cannot be built here (VTK, Boost,
// picojson absent), but that one function is self-contained -- plain double arithmetic, no VTK symbol -- so oracle/Makefile
// (target `ref`) cuts exactly its definition out of /root/reference/registration/imageGroup.cxx where it lies, into a
// temporary file outside the repository, passes that file's path as REF_WEIGHTS_SLICE, compiles this wrapper around it into
// oracle/_ref/libfrog_refweights.so and deletes the temporary file.  Only the .so stays (git-ignored; it travels to the GPU
// box like any other built .so); no reference text enters the tree.
//
// Purpose: pin D2 of SURVEY.md 8(a) -- the cubic B-spline basis the scatter (imageGroup.cxx:312-316) weights every tap with --
// bit for bit: tests/test_oracle_weights.py (the oracle's restatement) and tests/test_gpu_round5.py (the device's).

#ifndef REF_WEIGHTS_SLICE
#error "REF_WEIGHTS_SLICE must name the file holding the sliced definition (see oracle/Makefile)"
#endif
#include REF_WEIGHTS_SLICE

extern "C" void refweights_n(const double *f, int n, double *out4n)
{
    for (int i = 0; i < n; i++) vtkBSplineTransformWeights(out4n + 4 * (long)i, f[i]);
}
