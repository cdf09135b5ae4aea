// CheckDiffeomorphism: count the negative Jacobian determinants of a FROG transform chain on the
// voxel grid of a volume, on the GPU (tools/CheckDiffeomorphism.cxx).
//   CheckDiffeomorphism image transform [spacing]
// `image` gives the grid (NIfTI-1 .nii/.nii.gz or MetaImage .mhd header; the voxels are not read).
// With `spacing` the grid is resampled to that isotropic spacing over the same extent (upstream:
// vtkImageResize with OutputSpacing; here n = max(1, round(n_old * old_spacing / spacing)) nodes
// per axis from the same origin).  Exit code 1 if any determinant is negative, as upstream.
#include "frog_chain.h"
#include "frog_host.h"

#include <cmath>
#include <cstdlib>
#include <iomanip>
#include <iostream>

extern "C" const char *frog_last_error(void);
using std::cout;
using std::endl;

int main(int argc, char *argv[])
{
    if (argc < 3) {
        std::cout << "Usage : CheckDiffeomorphism image transform [spacing]" << std::endl;
        exit(1);
    }
    std::cout << "load : " << argv[1] << std::endl;
    uint32_t dimensions[3];
    double origin[3], spacing[3];
    if (frog_volume_geometry(argv[1], dimensions, spacing, origin)) { cout << "Error : cannot read the grid of " << argv[1] << endl; exit(1); }
    int status = 0;
    frog_transform_file *f = frog_transform_read(argv[2], &status);
    if (!f) { cout << "Error : cannot read transform " << argv[2] << endl; exit(1); }
    if (argc > 3) {
        const double sp = atof(argv[3]);
        if (sp > 0) {
            cout << "Resizing image with spacing : " << sp << endl;
            for (int k = 0; k < 3; k++) {
                dimensions[k] = (uint32_t)std::max(1.0, std::floor(dimensions[k] * spacing[k] / sp + 0.5));
                spacing[k] = sp;
            }
        }
    }
    cout << "Computing Jacobian determinants..." << endl;
    frog_chain *c = nullptr;
    uint64_t n = 0;
    double min_det = 0;
    if (frog_chain_create(frog_transform_links(f), frog_transform_num_links(f), 0, &c)
        || frog_chain_check(c, origin, spacing, dimensions, &n, &min_det)) {
        cout << "Error : " << frog_last_error() << endl;
        exit(1);
    }
    cout << n << " negative jacobian determinant values (" << std::setprecision(3)
         << (float)100.0 * n / ((double)dimensions[0] * dimensions[1] * dimensions[2]) << "%) " << endl;
    frog_chain_destroy(c);
    frog_transform_free(f);
    return n > 0;
}
