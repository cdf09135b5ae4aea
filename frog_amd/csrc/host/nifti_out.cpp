// nifti_out.cpp -- NIfTI-1 single-file (.nii / .nii.gz) writer for lattice images.
//
// Replaces the reference's use of vtkNIFTIImageWriter for the B-spline coefficient sidecars
// (tools/transformIO.h:196-208) and the error maps (registration/imageGroup.cxx:559-563).
// VTK is not part of this build, so the file is written straight from the NIfTI-1 standard
// (nifti1.h): 348-byte header, 4 zero extension bytes, voxel data at offset 352.  What the
// reference's readers take from such a file (tools/transformIO.h:439-453, transformIO.py:5-18,
// both through vtkNIFTIImageReader) is: the dimensions, the spacing (pixdim[1..3]), the
// components of a voxel (dim[5], stored plane by plane) and the origin = translation column of
// the qform matrix.  Hence: qform_code 1, unit quaternion (identity rotation, qfac +1),
// qoffset = image origin; the sform rows say the same.  Byte-for-byte identity with VTK's
// writer is not claimed (not checkable here: "parity unpinned"); the fields above are tested.
#include "frog_host.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <zlib.h>

namespace {

#pragma pack(push, 1)
struct Nifti1Header {
    int32_t sizeof_hdr;
    char data_type[10];
    char db_name[18];
    int32_t extents;
    int16_t session_error;
    char regular;
    char dim_info;
    int16_t dim[8];
    float intent_p1, intent_p2, intent_p3;
    int16_t intent_code;
    int16_t datatype;
    int16_t bitpix;
    int16_t slice_start;
    float pixdim[8];
    float vox_offset;
    float scl_slope, scl_inter;
    int16_t slice_end;
    char slice_code;
    char xyzt_units;
    float cal_max, cal_min;
    float slice_duration, toffset;
    int32_t glmax, glmin;
    char descrip[80];
    char aux_file[24];
    int16_t qform_code, sform_code;
    float quatern_b, quatern_c, quatern_d;
    float qoffset_x, qoffset_y, qoffset_z;
    float srow_x[4], srow_y[4], srow_z[4];
    char intent_name[16];
    char magic[4];
};
#pragma pack(pop)
static_assert(sizeof(Nifti1Header) == 348, "NIfTI-1 header is 348 bytes");

bool ends_with(const std::string &s, const char *suffix)
{
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

} // namespace

extern "C" int frog_nifti_write(const char *path, const uint32_t dims[3], const double spacing[3],
                                const double origin[3], uint32_t n_components, const float *interleaved)
{
    if (!path || !dims || !spacing || !origin || !interleaved || n_components == 0) return FROG_E_INVALID;
    for (int k = 0; k < 3; k++)
        if (dims[k] == 0 || dims[k] > 32767) return FROG_E_INVALID;             // dim[] is int16
    if (n_components > 32767) return FROG_E_INVALID;

    Nifti1Header h;
    std::memset(&h, 0, sizeof h);
    h.sizeof_hdr = 348;
    h.regular = 'r';
    h.dim[0] = n_components > 1 ? 5 : 3;
    h.dim[1] = (int16_t)dims[0]; h.dim[2] = (int16_t)dims[1]; h.dim[3] = (int16_t)dims[2];
    h.dim[4] = 1;
    h.dim[5] = (int16_t)n_components;
    h.dim[6] = 1; h.dim[7] = 1;
    h.intent_code = n_components > 1 ? 1007 : 0;                                // NIFTI_INTENT_VECTOR
    h.datatype = 16;                                                            // NIFTI_TYPE_FLOAT32
    h.bitpix = 32;
    h.pixdim[0] = 1.0f;                                                         // qfac
    for (int k = 0; k < 3; k++) h.pixdim[1 + k] = (float)spacing[k];
    h.pixdim[4] = 1.0f; h.pixdim[5] = 1.0f; h.pixdim[6] = 1.0f; h.pixdim[7] = 1.0f;
    h.vox_offset = 352.0f;
    h.scl_slope = 1.0f; h.scl_inter = 0.0f;
    h.xyzt_units = 2;                                                           // NIFTI_UNITS_MM
    std::snprintf(h.descrip, sizeof h.descrip, "frog_amd lattice image");
    h.qform_code = 1; h.sform_code = 1;                                         // NIFTI_XFORM_SCANNER_ANAT
    h.qoffset_x = (float)origin[0]; h.qoffset_y = (float)origin[1]; h.qoffset_z = (float)origin[2];
    h.srow_x[0] = (float)spacing[0]; h.srow_x[3] = (float)origin[0];
    h.srow_y[1] = (float)spacing[1]; h.srow_y[3] = (float)origin[1];
    h.srow_z[2] = (float)spacing[2]; h.srow_z[3] = (float)origin[2];
    std::memcpy(h.magic, "n+1", 4);

    // voxel data: the component is the slowest index (dim[5]), the lattice is x-fastest
    const size_t nvox = (size_t)dims[0] * dims[1] * dims[2];
    std::vector<float> planar(nvox * n_components);
    for (uint32_t c = 0; c < n_components; c++)
        for (size_t v = 0; v < nvox; v++) planar[c * nvox + v] = interleaved[v * n_components + c];
    const char ext[4] = { 0, 0, 0, 0 };

    const std::string p(path);
    if (ends_with(p, ".gz")) {
        // Level 1: float coefficients hardly compress (level 6 made the 700 sidecars of the benchmark group 8 % smaller than level 1
        // does and took three times as long: they are half of what bin/frog does after its last iteration).  A reader sees the same
        // bytes either way.  FROG_GZIP_LEVEL=0..9 overrides.
        static const int level = [] { const char *e = std::getenv("FROG_GZIP_LEVEL"); const int l = e ? std::atoi(e) : 1; return l < 0 || l > 9 ? 1 : l; }();
        const char mode[4] = { 'w', 'b', (char)('0' + level), 0 };
        gzFile f = gzopen(path, mode);
        if (!f) return FROG_E_IO;
        bool ok = gzwrite(f, &h, sizeof h) == (int)sizeof h && gzwrite(f, ext, 4) == 4;
        const char *bytes = reinterpret_cast<const char *>(planar.data());
        size_t left = planar.size() * sizeof(float);
        while (ok && left) {
            const unsigned chunk = (unsigned)std::min<size_t>(left, 1u << 30);
            ok = gzwrite(f, bytes, chunk) == (int)chunk;
            bytes += chunk; left -= chunk;
        }
        ok = (gzclose(f) == Z_OK) && ok;
        return ok ? FROG_OK : FROG_E_IO;
    }
    FILE *f = std::fopen(path, "wb");
    if (!f) return FROG_E_IO;
    bool ok = std::fwrite(&h, sizeof h, 1, f) == 1 && std::fwrite(ext, 4, 1, f) == 1
           && std::fwrite(planar.data(), sizeof(float), planar.size(), f) == planar.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok ? FROG_OK : FROG_E_IO;
}
