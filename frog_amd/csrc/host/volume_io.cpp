// volume_io.cpp -- scalar volumes in and out for the reslicing tool (bin/VolumeTransform).
//
// The reference reads its volumes through vtkRobustImageReader (vtkOpenSURF3D, a git submodule that is
// absent from /root/reference) and writes them with vtkMetaImageWriter / vtkNIFTIImageWriter
// (tools/VolumeTransform.cxx:86-101, :146-182).  Here: the two formats the pipeline itself produces and
// consumes -- NIfTI-1 single files (.nii, .nii.gz) and MetaImage (.mhd header + .raw/.zraw data, or .mha)
// -- read and written from their published layouts.  Geometry as the rest of this build takes it
// (frog_volume_geometry): spacing from pixdim / ElementSpacing, origin from the qform offsets / Offset;
// axes are taken as aligned with the world axes (vtkImageData has no direction matrix either).
// Little-endian files only; one component per voxel.
#include "../common/parallel_gzip.h"
#include "frog_host.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <zlib.h>

namespace {

bool has_suffix(const std::string &s, const char *suffix)
{
    const size_t n = std::strlen(suffix);
    if (s.size() < n) return false;
    std::string tail = s.substr(s.size() - n);
    std::transform(tail.begin(), tail.end(), tail.begin(), [](unsigned char c) { return (char)std::tolower(c); });
    return tail == suffix;
}

bool slurp(const std::string &path, std::vector<unsigned char> &out)
{
    out.clear();
    gzFile f = gzopen(path.c_str(), "rb");           // zlib reads plain files transparently
    if (!f) return false;
    unsigned char buf[1 << 16];
    int n;
    while ((n = gzread(f, buf, sizeof buf)) > 0) out.insert(out.end(), buf, buf + n);
    gzclose(f);
    return n == 0;
}

template <class T> T field(const std::vector<unsigned char> &raw, size_t off) { T v; std::memcpy(&v, raw.data() + off, sizeof(T)); return v; }

int nifti_dtype(int code)
{
    switch (code) {
    case 2: return FROG_V_U8;   case 256: return FROG_V_I8;
    case 512: return FROG_V_U16; case 4: return FROG_V_I16;
    case 768: return FROG_V_U32; case 8: return FROG_V_I32;
    case 16: return FROG_V_F32;  case 64: return FROG_V_F64;
    default: return -1;
    }
}

const int NIFTI_CODE[8] = { 2, 256, 512, 4, 768, 8, 16, 64 };
const char *const MET_NAME[8] = { "MET_UCHAR", "MET_CHAR", "MET_USHORT", "MET_SHORT", "MET_UINT", "MET_INT", "MET_FLOAT", "MET_DOUBLE" };

std::string trim(const std::string &s)
{
    const size_t a = s.find_first_not_of(" \t\r\n");
    if (a == std::string::npos) return "";
    return s.substr(a, s.find_last_not_of(" \t\r\n") - a + 1);
}

} // namespace

struct frog_volume_file {
    frog_volume v;
    std::vector<unsigned char> bytes;
};

extern "C" {

frog_volume_file *frog_volume_read(const char *path, int *status)
{
    auto fail = [&](int code) { if (status) *status = code; return (frog_volume_file *)nullptr; };
    if (!path) return fail(FROG_E_INVALID);
    const std::string p(path);
    frog_volume_file *f = new frog_volume_file;
    std::memset(&f->v, 0, sizeof f->v);
    auto bad = [&](int code) { delete f; return fail(code); };
    if (has_suffix(p, ".mhd") || has_suffix(p, ".mha")) {
        std::ifstream in(path, std::ios::binary);
        if (!in) return bad(FROG_E_IO);
        std::string line, data_file, type;
        bool compressed = false, msb = false, have_dims = false;
        long header_size = 0;
        int ndims = 3;
        for (int k = 0; k < 3; k++) { f->v.dims[k] = 1; f->v.spacing[k] = 1; f->v.origin[k] = 0; }
        while (std::getline(in, line)) {
            const size_t eq = line.find('=');
            if (eq == std::string::npos) continue;
            const std::string key = trim(line.substr(0, eq)), val = trim(line.substr(eq + 1));
            std::stringstream vals(val);
            if (key == "NDims") vals >> ndims;
            else if (key == "DimSize") { for (int k = 0; k < 3 && k < ndims; k++) vals >> f->v.dims[k]; have_dims = true; }
            else if (key == "ElementSpacing") { for (int k = 0; k < 3 && k < ndims; k++) vals >> f->v.spacing[k]; }
            else if (key == "Offset" || key == "Position" || key == "Origin") { for (int k = 0; k < 3 && k < ndims; k++) vals >> f->v.origin[k]; }
            else if (key == "ElementType") type = val;
            else if (key == "CompressedData") compressed = (val == "True" || val == "true");
            else if (key == "BinaryDataByteOrderMSB" || key == "ElementByteOrderMSB") msb = (val == "True" || val == "true");
            else if (key == "HeaderSize") header_size = std::atol(val.c_str());
            else if (key == "ElementNumberOfChannels") { if (std::atoi(val.c_str()) != 1) return bad(FROG_E_INVALID); }
            else if (key == "ElementDataFile") { data_file = val; break; }       // always the last key; data may follow (LOCAL)
        }
        f->v.dtype = -1;
        for (int k = 0; k < 8; k++) if (type == MET_NAME[k]) f->v.dtype = k;
        if (!have_dims || ndims < 1 || ndims > 3 || f->v.dtype < 0 || msb || data_file.empty()) return bad(FROG_E_INVALID);
        const size_t want = (size_t)f->v.dims[0] * f->v.dims[1] * f->v.dims[2] * frog_volume_voxel_bytes(f->v.dtype);
        std::vector<unsigned char> raw;
        if (data_file == "LOCAL") {
            raw.assign(std::istreambuf_iterator<char>(in), std::istreambuf_iterator<char>());
        } else {
            const size_t slash = p.find_last_of("/\\");
            const std::string full = (data_file[0] == '/' || slash == std::string::npos) ? data_file : p.substr(0, slash + 1) + data_file;
            std::ifstream d(full, std::ios::binary);
            if (!d) return bad(FROG_E_IO);
            raw.assign(std::istreambuf_iterator<char>(d), std::istreambuf_iterator<char>());
        }
        if (header_size > 0 && (size_t)header_size <= raw.size()) raw.erase(raw.begin(), raw.begin() + header_size);
        if (compressed) {
            if (want / 1100 > raw.size() + 1) return bad(FROG_E_INVALID);   // deflate cannot expand more than ~1032:1
            f->bytes.resize(want);
            uLongf got = (uLongf)want;
            if (uncompress(f->bytes.data(), &got, raw.data(), (uLong)raw.size()) != Z_OK || got != want) return bad(FROG_E_INVALID);
        } else {
            if (raw.size() < want) return bad(FROG_E_INVALID);
            raw.resize(want);
            f->bytes.swap(raw);
        }
    } else {
        std::vector<unsigned char> raw;
        if (!slurp(p, raw)) return bad(FROG_E_IO);
        if (raw.size() < 352 || field<int32_t>(raw, 0) != 348) return bad(FROG_E_INVALID);
        const int nd = field<int16_t>(raw, 40);
        if (nd < 1 || nd > 7) return bad(FROG_E_INVALID);
        size_t extra = 1;
        for (int k = 4; k <= nd; k++) extra *= (size_t)std::max<int>(1, field<int16_t>(raw, 40 + 2 * k));
        if (extra != 1) return bad(FROG_E_INVALID);                  // time series / vectors: not a scalar volume
        for (int k = 0; k < 3; k++) {
            f->v.dims[k] = k < nd ? (uint32_t)std::max<int>(1, field<int16_t>(raw, 42 + 2 * k)) : 1u;
            f->v.spacing[k] = field<float>(raw, 80 + 4 * k);
            f->v.origin[k] = field<float>(raw, 268 + 4 * k);
        }
        f->v.dtype = nifti_dtype(field<int16_t>(raw, 70));
        const size_t off = (size_t)field<float>(raw, 108);
        if (f->v.dtype < 0) return bad(FROG_E_INVALID);
        const size_t want = (size_t)f->v.dims[0] * f->v.dims[1] * f->v.dims[2] * frog_volume_voxel_bytes(f->v.dtype);
        if (off < 348 || raw.size() < off + want) return bad(FROG_E_INVALID);
        f->bytes.assign(raw.begin() + off, raw.begin() + off + want);
    }
    for (int k = 0; k < 3; k++) if (!(f->v.spacing[k] != 0.0)) f->v.spacing[k] = 1.0;
    f->v.data = f->bytes.data();
    if (status) *status = FROG_OK;
    return f;
}

void frog_volume_free(frog_volume_file *f) { delete f; }
void frog_volume_view(const frog_volume_file *f, frog_volume *out) { if (f && out) *out = f->v; }

int frog_volume_range(const frog_volume *v, double *lo, double *hi)
{
    if (!v || !v->data) return FROG_E_INVALID;
    const size_t n = (size_t)v->dims[0] * v->dims[1] * v->dims[2];
    if (!n) return FROG_E_INVALID;
    double mn = 0, mx = 0;
    auto scan = [&](auto *p) { mn = mx = (double)p[0]; for (size_t i = 1; i < n; i++) { const double x = (double)p[i]; if (x < mn) mn = x; if (x > mx) mx = x; } };
    switch (v->dtype) {
    case FROG_V_U8: scan((const uint8_t *)v->data); break;   case FROG_V_I8: scan((const int8_t *)v->data); break;
    case FROG_V_U16: scan((const uint16_t *)v->data); break; case FROG_V_I16: scan((const int16_t *)v->data); break;
    case FROG_V_U32: scan((const uint32_t *)v->data); break; case FROG_V_I32: scan((const int32_t *)v->data); break;
    case FROG_V_F32: scan((const float *)v->data); break;    case FROG_V_F64: scan((const double *)v->data); break;
    default: return FROG_E_INVALID;
    }
    if (lo) *lo = mn;
    if (hi) *hi = mx;
    return FROG_OK;
}

int frog_volume_write(const char *path, const frog_volume *v)
{
    if (!path || !v || !v->data || v->dtype < 0 || v->dtype > 7) return FROG_E_INVALID;
    const std::string p(path);
    const size_t bytes = (size_t)v->dims[0] * v->dims[1] * v->dims[2] * frog_volume_voxel_bytes(v->dtype);
    if (!bytes) return FROG_E_INVALID;
    if (has_suffix(p, ".mhd")) {
        // what vtkMetaImageWriter emits by default: text header + zlib-compressed <name>.zraw beside it
        const size_t slash = p.find_last_of("/\\");
        const std::string base = p.substr(slash == std::string::npos ? 0 : slash + 1, p.size() - 4 - (slash == std::string::npos ? 0 : slash + 1));
        const std::string data_name = base + ".zraw", data_path = p.substr(0, p.size() - 4) + ".zraw";
        std::vector<unsigned char> z(compressBound((uLong)bytes));
        uLongf zn = (uLongf)z.size();
        if (compress2(z.data(), &zn, (const Bytef *)v->data, (uLong)bytes, 6) != Z_OK) return FROG_E_IO;
        FILE *d = std::fopen(data_path.c_str(), "wb");
        if (!d) return FROG_E_IO;
        bool ok = std::fwrite(z.data(), 1, zn, d) == zn;
        ok = (std::fclose(d) == 0) && ok;
        std::ofstream h(path, std::ios::trunc);
        if (!h) return FROG_E_IO;
        h.precision(17);
        h << "ObjectType = Image\nNDims = 3\nBinaryData = True\nBinaryDataByteOrderMSB = False\nCompressedData = True\n"
          << "CompressedDataSize = " << (unsigned long long)zn << "\n"
          << "TransformMatrix = 1 0 0 0 1 0 0 0 1\n"
          << "Offset = " << v->origin[0] << " " << v->origin[1] << " " << v->origin[2] << "\n"
          << "CenterOfRotation = 0 0 0\n"
          << "ElementSpacing = " << v->spacing[0] << " " << v->spacing[1] << " " << v->spacing[2] << "\n"
          << "DimSize = " << v->dims[0] << " " << v->dims[1] << " " << v->dims[2] << "\n"
          << "AnatomicalOrientation = ???\n"
          << "ElementType = " << MET_NAME[v->dtype] << "\n"
          << "ElementDataFile = " << data_name << "\n";
        h.close();
        return ok && h ? FROG_OK : FROG_E_IO;
    }
    if (has_suffix(p, ".nii") || has_suffix(p, ".nii.gz")) {
        for (int k = 0; k < 3; k++) if (v->dims[k] > 32767) return FROG_E_INVALID;
        unsigned char h[352];
        std::memset(h, 0, sizeof h);
        auto put = [&](size_t off, auto value) { std::memcpy(h + off, &value, sizeof value); };
        put(0, (int32_t)348);
        h[38] = 'r';
        put(40, (int16_t)3); put(42, (int16_t)v->dims[0]); put(44, (int16_t)v->dims[1]); put(46, (int16_t)v->dims[2]);
        for (int k = 4; k < 8; k++) put(40 + 2 * k, (int16_t)1);
        put(70, (int16_t)NIFTI_CODE[v->dtype]);
        put(72, (int16_t)(8 * frog_volume_voxel_bytes(v->dtype)));
        put(76, 1.0f);
        for (int k = 0; k < 3; k++) put(80 + 4 * k, (float)v->spacing[k]);
        for (int k = 4; k < 8; k++) put(76 + 4 * k, 1.0f);
        put(108, 352.0f);
        put(112, 1.0f);                                              // scl_slope
        h[123] = 2;                                                  // millimetres
        std::snprintf((char *)h + 148, 80, "frog_amd VolumeTransform");
        put(252, (int16_t)1); put(254, (int16_t)1);                  // qform_code, sform_code
        for (int k = 0; k < 3; k++) put(268 + 4 * k, (float)v->origin[k]);
        for (int k = 0; k < 3; k++) { put(280 + 16 * k + 4 * k, (float)v->spacing[k]); put(280 + 16 * k + 12, (float)v->origin[k]); }
        std::memcpy(h + 344, "n+1", 4);
        // a compressed volume of some size: one gzip member deflated chunk by chunk on all host threads (common/parallel_gzip.h;
        // level 6 as before -- a 256^3 int16 volume: 0.95 s on one thread)
        if (has_suffix(p, ".gz") && bytes >= ((size_t)4 << 20))
            return frog::gzip_write_parallel(path, { { h, sizeof h }, { v->data, bytes } }, 6) ? FROG_OK : FROG_E_IO;
        gzFile f = gzopen(path, has_suffix(p, ".gz") ? "wb6" : "wbT");   // "T": transparent, no compression
        if (!f) return FROG_E_IO;
        bool ok = gzwrite(f, h, sizeof h) == (int)sizeof h;
        const char *src = (const char *)v->data;
        size_t left = bytes;
        while (ok && left) {
            const unsigned chunk = (unsigned)std::min<size_t>(left, 1u << 30);
            ok = gzwrite(f, src, chunk) == (int)chunk;
            src += chunk; left -= chunk;
        }
        ok = (gzclose(f) == Z_OK) && ok;
        return ok ? FROG_OK : FROG_E_IO;
    }
    return FROG_E_INVALID;
}

}
